#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 800 -p no:cacheprovider -k "total_only or count_multi or width_feedback or bitmap_pass_differential" > gpurun_out/t_tot.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/t_tot.log | cut -c1-500
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-find --no-bitset > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_q.json').read().strip().splitlines()[-1])
print("ms", d['ms_per_step'], "total_only", d['total_only'], "genome", d['genome']['ms_per_step'], d['genome']['total_only'], d['genome']['sorted_queries']['ms_per_step'])
PY
