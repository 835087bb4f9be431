#!/bin/bash
# sorted-batch tests of tests/test_gpu_intervals.py, then the kernel stats of tools/bench_find.py on sorted queries
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 800 -p no:cacheprovider -k "${TESTS:-sorted}" > gpurun_out/t_sorted.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|error" gpurun_out/t_sorted.log | tail -3 | cut -c1-300
MODE=sorted bash tools/prof_find.sh 2>&1 | grep -E "ivl_local|part_fill|lf_off|fx_tile|sorted_check|probe"
for r in 1 2; do MODE=sorted timeout 200 python tools/bench_find.py 2>&1 | tail -1 | cut -c95-135; done
