#!/bin/bash
# find tests, then tools/bench_find.py under each BXMI_OPTS setting of OPTSETS (";"-separated; "-" = defaults)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 800 -p no:cacheprovider -k "${TESTS:-find_through_the_exchange or find_join_scale}" > gpurun_out/t_find.log 2>&1
echo "find tests rc=$?"; tail -3 gpurun_out/t_find.log | cut -c1-600
IFS=';' read -ra SETS <<< "${OPTSETS:--}"
for o in "${SETS[@]}"; do
  for r in 1 2; do echo -n "[$o] "; BXMI_OPTS=$([ "$o" = "-" ] && echo "" || echo "$o") MODE=${MODE:-random} timeout 200 python tools/bench_find.py 2>&1 | tail -1 | cut -c95-135; done
done
