#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_intervals.py tests/test_gpu_builders_quicksect.py -m gpu -q -x --timeout 500 -p no:cacheprovider -k "find_through_the_exchange or find_join_scale or find_on_sorted or bitset_utils" > gpurun_out/t_find.log 2>&1
echo "find tests rc=$?"; tail -4 gpurun_out/t_find.log | cut -c1-400
MODE=random timeout 200 python tools/bench_find.py 2>/dev/null | tee gpurun_out/find_random.json | cut -c1-200
MODE=sorted timeout 200 python tools/bench_find.py 2>/dev/null | tee gpurun_out/find_sorted.json | cut -c1-200
