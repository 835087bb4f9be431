#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 500 -p no:cacheprovider -k "find_through_the_exchange or find_join_scale" > gpurun_out/t_find.log 2>&1
echo "find tests rc=$?"; tail -3 gpurun_out/t_find.log | cut -c1-400
VARIANTS="default nomarks default nomarks" bash tools/ab_find_variants.sh
