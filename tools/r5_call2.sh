#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/r5_debug.py > gpurun_out/debug_fx.log 2>&1; echo "debug rc=$?"; cat gpurun_out/debug_fx.log | cut -c1-700
timeout 900 python -m pytest tests/test_gpu_intervals.py -m gpu -q --timeout 600 -p no:cacheprovider \
  -k "not find_through_the_exchange and not cfg5_full_size and not find_join_scale and not cfg2_full and not genome_cfg4" > gpurun_out/t_rest.log 2>&1
echo "rest rc=$?"; tail -40 gpurun_out/t_rest.log | cut -c1-300
