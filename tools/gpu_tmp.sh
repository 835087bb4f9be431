#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2i
timeout 1200 python -m pytest tests/test_gpu_bitset.py -m gpu -x -q --timeout 900 -p no:cacheprovider > gpurun_out/r2i/bitset.log 2>&1; echo "bitset rc=$?"
timeout 1200 python -m pytest tests/test_gpu_cli.py tests/test_gpu_operations.py -m gpu -x -q --timeout 900 -p no:cacheprovider > gpurun_out/r2i/cli_ops.log 2>&1; echo "cli+ops rc=$?"
timeout 1500 python -m pytest tests/test_gpu_intervals.py -m gpu -x -q --timeout 1200 -p no:cacheprovider -k "cfg5_full" > gpurun_out/r2i/join.log 2>&1; echo "join rc=$?"
tail -15 gpurun_out/r2i/bitset.log; tail -5 gpurun_out/r2i/cli_ops.log; tail -15 gpurun_out/r2i/join.log
