#!/bin/bash
# sorted batches on cell images: parity tests, then old vs new kernel on configs[1] sorted by start and on a sparse index
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/sorted
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 500 -p no:cacheprovider -k "${K:-sorted or order_check or bitmap_pass_differential or random_differential}" > gpurun_out/sorted/tests.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/sorted/tests.log | cut -c1-600
ORDER=sorted REPS=10 VARIANTS="new:,old:ivl.sorted_cells=0" timeout 300 python tools/count_variants.py 2>&1 | cut -c1-200 | tee gpurun_out/sorted/dense.log
NT=800000 ORDER=sorted REPS=10 VARIANTS="new:,old:ivl.sorted_cells=0" timeout 300 python tools/count_variants.py 2>&1 | cut -c1-200 | tee gpurun_out/sorted/sparse.log
