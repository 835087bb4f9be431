#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3u
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_intervals.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "order_check or order_checks or count_width or sorted" 2>&1 | tail -4
cd /tmp
REPS=10 VARIANTS="auto:,nocheck:ivl.sorted_path=0,check:ivl.order_skip=0,auto2:" timeout 200 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/v.json 2> $OUT/trace.err
cd $REPO
cut -c1-100 $OUT/v.json
python tools/trace_segments.py $OUT/trace 20 4 | grep -A12 "per pass" | grep "per pass\|tile_sort\|params\|fold"
rm -rf $OUT/trace
ORDER=sorted REPS=10 VARIANTS="auto:,auto2:" timeout 200 python tools/count_variants.py 2>&1 | grep variant | cut -c1-100
