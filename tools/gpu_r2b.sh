#!/bin/bash
# Round-2 GPU call B: timing with the diagnostic variants, then PMC counters of the bitmap-cell pass.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r2c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_intervals.py -m gpu -x -q --timeout 600 -p no:cacheprovider -k "bitmap or scale_1M" > $OUT/pytest_fast.log 2>&1
echo "pytest_fast rc=$?" > $OUT/info.log
EXPS=1,2,3 REPS=5 timeout 600 python tools/bm_perf.py > $OUT/perf.jsonl 2> $OUT/perf.err
echo "perf rc=$?" >> $OUT/info.log
cd /tmp
REPS=5 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_only.py > $OUT/trace.log 2>&1
cd $REPO
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -E "Name|bm_|part_" "$f" > $OUT/kernel_stats.csv
rm -rf $OUT/trace
cat $OUT/info.log; tail -5 $OUT/pytest_fast.log; cat $OUT/perf.jsonl; tail -3 $OUT/perf.err; cut -c1-150 $OUT/kernel_stats.csv

