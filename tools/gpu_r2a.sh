#!/bin/bash
# Round-2 GPU call A: parity of the bitmap-cell count pass, its timing matrix, and a kernel trace of the default config.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/r2a
{ /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4; } > $OUT/info.log 2>&1
timeout 900 python -m pytest tests/test_gpu_intervals.py -m gpu -x -q --timeout 600 -p no:cacheprovider \
   -k "bitmap or random_differential or sorted_batches or beyond_16 or dense_bucket or scale_1M or golden or reference_vectors" > $OUT/pytest_fast.log 2>&1
echo "pytest_fast rc=$?" >> $OUT/info.log
NQ=${NQ:-100000000} REPS=5 timeout 900 python tools/bm_perf.py > $OUT/perf.jsonl 2> $OUT/perf.err
echo "perf rc=$?" >> $OUT/info.log
cd /tmp
REPS=5 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_only.py > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/info.log
cd $REPO
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f" > $OUT/kernel_stats.csv
rm -rf $OUT/trace
cat $OUT/info.log; tail -15 $OUT/pytest_fast.log; cat $OUT/perf.jsonl; tail -3 $OUT/perf.err; cat $OUT/kernel_stats.csv | cut -c1-200
