#!/bin/bash
# full GPU interval tests + timing (both orders) + kernel trace of the default configuration
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r2f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_intervals.py -m gpu -x -q --timeout 900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" > $OUT/info.log
REPS=10 timeout 600 python tools/bm_perf.py > $OUT/perf.jsonl 2> $OUT/perf.err
cd /tmp
REPS=5 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_only.py > $OUT/trace.log 2>&1
MODE=sorted REPS=5 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_s -o t --output-format csv -- python $REPO/tools/count_only.py > $OUT/trace_s.log 2>&1
cd $REPO
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|bm_|part_|ivl_local" "$f" > $OUT/kernel_stats.csv
f=$(find $OUT/trace_s -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|bm_|part_|ivl_local" "$f" > $OUT/kernel_stats_sorted.csv
rm -rf $OUT/trace $OUT/trace_s
cat $OUT/info.log; tail -8 $OUT/pytest.log; cat $OUT/perf.jsonl; tail -3 $OUT/perf.err
