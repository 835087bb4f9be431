#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/r2e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_intervals.py -m gpu -x -q --timeout 600 -p no:cacheprovider -k "scale_1M" > $OUT/pytest_fast.log 2>&1
echo "pytest_fast rc=$?" > $OUT/info.log
ORDERS=generated CONFIGS=0:2:1:1,0:2:1:3,0:4:1:1,0:4:1:3,2:4:1:3,2:2:1:3 REPS=10 timeout 600 python tools/bm_perf.py > $OUT/perf.jsonl 2> $OUT/perf.err
cat $OUT/info.log; tail -15 $OUT/pytest_fast.log; cat $OUT/perf.jsonl; tail -3 $OUT/perf.err
