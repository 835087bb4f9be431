#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/r2d
mkdir -p $OUT
export TMPDIR=/tmp
ORDERS=generated EXP_PAIR=0 EXPS=1,2,3,4,5,6,7 CONFIGS=0:4:0,0:8:0 REPS=5 timeout 600 python tools/bm_perf.py > $OUT/perf_nopair.jsonl 2> $OUT/perf.err
ORDERS=generated EXP_PAIR=1 EXPS=3,4,7 CONFIGS=0:8:1 REPS=5 timeout 600 python tools/bm_perf.py > $OUT/perf_pair.jsonl 2>> $OUT/perf.err
cat $OUT/perf_nopair.jsonl $OUT/perf_pair.jsonl; tail -3 $OUT/perf.err
