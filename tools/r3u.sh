#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3u
mkdir -p $OUT
export TMPDIR=/tmp
ORDER=sorted REPS=10 VARIANTS="base:,loop:ivl.lc_loop=1,base2:,loop2:ivl.lc_loop=1" timeout 100 python tools/count_variants.py 2>&1 | grep variant | cut -c1-90
cd /tmp
REPS=10 VARIANTS="base:,loop:ivl.lc_loop=1" timeout 200 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/v.json 2> $OUT/trace.err
cd $REPO
cut -c1-100 $OUT/v.json
python tools/trace_segments.py $OUT/trace 20 4 | grep -A12 "per pass" | grep "per pass\|ivl_local\|sorted_check\|plan\|transpose\|fold\|params"
rm -rf $OUT/trace
