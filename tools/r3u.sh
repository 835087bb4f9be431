#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3u
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ORDER=clustered REPS=5 VARIANTS="base:,again:" timeout 200 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/v.json 2> $OUT/trace.err
cd $REPO
cut -c1-100 $OUT/v.json
python tools/trace_segments.py $OUT/trace 20 4 | grep -A10 "per pass" | head -24
rm -rf $OUT/trace
