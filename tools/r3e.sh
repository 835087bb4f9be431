#!/bin/bash
# round 3, GPU call E: pipeline depth x lookup style matrix of the dense search kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3e
mkdir -p $OUT
export TMPDIR=/tmp
V=""
for d in 2 3 4 6; do for e in 1 2 0; do V="$V,d${d}e${e}:ivl.bd_depth=${d}+ivl.bd_exp=${e}"; done; done
export VARIANTS="${V:1}"
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/variants_traced.json 2> $OUT/trace.err; echo "trace rc=$?"
cut -c1-130 $OUT/variants_traced.json | grep variant
cd $REPO
python tools/trace_segments.py $OUT/trace 20 4 > $OUT/segments.txt 2>&1
grep "bd_search.*calls=5" $OUT/segments.txt
find $OUT/trace -name "*.csv" -size +20M -delete
