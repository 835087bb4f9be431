#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3g
mkdir -p $OUT
export TMPDIR=/tmp
V=""
for d in 2 4 6 8; do for e in 1 0; do V="$V,flat_d${d}e${e}:ivl.bd_depth=${d}+ivl.bd_exp=${e}"; done; done
V="$V,dense_d4e1:ivl.flat=0+ivl.bd_depth=4+ivl.bd_exp=1,dense_d8e1:ivl.flat=0+ivl.bd_depth=8+ivl.bd_exp=1,dense_d4e2:ivl.flat=0+ivl.bd_depth=4+ivl.bd_exp=2,dense_d8e2:ivl.flat=0+ivl.bd_depth=8+ivl.bd_exp=2,dense_d4e0:ivl.flat=0+ivl.bd_depth=4"
export VARIANTS="${V:1}"
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/variants_traced.json 2> $OUT/trace.err; echo "trace rc=$?"
cut -c1-110 $OUT/variants_traced.json | grep "variant"
cd $REPO
python tools/trace_segments.py $OUT/trace 20 4 > $OUT/segments.txt 2>&1
grep "bd_search.*calls=5" $OUT/segments.txt
find $OUT/trace -name "*.csv" -size +20M -delete
