#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3k
mkdir -p $OUT
export TMPDIR=/tmp
V="flat:,flat_nolook:ivl.bd_exp=1,flat_compute:ivl.bd_exp=3,flat_compute_p0:ivl.bd_exp=3+ivl.bd_pipe=0,flat_compute_d2:ivl.bd_exp=3+ivl.bd_depth=2+ivl.bd_pipe=0"
V="$V,dense:ivl.flat=0,dense_nolook:ivl.flat=0+ivl.bd_exp=1,dense_compute:ivl.flat=0+ivl.bd_exp=3,dense_p0:ivl.flat=0+ivl.bd_pipe=0"
export VARIANTS="$V"
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/variants_traced.json 2> $OUT/trace.err; echo "trace rc=$?"
cut -c1-120 $OUT/variants_traced.json | grep "variant"
cd $REPO
python tools/trace_segments.py $OUT/trace 20 4 > $OUT/segments.txt 2>&1
grep "bd_search.*calls=5" $OUT/segments.txt
find $OUT/trace -name "*.csv" -size +20M -delete
