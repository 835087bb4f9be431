#!/bin/bash
# round 3, GPU call C: dense search v2 (16-bit counts out of place, trimmed VALU, prefetch)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 900 -p no:cacheprovider \
  -k "bitmap_pass_differential or dense or random_differential or sorted or beyond_16 or count_multi or scale_1M or refused" > $OUT/tests.log 2>&1
echo "tests rc=$?" | tee -a $OUT/tests.log
tail -5 $OUT/tests.log
export VARIANTS="${VARIANTS:-dense:,dense_nolook:ivl.bd_exp=1,blocks:ivl.bd_blocks=1+ivl.bd_unit_log2=19,u18:ivl.bd_unit_log2=18,u18_nolook:ivl.bd_unit_log2=18+ivl.bd_exp=1,pair:ivl.dense=0}"
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/variants_traced.json 2> $OUT/trace.err; echo "trace rc=$?"
cut -c1-200 $OUT/variants_traced.json
cd $REPO
python tools/trace_segments.py $OUT/trace 20 4 > $OUT/segments.txt 2>&1
grep -A5 "^segment.*per pass" $OUT/segments.txt | grep -v "^--\|rs_scatter\|ivl_unpack\|rs_tile\|rocprim" | head -80
find $OUT/trace -name "*.csv" -size +20M -delete
