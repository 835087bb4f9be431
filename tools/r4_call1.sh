#!/bin/bash
# round 4, GPU call 1: SQ counters of the current search kernel, the kernels of a pass, and what the phases of the tile sort / the search cost
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r4a
mkdir -p $OUT
export TMPDIR=/tmp
VARIANTS="flat:" timeout 900 bash tools/pmc_flat.sh r4a/sq > $OUT/pmc_flat.log 2>&1
cp bx-python_amd/bxmi/libbxmi.so /tmp/lib_default.so
cd /tmp
VARIANTS="base:,nolook:ivl.bd_exp=1,norecs:ivl.bd_exp=3,w8off:ivl.bd_w8=0" REPS=6 timeout 400 rocprofv3 --kernel-trace -d $OUT/tr_base -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/variants_base.json 2> $OUT/variants_base.err
python $REPO/tools/trace_segments.py $OUT/tr_base 20 3 > $OUT/segments_base.txt 2>&1
rm -rf $OUT/tr_base
for v in ts1 ts2 ts3; do
  cp $REPO/build_variants/libbxmi_$v.so $REPO/bx-python_amd/bxmi/libbxmi.so
  VARIANTS="$v:" REPS=6 timeout 300 rocprofv3 --kernel-trace -d $OUT/tr_$v -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/variants_$v.json 2> $OUT/variants_$v.err
  python $REPO/tools/trace_segments.py $OUT/tr_$v 20 3 > $OUT/segments_$v.txt 2>&1
  rm -rf $OUT/tr_$v
done
cp /tmp/lib_default.so $REPO/bx-python_amd/bxmi/libbxmi.so
cd $REPO
tail -30 $OUT/sq/pmc_summary.txt 2>/dev/null
grep -h "tile_sort\|bd_search\|unpermute\|variant" $OUT/segments_*.txt | head -60
