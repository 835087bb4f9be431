#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_intervals.py -m gpu -q --timeout 900 -p no:cacheprovider \
  -k "bitmap_pass_differential or dense or random_differential or beyond_16 or clustered" > $OUT/tests.log 2>&1
echo "tests rc=$?" | tee -a $OUT/tests.log
grep -n "^E  \|passed\|failed" $OUT/tests.log | head -20
cd /tmp
ORDER=clustered VARIANTS="default:,slices:ivl.flat=0+ivl.dense=0+ivl.bitmap=-1" timeout 600 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/v.json 2> $OUT/trace.err
cd $REPO
cut -c1-230 $OUT/v.json | grep "variant\|order"
python tools/trace_segments.py $OUT/trace 20 4 | grep -A4 "per pass" | grep -v "rs_\|ivl_un\|ivl_make\|scan_\|part_b\|rocprim\|at::\|tree_level\|copyBuffer" | head -40
rm -rf $OUT/trace
VARIANTS="dense18:ivl.flat=0,dense19:ivl.flat=0+ivl.bd_unit_log2=19,flat:" python tools/count_variants.py 2>/dev/null | cut -c1-200 | grep variant
