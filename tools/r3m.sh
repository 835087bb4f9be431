#!/bin/bash
# round 3, GPU call M: key slices through the flat walk; the clustered workload; C-ABI all-reduce
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3m
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 900 -p no:cacheprovider \
  -k "bitmap_pass_differential or dense or random_differential or sorted or beyond_16 or count_multi or scale_1M or refused or allreduce or genome_sharded or find_through" > $OUT/tests.log 2>&1
echo "tests rc=$?" | tee -a $OUT/tests.log
tail -5 $OUT/tests.log
cd /tmp
export VARIANTS="default:,sl_flat:ivl.slice=1,sl_old:ivl.slice=1+ivl.sl_flat=0"
timeout 600 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/variants_traced.json 2> $OUT/trace.err; echo "trace rc=$?"
cut -c1-220 $OUT/variants_traced.json | grep "variant"
cd $REPO
python tools/trace_segments.py $OUT/trace 20 4 > $OUT/segments.txt 2>&1
grep -A9 "per pass" $OUT/segments.txt | grep -v "rs_\|ivl_un\|ivl_make\|scan_\|part_b\|rocprim\|at::\|tree_level\|copyBuffer" | head -60
rm -rf $OUT/trace
for order in clustered; do
cd /tmp
ORDER=$order VARIANTS="default:,no_sorted_path:ivl.sorted_path=0" timeout 600 rocprofv3 --kernel-trace -d $OUT/trace_$order -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/variants_$order.json 2> $OUT/trace_$order.err; echo "trace rc=$?"
cut -c1-220 $OUT/variants_$order.json | grep "variant\|order"
cd $REPO
python tools/trace_segments.py $OUT/trace_$order 20 4 > $OUT/segments_$order.txt 2>&1
grep -A9 "per pass" $OUT/segments_$order.txt | grep -v "rs_\|ivl_un\|ivl_make\|scan_\|part_b\|rocprim\|at::\|tree_level\|copyBuffer" | head -40
rm -rf $OUT/trace_$order
done
