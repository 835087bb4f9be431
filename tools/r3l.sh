#!/bin/bash
# round 3, GPU call J: hand-issued pipelined loads in the flat walk -- tests + timing matrix
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 900 -p no:cacheprovider \
  -k "bitmap_pass_differential or dense or random_differential or sorted or beyond_16 or count_multi or scale_1M or refused" > $OUT/tests.log 2>&1
echo "tests rc=$?" | tee -a $OUT/tests.log
tail -5 $OUT/tests.log
V="flat:,flat_p0:ivl.bd_pipe=0,dense:ivl.flat=0,flat_sorted_path_off:ivl.sorted_path=0"

export VARIANTS="$V"
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/variants_traced.json 2> $OUT/trace.err; echo "trace rc=$?"
cut -c1-150 $OUT/variants_traced.json | grep "variant"
cd $REPO
python tools/trace_segments.py $OUT/trace 20 4 > $OUT/segments.txt 2>&1
grep -A11 "per pass" $OUT/segments.txt | grep -v "rs_\|ivl_un\|ivl_make\|scan_\|part_b\|rocprim\|at::\|tree_level\|copyBuffer"
find $OUT/trace -name "*.csv" -size +20M -delete
