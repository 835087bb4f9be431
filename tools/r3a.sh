#!/bin/bash
# round 3, GPU call A: the dense-image search stage -- parity tests, the A/B matrix, a kernel trace of it.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 900 -p no:cacheprovider \
  -k "bitmap_pass_differential or dense or random_differential or sorted or beyond_16 or count_multi or scale_1M or refused" > $OUT/tests.log 2>&1
echo "tests rc=$?" | tee -a $OUT/tests.log
tail -5 $OUT/tests.log
timeout 600 python tools/count_variants.py > $OUT/variants.json 2> $OUT/variants.err; echo "variants rc=$?"
cat $OUT/variants.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/variants_traced.json 2> $OUT/trace.err; echo "trace rc=$?"
cd $REPO
python tools/trace_segments.py $OUT/trace > $OUT/segments.txt 2>&1
cat $OUT/segments.txt | head -150
# keep the merge-back small
find $OUT/trace -name "*.csv" -size +20M -delete
