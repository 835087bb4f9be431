#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3t
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_intervals.py -m gpu -q --timeout 900 -p no:cacheprovider -x \
  -k "bitmap_pass_differential or dense or clustered" 2>&1 | tail -15
cd /tmp
VARIANTS="packed:ivl.bd_pad=0,ring4:ivl.bd_depth=4,ring6:ivl.bd_depth=6,ring8:ivl.bd_depth=8,ring2:ivl.bd_depth=2" timeout 600 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/v.json 2> $OUT/trace.err
cd $REPO
cut -c1-130 $OUT/v.json | grep "variant"
python tools/trace_segments.py $OUT/trace 20 4 | grep -A10 "per pass" | grep -v "rs_\|ivl_un\|ivl_make\|scan_\|part_b\|rocprim\|at::\|tree_level\|copyBuffer" | head -60
rm -rf $OUT/trace
