#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3u
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_intervals.py -m gpu -q --timeout 500 -p no:cacheprovider -x -k "padded_runs or dense_pass or bitmap_pass_differential" 2>&1 | tail -3
cd /tmp
VARIANTS="packed:ivl.bd_pad=0,ring2:ivl.bd_depth=2,ring3:ivl.bd_depth=3,ring4:ivl.bd_depth=4,ring6:ivl.bd_depth=6,ring2b:ivl.bd_depth=2" timeout 200 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/v.json 2> $OUT/trace.err
cd $REPO
cut -c1-100 $OUT/v.json
python tools/trace_segments.py $OUT/trace 20 4 | grep -A10 "per pass" | grep "per pass\|bd_search\|tile_sort\|bd_unperm"
rm -rf $OUT/trace
