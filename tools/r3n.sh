#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3n
mkdir -p $OUT
export TMPDIR=/tmp
ORDER=sorted VARIANTS="loop4:,one_chunk:ivl.lc_loop=0" python tools/count_variants.py 2>&1 | cut -c1-120 | grep variant
cd /tmp
VARIANTS="loop4:,one_chunk:ivl.lc_loop=0" timeout 600 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/v.json 2> $OUT/trace.err
cd $REPO
cut -c1-120 $OUT/v.json | grep variant
python tools/trace_segments.py $OUT/trace 20 4 | grep "ivl_local_count.*calls=5\|per pass"
rm -rf $OUT/trace
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-find --no-bitset --no-sorted 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('count ms', d['ms_per_step'], 'genome', d['genome']['ms_per_step'], d['genome']['kernel_ms_slowest_rank'], d['genome']['parity'])
"
BXMI_OPTS=ivl.sl_flat=0 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-find --no-bitset --no-sorted 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sl_flat=0: genome', d['genome']['ms_per_step'], d['genome']['kernel_ms_slowest_rank'])
"
