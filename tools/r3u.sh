#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3u
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/bits -o b --output-format csv -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sorted --no-find --no-genome > $OUT/b.json 2> $OUT/b.err
python - <<PY
import csv,glob
f=glob.glob('$OUT/bits/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0]))):
    if 'bits_' in r['Name'] or 'tags_' in r['Name']: print("%-64s calls=%-5s avg=%9.1f us min=%7.1f max=%7.1f" % (r['Name'].split('(')[0][-64:], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
rm -rf $OUT/bits
