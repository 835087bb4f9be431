#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3w
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
WORLDS=8 timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/share -o s --output-format csv -- python $REPO/tools/rank_share.py > $OUT/share.json 2> $OUT/share.err
cat $OUT/share.json
python - <<PY
import csv,glob
f=glob.glob('$OUT/share/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:16]:
    print("%-64s calls=%-4s avg=%9.1f us" % (r['Name'].split('(')[0][-64:], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf $OUT/share
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/find -o f --output-format csv -- python $REPO/tools/bench_find.py > $OUT/find.json 2> $OUT/find.err
tail -3 $OUT/find.json | cut -c1-400
python - <<PY
import csv,glob
f=glob.glob('$OUT/find/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:22]:
    print("%-64s calls=%-4s avg=%9.1f us" % (r['Name'].split('(')[0][-64:], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf $OUT/find
