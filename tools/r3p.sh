#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3p
mkdir -p $OUT
python tools/rank_share.py 2>/dev/null | tail -1 | tee $OUT/rank_share.json
BXMI_OPTS=ivl.sl_flat=0 python tools/rank_share.py 2>/dev/null | tail -1 | tee $OUT/rank_share_old.json
cd /tmp
WORLDS=8 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python $REPO/tools/rank_share.py > /dev/null 2> $OUT/trace.err
cd $REPO
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3p/trace/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:16]:
    print("%-60s calls=%-4s avg=%9.1f us" % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf $OUT/trace
