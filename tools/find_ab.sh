#!/bin/bash
# A/B of the find() pipeline on configs[4] (tools/bench_find.py under rocprofv3 --kernel-trace --stats), one run per
# BXMI_OPTS setting in OPTS_LIST (space separated), then the find tests.  Runs on the GPU box (via gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/find_ab
mkdir -p $OUT
export TMPDIR=/tmp
[ -n "${NO_TESTS:-}" ] || timeout 600 python -m pytest tests/test_gpu_intervals.py -m gpu -q --timeout 500 -p no:cacheprovider -x -k "find" 2>&1 | tail -3
for o in ${OPTS_LIST:-"ivl.fill_pairs=1"}; do
cd /tmp
BXMI_OPTS="$o" timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/find -o f --output-format csv -- python $REPO/tools/bench_find.py > $OUT/find.json 2> $OUT/find.err
echo "== $o"; tail -1 $OUT/find.json | cut -c100-420
python - <<PY
import csv,glob
f=glob.glob('$OUT/find/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0]))):
    if ('sl_' in r['Name'] or 'bm_' in r['Name']) and float(r['AverageNs'])>100e3: print("%-64s calls=%-4s avg=%9.1f us" % (r['Name'].split('(')[0][-64:], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf $OUT/find
done
