#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3r
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_intervals.py -m gpu -q --timeout 600 -p no:cacheprovider -k "before_with_reversed or neighbours" 2>&1 | tail -2
for o in "" "ivl.sl_f=4" "ivl.sl_f=3" "ivl.sl_f=2" "ivl.sl_f=1"; do
  echo -n "opts=[$o] "; BXMI_OPTS=$o python tools/bench_find.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms'], d['every_hit_overlaps'], d['counts_match_count_path'], d['hits_in_tree_order'])"
done
cd /tmp
BXMI_OPTS="" timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python $REPO/tools/bench_find.py > /dev/null 2> $OUT/trace.err
cd $REPO
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3r/trace/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print("%-64s calls=%-4s avg=%9.1f us" % (r['Name'].split('(')[0][-64:], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf $OUT/trace
