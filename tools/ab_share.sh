#!/bin/bash
# A/B of library builds on the heaviest rank's share of the genome (tools/rank_share.py): LIBS, WORLDS, OUT; TRACE=1 adds kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/${OUT:-share}
mkdir -p $OUT
export TMPDIR=/tmp
cp bx-python_amd/bxmi/libbxmi.so /tmp/lib_default.so
for v in ${LIBS:-default}; do
  if [ $v = default ]; then cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so; else cp build_variants/libbxmi_$v.so bx-python_amd/bxmi/libbxmi.so; fi
  echo "=== lib $v"
  if [ -n "${TRACE:-}" ]; then
    (cd /tmp && timeout ${TMO:-300} rocprofv3 --kernel-trace --stats -d $OUT/tr_$v -o t --output-format csv -- python $REPO/tools/rank_share.py > $OUT/share_$v.json 2> $OUT/share_$v.err)
    python - $OUT/tr_$v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:9]:
    print("    %-70s calls=%-4s avg=%8.1f us" % (r['Name'].split('(')[0].replace('void ', '').replace('bxmi::', '')[:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
    rm -rf $OUT/tr_$v
  else
    timeout ${TMO:-300} python tools/rank_share.py > $OUT/share_$v.json 2> $OUT/share_$v.err
  fi
  echo "rc=$?"; cat $OUT/share_$v.json; tail -2 $OUT/share_$v.err
done
cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so
