#!/bin/bash
# The kernel list (rocprofv3 --kernel-trace --stats) of the heaviest rank's share of configs[3] for each world size in WORLDS
# (default "1 8"), shuffled input only (PLAIN_ONLY) or with SORTED=1 the sorted leg too.  -> gpurun_out/share_kernels_<world>.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in ${WORLDS:-1 8}; do
  out=prof_share_$w
  if [ "${SORTED:-0}" = "1" ]; then export PLAIN_ONLY=0; else export PLAIN_ONLY=1; fi
  (cd /tmp && WORLDS=$w rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/$out -o s --output-format csv -- python $REPO/tools/rank_share.py > /dev/null 2> $REPO/gpurun_out/$out.err)
  echo "--- world $w"
  python - "$REPO/gpurun_out/$out" "$w" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
keep = [r for r in rows if any(k in r["Name"] for k in ("bm_", "bd_", "bw_", "bs_", "bo_", "fx_", "sl_", "ivl_local"))]
with open("gpurun_out/share_kernels_%s.txt" % sys.argv[2], "w") as o:
    for r in keep[:24]:
        line = "%-72s calls=%-5s avg=%9.1f us" % (r["Name"].split("(")[0][-72:], r["Calls"], float(r["AverageNs"]) / 1e3)
        print(line); o.write(line + "\n")
PY
  rm -rf $REPO/gpurun_out/$out
done
