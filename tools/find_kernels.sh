#!/bin/bash
# The kernel list (rocprofv3 --kernel-trace --stats) of tools/bench_find.py under the caller's environment (MODE, BXMI_LIB, BXMI_OPTS ...)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fk -o s --output-format csv -- python $R/tools/bench_find.py > /dev/null 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_fk/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0]))):
    if any(k in r["Name"] for k in ("bm_", "bw_", "bd_", "bs_", "sl_", "fx_", "ivl_", "part_", "lf_", "scan_")) and float(r["AverageNs"]) > 3000:
        print("%-70s calls=%-4s avg=%8.1f us" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/prof_fk
