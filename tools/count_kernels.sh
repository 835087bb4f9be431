#!/bin/bash
# The kernel list (rocprofv3 --kernel-trace --stats) of tools/count_only.py under the caller's environment (MODE, NQ, NOCOUNTS, BXMI_OPTS ...)
# -> stdout and gpurun_out/count_kernels.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ck -o s --output-format csv -- python $R/tools/count_only.py > /dev/null 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_ck/**/*kernel_stats.csv", recursive=True)
o = open("gpurun_out/count_kernels.txt", "w")
for r in list(csv.DictReader(open(f[0]))):
    if any(k in r["Name"] for k in ("bm_", "bw_", "bd_", "bs_", "sl_", "ivl_local")):
        line = "%-70s calls=%-4s avg=%8.1f us" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3)
        print(line); o.write(line + "\n")
PY
rm -rf gpurun_out/prof_ck
