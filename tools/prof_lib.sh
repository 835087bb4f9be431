#!/bin/bash
# rocprofv3 kernel stats of the count pass with a library variant from build_variants/ (V=name)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
[ -n "${V:-}" ] && cp build_variants/libbxmi_$V.so bx-python_amd/bxmi/libbxmi.so
cd /tmp; export TMPDIR=/tmp
REPS=3 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_v -o v --output-format csv -- python /root/repo/tools/count_only.py > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/root/repo/gpurun_out/prof_v/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0]))):
    if 'part_' in r['Name']: print("%-40s calls=%-4s avg=%9.1f us" % (r['Name'].split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf /root/repo/gpurun_out/prof_v
