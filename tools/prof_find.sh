#!/bin/bash
# rocprofv3 kernel stats of tools/bench_find.py (MODE=random|sorted)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_find -o f --output-format csv -- python /root/repo/tools/bench_find.py > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/root/repo/gpurun_out/prof_find/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:40]:
    print("%-64s calls=%-4s avg=%9.1f us" % (r['Name'].split('(')[0][-64:], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf /root/repo/gpurun_out/prof_find
