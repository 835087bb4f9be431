cd /tmp; export TMPDIR=/tmp
MODE=sorted REPS=3 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_sorted -o s --output-format csv -- python /root/repo/tools/count_only.py > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/root/repo/gpurun_out/prof_sorted/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print("%-60s calls=%-4s avg=%9.1f us" % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf /root/repo/gpurun_out/prof_sorted
