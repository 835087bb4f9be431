#!/bin/bash
# quick GPU check: interval tests + count-kernel timing for both paths
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/test_intervals.log 2>&1; echo "intervals rc=$?"
tail -15 gpurun_out/test_intervals.log
for opts in "ivl.partition=0" "ivl.partition=1"; do echo -n "$opts  "; BXMI_OPTS=$opts REPS=5 python tools/count_only.py 2>&1 | tail -1; done
for opts in "ivl.sorted_path=0" "ivl.sorted_path=1"; do echo -n "$opts  "; MODE=sorted BXMI_OPTS=$opts REPS=5 python tools/count_only.py 2>&1 | tail -1; done
cd /tmp; export TMPDIR=/tmp
BXMI_OPTS=ivl.partition=1 REPS=3 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_part -o part --output-format csv -- python /root/repo/tools/count_only.py > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/root/repo/gpurun_out/prof_part/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print("%-60s calls=%-4s avg=%9.1f us" % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf /root/repo/gpurun_out/prof_part
