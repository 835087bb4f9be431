#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 500 -p no:cacheprovider -k "find_through_the_exchange or find_join_scale or find_on_sorted" > gpurun_out/t_find.log 2>&1
echo "find tests rc=$?"; tail -4 gpurun_out/t_find.log | cut -c1-400
grep -q passed gpurun_out/t_find.log && ! grep -q failed gpurun_out/t_find.log || exit 0
MODE=random timeout 200 python tools/bench_find.py 2>/dev/null | cut -c1-330
for fu in 1 0; do
  echo "fused=$fu"; BXMI_OPTS="ivl.find_fused=$fu" MODE=sorted timeout 200 python tools/bench_find.py 2>/dev/null | cut -c1-330
done
trace() {
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_find -o f --output-format csv -- python $REPO/tools/bench_find.py > /dev/null 2>&1
  cd $REPO
  python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_find/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0]))):
    n=r['Name']
    if any(k in n for k in ('bxmi::','scan_')) and not any(k in n for k in ('rs_','ivl_unpack','ivl_make','seal')):
        print("%-70s calls=%-4s avg=%9.1f us" % (n.split('(')[0].replace('void ','').replace('bxmi::','')[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
  rm -rf gpurun_out/prof_find
}
echo "--- trace random"; MODE=random trace | tee gpurun_out/find_kernels.txt
echo "--- trace sorted fused"; MODE=sorted trace | tee gpurun_out/find_sorted_kernels.txt
echo "--- trace sorted unfused"; BXMI_OPTS="ivl.find_fused=0" MODE=sorted trace | tee gpurun_out/find_sorted_unfused_kernels.txt
