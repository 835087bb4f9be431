#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cp bx-python_amd/bxmi/libbxmi.so /tmp/lib_default.so
cp build_variants/libbxmi_peek.so bx-python_amd/bxmi/libbxmi.so
REPS=1 timeout 120 python tools/r5_debug3.py 2>&1 | head -8 | cut -c1-300
cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so
timeout 600 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 500 -p no:cacheprovider -k "find_through_the_exchange or find_join_scale" > gpurun_out/t_find.log 2>&1
echo "find tests rc=$?"; tail -6 gpurun_out/t_find.log | cut -c1-400
grep -q passed gpurun_out/t_find.log && ! grep -q failed gpurun_out/t_find.log || exit 0
for fx in 1 0; do
  BXMI_OPTS="ivl.fx_fill=$fx" MODE=random timeout 200 python tools/bench_find.py > gpurun_out/find_random_fx$fx.json 2> gpurun_out/find_random_fx$fx.err
  echo "fx=$fx rc=$?"; cat gpurun_out/find_random_fx$fx.json | cut -c1-600
done
for fu in 1 0; do
  BXMI_OPTS="ivl.find_fused=$fu" MODE=sorted timeout 200 python tools/bench_find.py > gpurun_out/find_sorted_fused$fu.json 2>/dev/null; echo "fused=$fu"; cut -c1-400 gpurun_out/find_sorted_fused$fu.json
done
cd /tmp
MODE=random timeout 200 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_find -o f --output-format csv -- python $REPO/tools/bench_find.py > /dev/null 2>&1
cd $REPO
python - <<'PY' | tee gpurun_out/find_kernels.txt
import csv,glob
f=glob.glob('gpurun_out/prof_find/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:26]:
    print("%-70s calls=%-4s avg=%9.1f us" % (r['Name'].split('(')[0].replace('void ','').replace('bxmi::','')[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf gpurun_out/prof_find
cd /tmp
MODE=sorted timeout 200 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_find -o f --output-format csv -- python $REPO/tools/bench_find.py > /dev/null 2>&1
cd $REPO
python - <<'PY' | tee gpurun_out/find_sorted_kernels.txt
import csv,glob
f=glob.glob('gpurun_out/prof_find/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print("%-70s calls=%-4s avg=%9.1f us" % (r['Name'].split('(')[0].replace('void ','').replace('bxmi::','')[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf gpurun_out/prof_find
timeout 900 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 800 -p no:cacheprovider -k "cfg5_full_size" > gpurun_out/t_cfg5.log 2>&1
echo "cfg5 golden rc=$?"; tail -3 gpurun_out/t_cfg5.log | cut -c1-300
