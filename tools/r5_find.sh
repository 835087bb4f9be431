#!/bin/bash
# Round 5, GPU call 1: the find() pipeline of find_exchange.hpp -- parity first, then times, then the kernel trace.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 800 -p no:cacheprovider \
  -k "find_through_the_exchange or find_join_scale or find_on_sorted or count_and_find_match" > gpurun_out/t_find.log 2>&1
echo "find tests rc=$?"; tail -15 gpurun_out/t_find.log | cut -c1-400
for fx in 1 0; do
  BXMI_OPTS="ivl.fx_fill=$fx" MODE=random timeout 300 python tools/bench_find.py > gpurun_out/find_random_fx$fx.json 2> gpurun_out/find_random_fx$fx.err
  echo "fx=$fx rc=$?"; cat gpurun_out/find_random_fx$fx.json | cut -c1-600; tail -3 gpurun_out/find_random_fx$fx.err
done
MODE=sorted timeout 300 python tools/bench_find.py > gpurun_out/find_sorted.json 2>&1; cut -c1-400 gpurun_out/find_sorted.json
cd /tmp
MODE=random timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_find -o f --output-format csv -- python $REPO/tools/bench_find.py > /dev/null 2>&1
cd $REPO
python - <<'PY' | tee gpurun_out/find_kernels.txt
import csv,glob
f=glob.glob('gpurun_out/prof_find/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:24]:
    print("%-70s calls=%-4s avg=%9.1f us" % (r['Name'].split('(')[0].replace('void ','').replace('bxmi::','')[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf gpurun_out/prof_find
timeout 1200 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "cfg5_full_size" > gpurun_out/t_cfg5.log 2>&1
echo "cfg5 golden rc=$?"; tail -5 gpurun_out/t_cfg5.log | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_bitset.py -m gpu -q -x --timeout 800 -p no:cacheprovider -k "cfg3 or gated" > gpurun_out/t_bits.log 2>&1
echo "bitset rc=$?"; tail -5 gpurun_out/t_bits.log | cut -c1-400
