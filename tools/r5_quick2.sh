#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 500 -p no:cacheprovider -k "find_through_the_exchange or find_on_sorted" > gpurun_out/t_find.log 2>&1
echo "find tests rc=$?"; tail -4 gpurun_out/t_find.log | cut -c1-600
cp bx-python_amd/bxmi/libbxmi.so /tmp/lib_default.so
for v in default pfc0 pfc8; do
  if [ $v = default ]; then cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so; else cp build_variants/libbxmi_$v.so bx-python_amd/bxmi/libbxmi.so; fi
  echo -n "$v: "; MODE=sorted timeout 200 python tools/bench_find.py 2>/dev/null | cut -c100-200
done
cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so
