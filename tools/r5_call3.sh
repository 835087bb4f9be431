#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
cp bx-python_amd/bxmi/libbxmi.so /tmp/lib_default.so
for v in fxA fxB fxC fxD; do
  cp build_variants/libbxmi_$v.so bx-python_amd/bxmi/libbxmi.so
  echo "=== $v"; timeout 200 python tools/r5_debug2.py 2>&1 | tail -20
done > gpurun_out/debug_variants.log 2>&1
cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so
cat gpurun_out/debug_variants.log
