#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
cp bx-python_amd/bxmi/libbxmi.so /tmp/lib_default.so
cp build_variants/libbxmi_peek.so bx-python_amd/bxmi/libbxmi.so
timeout 300 python tools/r5_debug3.py > gpurun_out/debug3.log 2>&1
cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so
cat gpurun_out/debug3.log | cut -c1-600
