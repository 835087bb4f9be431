#!/bin/bash
# A/B of library builds on tools/bench_find.py: LIBS, MODE (sorted|random)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp bx-python_amd/bxmi/libbxmi.so /tmp/lib_default.so
for v in ${LIBS:-default}; do
  if [ $v = default ]; then cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so; else cp build_variants/libbxmi_$v.so bx-python_amd/bxmi/libbxmi.so; fi
  echo "=== lib $v"; MODE=${MODE:-sorted} timeout 200 python tools/bench_find.py 2>/dev/null | cut -c90-260
done
cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so
