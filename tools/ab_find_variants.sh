#!/bin/bash
# A/B of libbxmi variants (tools/build_variant.sh) on the find pipeline: VARIANTS="default a b", MODE=random|sorted
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
cp bx-python_amd/bxmi/libbxmi.so /tmp/lib_default.so
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so; else cp build_variants/libbxmi_$v.so bx-python_amd/bxmi/libbxmi.so; fi
  for r in 1 2; do echo -n "$v: "; MODE=${MODE:-random} timeout 200 python tools/bench_find.py 2>/dev/null | cut -c95-135; done
done
cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so
