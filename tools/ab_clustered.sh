#!/bin/bash
# tools/bench_clustered.py under each variant library of VARIANTS (tools/build_variant.sh)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp bx-python_amd/bxmi/libbxmi.so /tmp/lib_default.so
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so; else cp build_variants/libbxmi_$v.so bx-python_amd/bxmi/libbxmi.so; fi
  echo -n "$v: "; REPS=5 timeout 200 python tools/bench_clustered.py 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['generated_order']['ms'], d['generated_order']['same_as_direct_kernel'], d['sorted_by_start']['ms'], d['search_stage_of_this_index'])"
done
cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so
