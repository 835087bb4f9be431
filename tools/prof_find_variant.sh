#!/bin/bash
# kernel stats of tools/bench_find.py under each variant of build_variants/ named in VARIANTS (diagnostic builds may give wrong results)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp bx-python_amd/bxmi/libbxmi.so /tmp/lib_default.so
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so; else cp build_variants/libbxmi_$v.so bx-python_amd/bxmi/libbxmi.so; fi
  echo "== $v"; bash tools/prof_find.sh 2>&1 | grep -E "fx_|bd_search|bd_transpose|bd_plan|sl_search|fx_fill|fx_hits|unpermute|tile_sort|ivl_local|part_fill|lf_off|fx_tile"
done
cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so
