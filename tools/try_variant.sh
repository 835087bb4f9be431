#!/bin/bash
# a variant library (tools/build_variant.sh) in place of the default: TESTS (-k expression) of tests/test_gpu_intervals.py, then the kernel stats of tools/bench_find.py
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
cp bx-python_amd/bxmi/libbxmi.so /tmp/lib_default.so
cp build_variants/libbxmi_$1.so bx-python_amd/bxmi/libbxmi.so
timeout 900 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 800 -p no:cacheprovider -k "${TESTS:-sorted}" > gpurun_out/t_variant.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|error" gpurun_out/t_variant.log | tail -3 | cut -c1-300
bash tools/prof_find.sh 2>&1 | grep -E "bd_search|sl_search|fx_fill|fx_hits|unpermute|tile_sort|ivl_local|part_fill|lf_off|fx_tile|fx_off"
cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so
