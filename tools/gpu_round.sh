#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, GPU parity tests, a short bench.  Logs -> gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "=== rocminfo ==="; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -8
  echo "=== smoke ==="; timeout 600 python -c "import __graft_entry__ as g; g.smoke()"; echo "smoke rc=$?"
} > gpurun_out/smoke.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_intervals.py -m gpu -q --timeout 900 -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/test_intervals.log 2>&1; echo "intervals rc=$?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests/test_gpu_bitset.py -m gpu -q --timeout 900 -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/test_bitset.log 2>&1; echo "bitset rc=$?" >> gpurun_out/smoke.log
if [ -f tests/test_gpu_cli.py ]; then
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/test_cli.log 2>&1; echo "cli rc=$?" >> gpurun_out/smoke.log
fi
if [ -f tests/test_gpu_builders_quicksect.py ]; then
timeout 600 python -m pytest tests/test_gpu_builders_quicksect.py -m gpu -q --timeout 500 -p no:cacheprovider > gpurun_out/test_builders.log 2>&1; echo "builders rc=$?" >> gpurun_out/smoke.log
fi
if [ -f tests/test_gpu_operations.py ]; then
timeout 900 python -m pytest tests/test_gpu_operations.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/test_operations.log 2>&1; echo "operations rc=$?" >> gpurun_out/smoke.log
fi
timeout 900 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/smoke.log
tail -6 gpurun_out/smoke.log; tail -3 gpurun_out/test_intervals.log; tail -3 gpurun_out/test_bitset.log; cat gpurun_out/bench.json
