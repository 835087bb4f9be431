#!/bin/bash
# One GPU iteration on the host-pointer entry points (via gpurun): the PCIe probe, then tools/bench_host.py.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${PROBE:-1}" = "1" ]; then
  /opt/rocm/bin/hipcc -O2 -o /tmp/pcie_probe tools/micro/pcie_probe.hip -lpthread 2>/dev/null && timeout 120 /tmp/pcie_probe | tee gpurun_out/pcie_probe.txt
fi
timeout 900 python tools/bench_host.py 2>&1 | tee gpurun_out/bench_host.txt | grep -v '^{'
