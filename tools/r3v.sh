#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 bash tools/profile.sh > gpurun_out/profile.log 2>&1
tail -5 gpurun_out/profile.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.json | cut -c1-1500
