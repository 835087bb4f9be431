#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
BENCH_STEPS=20 timeout 1500 bash tools/gpu_round.sh > gpurun_out/round.log 2>&1
tail -14 gpurun_out/round.log | cut -c1-6000
WORLDS=1,8 timeout 300 python tools/rank_share.py > gpurun_out/share_all.json 2> gpurun_out/share_all.err; cat gpurun_out/share_all.json; tail -2 gpurun_out/share_all.err
