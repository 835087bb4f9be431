#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 bash tools/profile.sh > gpurun_out/profile.log 2>&1; tail -3 gpurun_out/profile.log
