#!/bin/bash
# offset-cell images: parity tests, the fuzz tool, then the genome share at WORLDS
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/sparse
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 500 -p no:cacheprovider -k "${K:-sparse or count_multi or genome_cfg4}" > gpurun_out/sparse/tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/sparse/tests.log | cut -c1-400
ROUNDS=${ROUNDS:-10} timeout 400 python tools/fuzz_intervals.py > gpurun_out/sparse/fuzz.log 2>&1
echo "fuzz rc=$?"; tail -3 gpurun_out/sparse/fuzz.log | cut -c1-600
WORLDS=${WORLDS:-1,8} timeout 300 python tools/rank_share.py > gpurun_out/sparse/share.json 2> gpurun_out/sparse/share.err
echo "share rc=$?"; cat gpurun_out/sparse/share.json; tail -2 gpurun_out/sparse/share.err
