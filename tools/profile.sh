#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats and, in SEPARATE passes,
# the two HBM PMC counters, all on the same short bench command.  Output -> gpurun_out/prof_*.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps ${BENCH_STEPS:-5} --warmup 2 --no-cpu-baseline --no-sorted --no-find --no-bitset --no-genome --no-pcie"
export PROFILE_CMD="${CMD/$REPO\//}"
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_stats -o stats --output-format csv -- $CMD > $REPO/gpurun_out/prof_stats.json 2> $REPO/gpurun_out/prof_stats.err
echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE -d $REPO/gpurun_out/prof_fetch -o fetch --output-format csv -- $CMD > /dev/null 2> $REPO/gpurun_out/prof_fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE -d $REPO/gpurun_out/prof_write -o write --output-format csv -- $CMD > /dev/null 2> $REPO/gpurun_out/prof_write.err
echo "write rc=$?"
cd $REPO
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -type f | head -30
# keep the merge-back small: drop the huge per-dispatch traces of torch helper kernels
python tools/summarize_profile.py gpurun_out > gpurun_out/profile_summary.txt 2>&1
tail -40 gpurun_out/profile_summary.txt
