#!/bin/bash
# One GPU call for the end of round 5: smoke + every GPU test + the bench line (tools/gpu_round.sh), the rocprofv3 summaries of the
# count pass (tools/profile.sh), the counters of the find pipeline, the bitset kernels, the 8-GPU rank share.  -> gpurun_out/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
BENCH_STEPS=20 timeout 1500 bash tools/gpu_round.sh > gpurun_out/round.log 2>&1
tail -12 gpurun_out/round.log | cut -c1-300
timeout 400 bash tools/profile.sh > gpurun_out/profile.log 2>&1; tail -3 gpurun_out/profile.log
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum;SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" timeout 400 bash tools/pmc_find.sh > gpurun_out/pmc_find.log 2>&1; grep -c "" gpurun_out/pmc_find/summary.txt
cp gpurun_out/pmc_find/summary.txt gpurun_out/pmc_find_random.txt 2>/dev/null  # (the sorted run below overwrites summary.txt)
MODE=sorted PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" timeout 200 bash tools/pmc_find.sh > /dev/null 2>&1; cp gpurun_out/pmc_find/summary.txt gpurun_out/pmc_find_sorted.txt 2>/dev/null
timeout 300 bash tools/prof_bits.sh > gpurun_out/prof_bits.log 2>&1; tail -3 gpurun_out/bits_prof.txt | cut -c1-200
WORLDS=1,2,4,8 timeout 400 python tools/rank_share.py > gpurun_out/share_all.json 2> gpurun_out/share_all.err; cut -c1-600 gpurun_out/share_all.json
ROUNDS=30 SEED=2025 timeout 400 python tools/fuzz_intervals.py > gpurun_out/fuzz.log 2>&1; tail -1 gpurun_out/fuzz.log
