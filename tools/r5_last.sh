#!/bin/bash
# after the profiles were stamped: the find pipeline's counters (generated order) and the plain bench line that quotes the stamped traffic
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum;SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" MODE=random timeout 400 bash tools/pmc_find.sh > gpurun_out/pmc_find.log 2>&1
cp gpurun_out/pmc_find/summary.txt gpurun_out/pmc_find_random.txt; grep -c "" gpurun_out/pmc_find_random.txt
timeout 900 python bench.py --steps 20 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/bench.json
