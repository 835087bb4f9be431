#!/bin/bash
# The round's closing evidence on one box (via gpurun): the bench line, the count pass under rocprofv3 (stats + the two HBM counters),
# the clustered leg's kernels and SQ counters in both layouts, find's kernels, the fuzz.  PARTS selects (default: all).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
PARTS=${PARTS:-bench profile clustered find share bits fuzz}
for p in $PARTS; do
  case $p in
    bench) timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/bench.json;;
    profile) bash tools/profile.sh > gpurun_out/profile.log 2>&1; tail -25 gpurun_out/profile_summary.txt;;
    clustered) bash tools/clustered_kernels.sh > gpurun_out/clustered_kernels.txt 2>&1; cat gpurun_out/clustered_kernels.txt gpurun_out/clustered.json
               bash tools/pmc_clustered.sh > gpurun_out/pmc_clustered.log 2>&1; cat gpurun_out/clustered_sq.txt;;
    find) for m in random sorted; do echo "--- $m order" ; MODE=$m bash tools/find_kernels.sh 2>&1 | head -14; done > gpurun_out/find_kernels.txt; cat gpurun_out/find_kernels.txt
          PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" MODE=random bash tools/pmc_find.sh > gpurun_out/pmc_find.log 2>&1; cp gpurun_out/pmc_find/summary.txt gpurun_out/find_pmc.txt 2>/dev/null; tail -30 gpurun_out/find_pmc.txt;;
    share) WORLDS=1,2,4,8 timeout 900 python tools/rank_share.py 2> gpurun_out/rank_share.err | grep '^{' > gpurun_out/rank_share.json; cat gpurun_out/rank_share.json
           WORLDS="1 8" bash tools/share_kernels.sh;;
    bits) bash tools/prof_bits.sh > gpurun_out/prof_bits.log 2>&1; tail -30 gpurun_out/bits_prof.txt;;
    fuzz) : > gpurun_out/fuzz.txt
          for seed in ${SEEDS:-1 2 3 4 5 6 7}; do SEED=$seed ROUNDS=${ROUNDS:-50} timeout 900 python tools/fuzz_intervals.py 2>&1 | tail -2 | tee -a gpurun_out/fuzz.txt; done;;
    suite) timeout 1700 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/test_all.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/test_all.log;;
  esac
done
