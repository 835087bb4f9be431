#!/bin/bash
# SQ counters of the clustered leg's kernels (tools/bench_clustered.py); one counter group per rocprofv3 run
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/pmc_cl
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  REPS=3 timeout 300 rocprofv3 --pmc $grp -d $OUT/run$i -o p --output-format csv -- python $REPO/tools/bench_clustered.py > $OUT/run$i.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob('$OUT/run$i/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    kn = r['Kernel_Name']
    if any(k in kn for k in ("bd_search", "bm_tile_sort", "bd_unpermute")):
        acc[kn.split('(')[0].replace('void ', '').replace('bxmi::', '')[:40] + ' ' + r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    big = [x for x in v if x > 0.5 * max(v)]  # (the launches of the sorted leg stand down: near-zero counters)
    print('%-70s mean_of_working_launches=%.6g n=%d of %d' % (k, sum(big) / len(big), len(big), len(v)))
PY
done
rm -rf $OUT
