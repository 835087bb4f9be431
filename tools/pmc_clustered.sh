#!/bin/bash
# SQ counters of the clustered leg's search kernel (tools/bench_clustered.py), one counter group per rocprofv3 run, for BOTH
# layouts that can serve a duplicate-heavy index: the dense unit images (ivl.clumped=0: bd_search_kernel, what rounds 4-5 shipped)
# and the offset cells in the clumped layout (default since round 6: bw_search_kernel).  Prints per-QUERY figures of the search
# kernel's working launches (generated order) -> gpurun_out/clustered_sq.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/pmc_cl
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
: > $REPO/gpurun_out/clustered_sq.txt
for layout in "ivl.clumped=0" "ivl.clumped=-1"; do
  i=0
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    BXMI_OPTS=$layout REPS=3 timeout 300 rocprofv3 --pmc $grp -d $OUT/run$i -o p --output-format csv -- python $REPO/tools/bench_clustered.py > $OUT/run$i.log 2>&1
  done
  LAYOUT=$layout python - <<PY | tee -a $REPO/gpurun_out/clustered_sq.txt
import csv, glob, collections, os
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/run*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r['Kernel_Name']
        if any(k in kn for k in ("bd_search", "bw_search")):
            acc[(kn.split('(')[0].replace('void ', '').replace('bxmi::', '')[:44], r['Counter_Name'])].append(float(r['Counter_Value']))
print("== %s" % os.environ["LAYOUT"])
kernels = sorted({k for k, _ in acc})
NQ = 100e6
for kn in kernels:
    m = {}
    for (k, c), v in acc.items():
        if k != kn: continue
        big = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v  # (the launches of the sorted leg stand down: near-zero counters)
        m[c] = sum(big) / len(big)
    g = lambda c: m.get(c, float('nan'))
    print("%-46s VALU instructions per query %.1f | LDS instructions per query %.1f | LDS bank-conflict share of LDS cycles %.1f %% | waves waiting %.1f %% of wave cycles"
          % (kn, g('SQ_INSTS_VALU') * 64 / NQ, g('SQ_INSTS_LDS') * 64 / NQ, 100 * g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1), 100 * g('SQ_WAIT_ANY') / max(g('SQ_WAVE_CYCLES'), 1)))
    print("    raw: " + ", ".join("%s=%.4g" % (c, m[c]) for c in sorted(m)))
PY
  grep -h '^{' $OUT/run1.log | tail -1 | cut -c1-400 | tee -a $REPO/gpurun_out/clustered_sq.txt
  rm -rf $OUT/run*
done
rm -rf $OUT
