#!/bin/bash
# round 3, GPU call B: what bounds the dense search kernel -- the walk without lookups at three run lengths, SQ counters.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3b
mkdir -p $OUT
export TMPDIR=/tmp
export VARIANTS="dense:,dense_nolook:ivl.bd_exp=1,u18:ivl.bd_unit_log2=18,u18_nolook:ivl.bd_unit_log2=18+ivl.bd_exp=1,u17_nolook:ivl.bd_unit_log2=17+ivl.bd_exp=1,u18_t16k_nolook:ivl.bd_unit_log2=18+ivl.bd_exp=1+ivl.bm_variant=0,pair:ivl.dense=0"
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/variants_traced.json 2> $OUT/trace.err; echo "trace rc=$?"
cat $OUT/variants_traced.json
cd $REPO
python tools/trace_segments.py $OUT/trace 20 4 > $OUT/segments.txt 2>&1
grep -A4 "^segment.*per pass" $OUT/segments.txt | grep -v "^--" | head -80
find $OUT/trace -name "*.csv" -size +20M -delete
export VARIANTS="dense:"
export REPS=2
cd /tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $OUT/pmc$i -o p --output-format csv -- python $REPO/tools/count_variants.py > $OUT/pmc$i.log 2>&1
  echo "pmc$i [$grp] rc=$?" >> $OUT/pmc_index.txt
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
GROUPS
cd $REPO
python - <<'PY'
import csv, glob, collections
out = open('gpurun_out/r3b/pmc_summary.txt', 'w')
for l in open('gpurun_out/r3b/pmc_index.txt'):
    run = l.split()[0]
    f = glob.glob('gpurun_out/r3b/%s/**/*counter_collection.csv' % run, recursive=True)
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            kn = r['Kernel_Name']
            if 'bm_' in kn or 'bd_' in kn or 'sl_' in kn:
                acc[kn.split('(')[0].replace('void ', '').replace('bxmi::', '')[:28] + ' ' + r['Counter_Name']].append(float(r['Counter_Value']))
    out.write(l.strip() + '\n')
    for k, v in sorted(acc.items()):
        out.write('    %-60s mean=%.6g n=%d\n' % (k, sum(v) / len(v), len(v)))
    if not f:
        out.write('    (no counter file) ' + open('gpurun_out/r3b/%s.log' % run).read()[-300:].replace('\n', ' | ') + '\n')
out.close()
print(open('gpurun_out/r3b/pmc_summary.txt').read())
PY
rm -rf gpurun_out/r3b/pmc*/
