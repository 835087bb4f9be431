#!/bin/bash
# SQ / LDS counters of the search kernel of the count pass (tools/count_variants.py, one variant); OUT dir as $1
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/${1:-pmc_flat}
mkdir -p $OUT
export TMPDIR=/tmp
export VARIANTS="${VARIANTS:-flat:}"
export REPS=2
cd /tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $OUT/pmc$i -o p --output-format csv -- python $REPO/tools/count_variants.py > $OUT/pmc$i.log 2>&1
  echo "pmc$i [$grp] rc=$?" >> $OUT/pmc_index.txt
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_SALU
GROUPS
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, collections, sys, os
O = sys.argv[1]
out = open(os.path.join(O, 'pmc_summary.txt'), 'w')
for l in open(os.path.join(O, 'pmc_index.txt')):
    run = l.split()[0]
    f = glob.glob(os.path.join(O, run, '**', '*counter_collection.csv'), recursive=True)
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            kn = r['Kernel_Name']
            if 'bd_search' in kn or 'bw_search' in kn:
                acc[kn.split('(')[0].replace('void ', '').replace('bxmi::', '')[:36] + ' ' + r['Counter_Name']].append(float(r['Counter_Value']))
    out.write(l.strip() + '\n')
    for k, v in sorted(acc.items()):
        out.write('    %-64s mean=%.6g n=%d\n' % (k, sum(v) / len(v), len(v)))
    if not f:
        out.write('    (no counter file) ' + open(os.path.join(O, run + '.log')).read()[-300:].replace('\n', ' | ') + '\n')
out.close()
print(open(os.path.join(O, 'pmc_summary.txt')).read())
PY
rm -rf $OUT/pmc*/
