#!/bin/bash
# PMC passes over the find() pipeline (tools/bench_find.py, configs[4]); one counter group per rocprofv3 run.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/pmc_find
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
# PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" limits the passes (one rocprofv3 run of ~25 s each)
if [ -n "${PMC_GROUPS:-}" ]; then exec 3< <(echo "$PMC_GROUPS" | tr ';' '\n'); else exec 3< <(cat <<'GROUPS'
FETCH_SIZE
WRITE_SIZE
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
GROUPS
); fi
while read -r grp <&3; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  MODE=${MODE:-random} timeout 300 rocprofv3 --pmc $grp -d $OUT/run$i -o p --output-format csv -- python $REPO/tools/bench_find.py > $OUT/run$i.log 2>&1
  echo "run$i [$grp] rc=$?" >> $OUT/index.txt
done
cd $REPO
python - <<'PY'
import csv, glob, collections
out = open('gpurun_out/pmc_find/summary.txt', 'w')
for l in open('gpurun_out/pmc_find/index.txt'):
    run = l.split()[0]
    f = glob.glob('gpurun_out/pmc_find/%s/**/*counter_collection.csv' % run, recursive=True)
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            kn = r['Kernel_Name']
            if any(k in kn for k in ("sl_", "bm_", "scan_", "fx_", "ivl_local", "part_fill")):
                acc[kn.split('(')[0].replace('void ', '').replace('bxmi::', '')[:34] + ' ' + r['Counter_Name']].append(float(r['Counter_Value']))
    out.write(l.strip() + '\n')
    for k, v in sorted(acc.items()):
        out.write('    %-66s mean=%.6g n=%d\n' % (k, sum(v) / len(v), len(v)))
out.close()
print(open('gpurun_out/pmc_find/summary.txt').read())
PY
rm -rf gpurun_out/pmc_find/run*/
