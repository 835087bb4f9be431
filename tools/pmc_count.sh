#!/bin/bash
# PMC passes over the count kernel only (tools/count_only.py); one counter group per rocprofv3 run.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $REPO/gpurun_out/pmc/counters.txt 2>&1
i=0
for mode in ${MODES:-random sorted}; do
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" ; do
  i=$((i+1))
  MODE=$mode REPS=2 rocprofv3 --pmc $grp -d $REPO/gpurun_out/pmc/run$i -o p --output-format csv -- python $REPO/tools/count_only.py > $REPO/gpurun_out/pmc/run$i.log 2>&1
  echo "run$i mode=$mode [$grp] rc=$?" >> $REPO/gpurun_out/pmc/index.txt
done
done
cd $REPO
python - <<'PY'
import csv, glob, os, collections
idx = [l.split() for l in open('gpurun_out/pmc/index.txt')]
out = open('gpurun_out/pmc/summary.txt', 'w')
for l in open('gpurun_out/pmc/index.txt'):
    run = l.split()[0]
    f = glob.glob('gpurun_out/pmc/%s/**/*counter_collection.csv' % run, recursive=True)
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            kn = r['Kernel_Name']
            if any(f in kn for f in os.environ.get('KFILTER', 'ivl_count_kernel').split(',')):
                acc[kn.split('(')[0].replace('bxmi::','')[:24] + ' ' + r['Counter_Name']].append(float(r['Counter_Value']))
    out.write(l.strip() + '\n')
    for k, v in sorted(acc.items()):
        out.write('    %-52s mean=%.6g n=%d\n' % (k, sum(v) / len(v), len(v)))
    if not f:
        out.write('    (no counter file) ' + open('gpurun_out/pmc/%s.log' % run).read()[-300:].replace('\n', ' | ') + '\n')
out.close()
print(open('gpurun_out/pmc/summary.txt').read())
PY
rm -rf gpurun_out/pmc/run*/  # keep the merge-back small
