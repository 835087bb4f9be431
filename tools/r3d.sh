#!/bin/bash
# round 3, GPU call D: dense search v3 (batched lookups) -- tests, timing, SQ + HBM counters of the walk
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/r3d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 900 -p no:cacheprovider \
  -k "bitmap_pass_differential or dense or random_differential or beyond_16 or count_multi" > $OUT/tests.log 2>&1
echo "tests rc=$?" | tee -a $OUT/tests.log
tail -5 $OUT/tests.log
export VARIANTS="dense:,dense_nolook:ivl.bd_exp=1,blocks:ivl.bd_blocks=1+ivl.bd_unit_log2=19,u18_nolook:ivl.bd_unit_log2=18+ivl.bd_exp=1"
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/variants_traced.json 2> $OUT/trace.err; echo "trace rc=$?"
cut -c1-200 $OUT/variants_traced.json
cd $REPO
python tools/trace_segments.py $OUT/trace 20 4 > $OUT/segments.txt 2>&1
grep -A5 "^segment.*per pass" $OUT/segments.txt | grep "bd_search\|bd_unperm\|tile_sort\|segment" | head -40
find $OUT/trace -name "*.csv" -size +20M -delete
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
FETCH_SIZE
WRITE_SIZE
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
GROUPS
cd $REPO
python - <<'PY'
import csv, glob, collections
out = open('gpurun_out/r3d/pmc_summary.txt', 'w')
for l in open('gpurun_out/r3d/pmc_index.txt'):
    run = l.split()[0]
    f = glob.glob('gpurun_out/r3d/%s/**/*counter_collection.csv' % run, recursive=True)
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            kn = r['Kernel_Name']
            if 'bd_search' in kn or 'bd_unperm' in kn or 'tile_sort' in kn:
                acc[kn.split('(')[0].replace('void ', '').replace('bxmi::', '')[:32] + ' ' + r['Counter_Name']].append(float(r['Counter_Value']))
    out.write(l.strip() + '\n')
    for k, v in sorted(acc.items()):
        out.write('    %-64s mean=%.6g n=%d\n' % (k, sum(v) / len(v), len(v)))
    if not f:
        out.write('    (no counter file) ' + open('gpurun_out/r3d/%s.log' % run).read()[-300:].replace('\n', ' | ') + '\n')
out.close()
print(open('gpurun_out/r3d/pmc_summary.txt').read())
PY
rm -rf gpurun_out/r3d/pmc*/
