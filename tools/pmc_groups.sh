#!/bin/bash
# PMC counters of the count pass's kernels: tools/count_variants.py (VARIANTS, one is enough) under rocprofv3 --pmc, one run per
# counter group.  OUT=name  LIB=default|variant  PMC_GROUPS="A B;C D"  KERNELS=regex of kernel names to keep
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/${OUT:-pmc_groups}
mkdir -p $OUT
export TMPDIR=/tmp
export REPS=${REPS:-2}
cp bx-python_amd/bxmi/libbxmi.so /tmp/lib_default.so
LIB=${LIB:-default}
if [ $LIB != default ]; then cp build_variants/libbxmi_$LIB.so bx-python_amd/bxmi/libbxmi.so; fi
: > $OUT/index_$LIB.txt
cd /tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $OUT/${LIB}_pmc$i -o p --output-format csv -- python $REPO/tools/count_variants.py > $OUT/${LIB}_pmc$i.log 2>&1
  echo "${LIB}_pmc$i [$grp] rc=$?" >> $OUT/index_$LIB.txt
done < <(echo "${PMC_GROUPS}" | tr ';' '\n')
cd $REPO
cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so
KERNELS="${KERNELS:-tile_sort|_search|unpermute}" python - "$OUT" "$LIB" <<'PY'
import csv, glob, collections, sys, os, re
O, LIB = sys.argv[1], sys.argv[2]
keep = re.compile(os.environ['KERNELS'])
out = open(os.path.join(O, 'summary_%s.txt' % LIB), 'w')
for l in open(os.path.join(O, 'index_%s.txt' % LIB)):
    run = l.split()[0]
    f = glob.glob(os.path.join(O, run, '**', '*counter_collection.csv'), recursive=True)
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            kn = r['Kernel_Name']
            if keep.search(kn):
                acc[kn.split('(')[0].replace('void ', '').replace('bxmi::', '')[:40] + ' ' + r['Counter_Name']].append(float(r['Counter_Value']))
    out.write(l.strip() + '\n')
    for k, v in sorted(acc.items()):
        v = v[1:] if len(v) > 1 else v  # (the first dispatch is the warm-up)
        out.write('    %-72s mean=%.6g n=%d\n' % (k, sum(v) / len(v), len(v)))
    if not f:
        out.write('    (no counter file) ' + open(os.path.join(O, run + '.log')).read()[-300:].replace('\n', ' | ') + '\n')
out.close()
print(open(os.path.join(O, 'summary_%s.txt' % LIB)).read())
PY
rm -rf $OUT/${LIB}_pmc*/
