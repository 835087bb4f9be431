#!/bin/bash
# rocprofv3 of the bitset kernels (configs[2]): kernel trace + FETCH_SIZE / WRITE_SIZE in separate passes -> gpurun_out/bits_prof.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/bits_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python $REPO/tools/bench_bits.py > $OUT/bench.json 2> $OUT/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c -d $OUT/$c -o p --output-format csv -- python $REPO/tools/bench_bits.py > /dev/null 2> $OUT/$c.err
done
cd $REPO
python - <<'PY' | tee gpurun_out/bits_prof.txt
import csv, glob, collections, json
print("bitset kernels of configs[2] (tools/prof_bits.sh: tools/bench_bits.py under rocprofv3; --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in their own passes; KiB raw per dispatch: FETCH_SIZE reads half of a streaming read on gfx950, WRITE_SIZE is exact)")
print(open('gpurun_out/bits_prof/bench.json').read().strip().splitlines()[-1][:1500])
f = glob.glob('gpurun_out/bits_prof/trace/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0]))):
    if 'bits_' in r['Name']:
        print("%-64s calls=%-5s avg=%9.1f us min=%8.1f max=%8.1f" % (r['Name'].split('(')[0].replace('void ', '').replace('bxmi::', '')[:64], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('gpurun_out/bits_prof/%s/**/*counter_collection.csv' % c, recursive=True)
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if 'bits_' in r['Kernel_Name']:
                acc[r['Kernel_Name'].split('(')[0].replace('void ', '').replace('bxmi::', '')[:64]].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        print("  %-10s %-64s mean=%.6g KiB n=%d" % (c, k, sum(v) / len(v), len(v)))
PY
rm -rf $OUT
