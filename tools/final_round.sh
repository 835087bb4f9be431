#!/bin/bash
# One GPU call for the end of a round: smoke + every GPU test + the bench line (tools/gpu_round.sh), the rocprofv3 summaries of the
# count pass (tools/profile.sh), the counters of the find pipeline and the 8-GPU rank share.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
BENCH_STEPS=20 timeout 1500 bash tools/gpu_round.sh > gpurun_out/round.log 2>&1
tail -12 gpurun_out/round.log | cut -c1-300
timeout 400 bash tools/profile.sh > gpurun_out/profile.log 2>&1; tail -3 gpurun_out/profile.log
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" timeout 300 bash tools/pmc_find.sh > gpurun_out/pmc_find.log 2>&1; grep -c "" gpurun_out/pmc_find/summary.txt
WORLDS=1,2,4,8 timeout 300 python tools/rank_share.py > gpurun_out/share_all.json 2> gpurun_out/share_all.err
timeout 300 bash tools/pmc_flat.sh search_sq > gpurun_out/search_sq.log 2>&1
ROUNDS=40 SEED=2024 timeout 400 python tools/fuzz_intervals.py > gpurun_out/fuzz.log 2>&1; tail -1 gpurun_out/fuzz.log
cd /tmp
WORLDS=8 timeout 200 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/share -o s --output-format csv -- python $REPO/tools/rank_share.py > $REPO/gpurun_out/share.json 2> $REPO/gpurun_out/share.err
cd $REPO
python - <<'PY' > gpurun_out/rank_share_summary.json
import csv, glob, json
out = json.loads(open('gpurun_out/share_all.json').read().strip().splitlines()[-1])
traced = json.loads(open('gpurun_out/share.json').read().strip().splitlines()[-1])
f = glob.glob('gpurun_out/share/**/*kernel_stats.csv', recursive=True)
ks = []
for r in list(csv.DictReader(open(f[0])))[:14]:
    ks.append(dict(kernel=r['Name'].split('(')[0].replace('void ', '').replace('bxmi::', '')[:70], calls=int(r['Calls']), avg_us=round(float(r['AverageNs']) / 1e3, 1)))
print(json.dumps(dict(rank_share=out, rank_share_8_under_rocprof=traced, kernels_of_the_run=ks, note="rank_share: tools/rank_share.py WORLDS=1,2,4,8 untraced (speed-ups against the one-GPU pass of the same run); the kernel list: WORLDS=8 under rocprofv3 --kernel-trace --stats; the kernel list covers index builds and warm-up too, the count pass is bm_params .. bm_fold_totals"), indent=1))
PY
rm -rf gpurun_out/share
cat gpurun_out/rank_share_summary.json | head -12
