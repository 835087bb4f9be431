#!/bin/bash
# kernel list of tools/bench_clustered.py under the caller's environment (BXMI_LIB, BXMI_OPTS)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cl -o f --output-format csv -- python $R/tools/bench_clustered.py > $R/gpurun_out/clustered.json 2>/dev/null
python - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_cl/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    if 'bxmi' in r['Name']: print("%-72s calls=%-4s avg=%9.1f us" % (r['Name'].split('(')[0][-72:], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf $R/gpurun_out/prof_cl
