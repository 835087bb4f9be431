#!/bin/bash
# One GPU iteration on find() (via gpurun): [TESTS=1] the find parity tests, tools/bench_find.py (configs[4]) in generated and sorted
# order under each BXMI_OPTS setting of OPTSETS (";"-separated, "-" = defaults), [STATS=1] the kernel list of both.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${TESTS:-1}" = "1" ]; then
  timeout 1500 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "find or join or cfg5 or csr" > gpurun_out/test_find.log 2>&1
  echo "find tests rc=$?"; tail -3 gpurun_out/test_find.log
fi
IFS=';' read -ra SETS <<< "${OPTSETS:--}"
for o in "${SETS[@]}"; do
  for mode in ${MODES:-random sorted}; do
    [ "$o" = "-" ] && oo="" || oo="$o"
    echo -n "[$o] $mode: "; MODE=$mode BXMI_OPTS="$oo" timeout 600 python tools/bench_find.py 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms'], 'ms', d['frac_of_8tbs'], 'ok' if d['every_hit_overlaps'] and d['counts_match_count_path'] and d['hits_in_tree_order'] else 'WRONG')"
  done
done
if [ "${STATS:-1}" = "1" ]; then
  for mode in ${MODES:-random sorted}; do
    echo "--- kernels, $mode order"
    (cd /tmp && MODE=$mode rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_find -o f --output-format csv -- python $REPO/tools/bench_find.py > /dev/null 2>&1)
    python - "$mode" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/prof_find/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if any(k in r["Name"] for k in ("bm_", "bd_", "fx_", "sl_", "ivl_local", "part_fill", "lf_", "ivl_sorted", "scan_"))]
with open("gpurun_out/find_kernels_%s.txt" % sys.argv[1], "w") as o:
    for r in rows[:20]:
        line = "%-64s calls=%-4s avg=%9.1f us" % (r["Name"].split("(")[0][-64:], r["Calls"], float(r["AverageNs"]) / 1e3)
        print(line); o.write(line + "\n")
PY
    rm -rf gpurun_out/prof_find
  done
fi
