#!/bin/bash
# One GPU iteration (via gpurun): [TESTS=1] the interval parity tests, a short bench line, the count pass's kernel list
# (rocprofv3 --kernel-trace --stats), [SHARE=1] the projected rank shares and the kernel list of the 8-GPU share.
# Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${TESTS:-1}" = "1" ]; then
  timeout 1500 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 900 -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/test_intervals.log 2>&1
  echo "intervals rc=$?"; tail -4 gpurun_out/test_intervals.log
fi
if [ "${COUNT:-1}" = "1" ]; then
timeout 600 python bench.py --steps 20 --warmup 3 ${BENCH_ARGS:---no-cpu-baseline --no-bitset --no-find --no-genome} > gpurun_out/bench_short.json 2> gpurun_out/bench_short.err
echo "bench rc=$?"; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_short.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "| sorted", (d.get("sorted_queries") or {}).get("ms_per_pass"),
          "| total_only", (d.get("total_only") or {}).get("ms_per_pass"), "|", d["parity"][:160])
except Exception as ex:
    print("no bench line:", ex)
PY
fi
stats() {  # $1 = output dir, rest = command
  local out=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/$out -o s --output-format csv -- "$@" > /dev/null 2> $REPO/gpurun_out/$out.err)
  python - "$REPO/gpurun_out/$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
keep = [r for r in rows if any(k in r["Name"] for k in ("bm_", "bd_", "bw_", "bs_", "fx_", "sl_", "ivl_local", "part_fill"))]
with open(sys.argv[1] + "_kernels.txt", "w") as o:
    for r in keep[:24]:
        line = "%-72s calls=%-5s avg=%9.1f us" % (r["Name"].split("(")[0][-72:], r["Calls"], float(r["AverageNs"]) / 1e3)
        print(line); o.write(line + "\n")
PY
  rm -rf $REPO/gpurun_out/$out
}
if [ "${COUNT:-1}" = "1" ]; then
echo "--- count pass kernels (configs[1])"
stats prof_count python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sorted --no-find --no-bitset --no-genome
fi
if [ "${SHARE:-1}" = "1" ]; then
  echo "--- rank shares"
  WORLDS=1,2,4,8 timeout 900 python tools/rank_share.py 2> gpurun_out/rank_share.err | grep '^{' > gpurun_out/rank_share.json; cat gpurun_out/rank_share.json
  echo "--- kernels of the 8-GPU share"
  PLAIN_ONLY=1 WORLDS=8 stats prof_share8 python $REPO/tools/rank_share.py
fi
if [ -n "${SWEEP:-}" ]; then
  echo "--- 8-GPU share under options"
  for o in $SWEEP; do echo -n "$o  "; PLAIN_ONLY=1 WORLDS=${SWEEP_WORLDS:-8} BXMI_OPTS=$o timeout 600 python tools/rank_share.py 2>/dev/null | grep '^{' | cut -c1-300; done
fi
