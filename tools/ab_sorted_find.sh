#!/bin/bash
# A/B of builds on the find of configs[4] (via gpurun): LIBS = space-separated library names under bx-python_amd/bxmi/,
# alternated ROUNDS times on one box; [TESTS=1] the find parity tests on the LAST library; the kernel list of each.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
LIBS=${LIBS:-libbxmi.so libbxmi_exp.so}
if [ "${TESTS:-0}" = "1" ]; then
  last=${LIBS##* }
  BXMI_LIB=$PWD/bx-python_amd/bxmi/$last timeout 1500 python -m pytest tests/test_gpu_intervals.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "find or join or cfg5 or csr" > gpurun_out/test_find.log 2>&1
  echo "find tests [$last] rc=$?"; tail -3 gpurun_out/test_find.log
fi
for r in $(seq 1 ${ROUNDS:-3}); do
  for lib in $LIBS; do
    for mode in ${MODES:-sorted}; do
    echo -n "[$lib] $mode: "; MODE=$mode BXMI_LIB=$PWD/bx-python_amd/bxmi/$lib timeout 600 python tools/bench_find.py 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms'], 'ms', d['frac_of_8tbs'], 'ok' if d['every_hit_overlaps'] and d['counts_match_count_path'] and d['hits_in_tree_order'] else 'WRONG')"
    done
  done
done
if [ "${STATS:-1}" = "1" ]; then
  for lib in $LIBS; do
  echo "--- kernels [$lib]"
  BXMI_LIB=$PWD/bx-python_amd/bxmi/$lib MODE=${STAT_MODE:-sorted} bash tools/find_kernels.sh | head -${STAT_LINES:-8}
  done
fi
