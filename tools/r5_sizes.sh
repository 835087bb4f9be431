#!/bin/bash
# tools/bench_find.py at several batch sizes under each BXMI_OPTS setting of OPTSETS
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
IFS=';' read -ra SETS <<< "${OPTSETS:--}"
for nq in ${SIZES:-4000000 8000000 16000000}; do
for o in "${SETS[@]}"; do
  echo -n "NQ=$nq [$o] "; NQ=$nq BXMI_OPTS=$([ "$o" = "-" ] && echo "" || echo "$o") MODE=${MODE:-random} timeout 200 python tools/bench_find.py 2>&1 | tail -1 | cut -c95-135
done; done
