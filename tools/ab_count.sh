#!/bin/bash
# A/B of libbxmi options on the count pass: OPTS="a=1 b=0,c=1" MODE=random|sorted
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in 1 2; do
for opts in ${OPTS:-ivl.sorted_path=0 ivl.sorted_path=1}; do echo -n "$opts  "; BXMI_OPTS=$opts REPS=${REPS:-10} python tools/count_only.py 2>&1 | tail -1; done
done
