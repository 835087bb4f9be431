#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for o in "ivl.bm_chunk=0" "ivl.bm_chunk=40000" "ivl.bm_chunk=65536" "ivl.bm_chunk=131072" "ivl.sl_flat=0"; do
echo "== $o: $(BXMI_OPTS=$o WORLDS=8 timeout 100 python tools/rank_share.py 2>&1 | tail -1 | cut -c1-120)"
done
