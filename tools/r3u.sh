#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_intervals.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "count_width or padded_runs or dense_pass or bitmap_pass_differential or scale_ or random_differential or beyond_16" 2>&1 | tail -8
REPS=10 VARIANTS="auto:,w16:ivl.bd_w8=0,auto2:" timeout 200 python tools/count_variants.py 2>&1 | grep variant | cut -c1-100
