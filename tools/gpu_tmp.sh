#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/bx-python_amd:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_intervals.py -x -q -k "find_through" 2>&1 | tail -3
for o in "ivl.sl_hu_parts=1" "ivl.sl_hu_parts=4" "ivl.sl_hu_parts=8" "ivl.sl_hu_parts=16"; do
echo "== $o"
BXMI_OPTS=$o MODE=random python tools/bench_find.py 2>/dev/null | tail -1 | cut -c1-140
BXMI_OPTS=$o MODE=random bash tools/prof_find.sh 2>&1 | grep "sl_hits" | head -3
done
