#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/bx-python_amd:$PYTHONPATH
timeout 1500 python -m pytest tests/test_gpu_intervals.py tests/test_gpu_bitset.py -x -q 2>&1 | tail -3
MODE=random python tools/bench_find.py 2>/dev/null | tail -1 | cut -c1-140
MODE=sorted python tools/bench_find.py 2>/dev/null | tail -1 | cut -c1-140
MODE=random bash tools/prof_find.sh 2>&1 | grep "scan_" | head -4
