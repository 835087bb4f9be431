#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD/bx-python_amd:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_intervals.py -x -q -k "find_through or bitmap_pass_diff or count_multi" 2>&1 | tail -3
MODE=random python tools/bench_find.py 2>/dev/null | tail -1 | cut -c1-140
MODE=random bash tools/prof_find.sh 2>&1 | grep "sl_" | head -4
timeout 600 python bench.py --workload genome --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('genome', d['ms_per_step'])"
