#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-find --no-bitset 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('count ms', d['ms_per_step'], 'sorted', d['sorted_queries']['ms_per_pass'], 'genome', d['genome']['ms_per_step'], d['genome']['kernel_ms_slowest_rank'], d['genome']['parity'])
"
for o in "ivl.sl_flat=0" "ivl.bd_depth=4"; do
BXMI_OPTS=$o python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-find --no-bitset --no-sorted 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$o: genome', d['genome']['ms_per_step'], d['genome']['kernel_ms_slowest_rank'])
"
done
