#!/bin/bash
# A/B of builds on the bench line's legs (via gpurun): LIBS = library names under bx-python_amd/bxmi/, alternated ROUNDS times on one box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for r in $(seq 1 ${ROUNDS:-2}); do
  for lib in ${LIBS:-libbxmi_base.so libbxmi.so}; do
    BXMI_LIB=$PWD/bx-python_amd/bxmi/$lib timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pcie ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
g=lambda *k: (lambda x: x)(__import__('functools').reduce(lambda a,b: (a or {}).get(b) if isinstance(a,dict) else None, k, d))
print('[$lib] pass', d['ms_per_step'], 'kernel', d['roofline'].get('kernel_ms'), '| sorted', g('sorted_queries','ms_per_pass'), '| total-only', g('total_only','ms_per_pass'),
      '| genome', g('genome','ms_per_step'), '| clustered', g('clustered','generated_order','ms'), g('clustered','sorted_by_start','ms'),
      '| find', g('find_csr','generated_order','ms'), g('find_csr','sorted_by_start','ms'),
      '| bits group pop/iand', g('bitset','one_launch_per_genome','ms','popcount'), g('bitset','one_launch_per_genome','ms','iand'), 'per-chrom', g('bitset','ms','popcount'), g('bitset','ms','iand'),
      '|', d['parity'][:60])"
  done
done
