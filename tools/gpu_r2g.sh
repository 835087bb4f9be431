#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/r2g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_intervals.py -m gpu -x -q --timeout 900 -p no:cacheprovider -k "genome" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" > $OUT/info.log
timeout 600 python bench.py --workload genome --steps 10 --warmup 3 > $OUT/bench_genome.json 2> $OUT/bench_genome.err
echo "genome rc=$?" >> $OUT/info.log
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/info.log
cat $OUT/info.log; tail -8 $OUT/pytest.log; cat $OUT/bench_genome.json; tail -3 $OUT/bench_genome.err; cat $OUT/bench.json; tail -5 $OUT/bench.err
