"""
GPU tests of bench.py's own launcher: `python bench.py --gpus N` (the driver's command shape, no torchrun in front)
starts N ranks itself.  On a one-GPU box the N > 1 path is rehearsed with BENCH_DRY_MULTI=1 (every rank on cuda:0,
gloo); without it, more ranks than devices is an error, never a quiet one-rank run.
Scaled shape: scripts/interval_join.py:21-28 (a dict of per-chromosome trees, dealt to the ranks).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from bxmi import synth
from conftest import ROOT

pytestmark = pytest.mark.gpu


def _env(**kw):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_DRY_MULTI", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(kw)
    return env


def _line(stdout):
    rows = [r for r in stdout.splitlines() if r.startswith("{")]
    assert len(rows) == 1, stdout[-2000:]
    return json.loads(rows[0])


def test_bench_launches_its_own_ranks_dry_two():
    nq, nt = 4_000_000, 400_000
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--queries", str(nq),
                        "--targets", str(nt), "--no-cpu-baseline"], capture_output=True, text=True, timeout=1500, env=_env(BENCH_DRY_MULTI="1"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = _line(p.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "dry_run" in line
    assert line["collective"]["world"] == 2 and line["collective"]["ranks_counted_by_all_reduce_of_ones"] == 2
    # the all-reduced total of a step = both ranks' totals, each rank's queries drawn from its own seed (bench.py: 202 + 1000 * rank)
    (ts, te), _ = synth.cfg2(nt, 1)
    s_sorted, e_sorted = np.sort(ts), np.sort(te)
    want = []
    for rank in range(2):
        qs, qe = synth.uniform_intervals(nq, 202 + 1000 * rank)
        want.append(int((np.searchsorted(s_sorted, qe, "left") - np.searchsorted(e_sorted, qs, "right")).sum(dtype=np.int64)))  # proper intervals
    assert line["overlaps_per_step_rank0"] == want[0]
    assert line["overlaps_per_step_all_ranks"] == want[0] + want[1]
    assert "all-reduced total == sum of the ranks' totals: True" in line["parity"], line["parity"]
    assert line["value"] > 0 and abs(line["value"] - 2 * nq * 2 / (line["ms_per_step"] * 2 * 1e-3) / 1e6) < 1e-2 * line["value"]
    # the strong-scaling leg rode along: the genome on two ranks against rank 0 alone, speed-up at top level
    g = line["genome"]
    assert g["n_gpus"] == 2 and g["scaling"] == "strong" and g["parity"]["reduced_totals_equal_sum_of_owner_counts_every_step"]
    assert line["speedup_vs_1gpu"] == g["speedup_vs_1gpu"] and g["speedup_vs_1gpu"] > 0


def test_bench_refuses_more_ranks_than_devices_on_the_gpu_box():
    from bxmi import _ffi  # (not torch: its bundled HIP runtime would become this process's first, and libbxmi's RCCL binding a mix of two)

    n = _ffi.device_count() + 7
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=_env())
    assert p.returncode != 0 and "refusing to run fewer ranks than asked" in p.stderr, p.stderr[-2000:]
    assert not [r for r in p.stdout.splitlines() if r.startswith("{")]
