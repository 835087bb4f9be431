#!/usr/bin/env python3
"""configs[3] as ONE rank of WORLD sees it (the heaviest rank of the LPT deal), on this GPU: its fused count pass alone, the same
followed by a bxmi_allreduce_i64 of the 24 per-chromosome totals (a communicator of ONE rank: it prices the launch and the RCCL
entry, not the wire -- RCCL refuses two ranks on one device), and the share with every chromosome's queries sorted by start.
Predicts the strong-scaling curve without the other GPUs; NO curve has been measured on hardware.  WORLDS env (default 1,2,4,8)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import torch

from bxmi import shard, synth
from bxmi.intervals import IntervalIndex

comm = None
try:
    comm = shard.Comm(0, 1, lambda raw: raw)
except Exception as ex:  # (no RCCL in reach: the share is reported without the collective)
    print("no communicator: %r" % (ex,), file=sys.stderr)
red = torch.zeros(24, dtype=torch.int64, device="cuda")


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {}
for world in [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
    tsz, qsz = synth.cfg4_sizes(10_000_000), synth.cfg4_sizes(100_000_000)
    weights = {c: tsz[c] + qsz[c] for c in synth.HG19_SIZES}
    assign = shard.lpt_assign(weights, world)
    mine = max(assign, key=lambda part: sum(weights[c] for c in part))
    ixs, qs, qe, cnt = [], [], [], []
    for c in mine:
        (ts, te), (a, b) = synth.cfg4_chrom(c)
        ix = IntervalIndex()
        ix.append(ts, te)
        ix.seal()
        ixs.append(ix)
        qs.append(torch.from_numpy(a).cuda()), qe.append(torch.from_numpy(b).cuda())
        cnt.append(torch.empty(len(a), dtype=torch.int32, device="cuda"))
    tot = torch.zeros(len(mine), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        IntervalIndex.count_multi_dev(ixs, [x.data_ptr() for x in qs], [x.data_ptr() for x in qe], [x.numel() for x in qs], [x.data_ptr() for x in cnt],
                                      [tot[i:].data_ptr() for i in range(len(mine))], stream)

    ms = timed(step)
    out[world] = dict(chromosomes=len(mine), queries=int(sum(x.numel() for x in qs)), ms=round(ms, 4))

    def step_total():  # the per-chromosome TOTALS only (configs[3]: "RCCL all-reduce on counts"): nothing stored per query
        IntervalIndex.count_multi_dev(ixs, [x.data_ptr() for x in qs], [x.data_ptr() for x in qe], [x.numel() for x in qs], [None] * len(mine),
                                      [tot[i:].data_ptr() for i in range(len(mine))], stream)

    tot.zero_()
    step()
    want = tot.clone()
    tot.zero_()
    step_total()
    torch.cuda.synchronize()
    assert torch.equal(tot, want), (tot, want)
    out[world]["ms_total_only"] = round(timed(step_total), 4)
    if os.environ.get("PLAIN_ONLY"):  # (profiling runs: only the shuffled share's steady-state passes in the kernel list)
        for ix in ixs:
            ix.close()
        del qs, qe, cnt
        torch.cuda.empty_cache()
        continue
    if comm is not None:
        def step_reduce():
            step()
            red[: len(mine)].copy_(tot, non_blocking=True)
            comm.allreduce_i64(red.data_ptr(), 24, stream)

        out[world]["ms_with_allreduce_world_of_one"] = round(timed(step_reduce), 4)
    # the same share with every chromosome's queries sorted by start (the sorted walk over segments, no exchange)
    for i in range(len(qs)):
        o = torch.argsort(qs[i], stable=True)
        qs[i], qe[i] = qs[i][o].contiguous(), qe[i][o].contiguous()
        del o
    for _ in range(4):  # (the exact order check comes back one call after the probe saw no descent)
        step()
        torch.cuda.synchronize()
    out[world]["ms_sorted_queries"] = round(timed(step, warm=1), 4)
    for ix in ixs:
        ix.close()
    del qs, qe, cnt
    torch.cuda.empty_cache()
base = out[min(out)]["ms"]
for w in out:
    if "ms_sorted_queries" not in out[w]:
        continue
    out[w]["speedup_without_collective"] = round(base / out[w]["ms"], 2)
    if "ms_with_allreduce_world_of_one" in out[w]:
        out[w]["speedup_with_allreduce_world_of_one"] = round(base / out[w]["ms_with_allreduce_world_of_one"], 2)
    out[w]["speedup_sorted_vs_sorted"] = round(out[min(out)]["ms_sorted_queries"] / out[w]["ms_sorted_queries"], 2)
    out[w]["speedup_total_only"] = round(out[min(out)]["ms_total_only"] / out[w]["ms_total_only"], 2)
print(json.dumps(out))
