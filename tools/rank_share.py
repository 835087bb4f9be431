#!/usr/bin/env python3
"""configs[3] as ONE rank of WORLD sees it (the heaviest rank of the LPT deal), on this GPU: its fused count pass alone.
Predicts the strong-scaling curve without the other GPUs (no collective here).  WORLD env (default 8)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import torch

from bxmi import shard, synth
from bxmi.intervals import IntervalIndex

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

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    out[world] = dict(chromosomes=len(mine), queries=int(sum(x.numel() for x in qs)), ms=round(e0.elapsed_time(e1) / reps, 4))
    for ix in ixs:
        ix.close()
    del qs, qe, cnt
    torch.cuda.empty_cache()
base = out[min(out)]["ms"]
for w in out:
    out[w]["speedup_without_collective"] = round(base / out[w]["ms"], 2)
print(json.dumps(out))
