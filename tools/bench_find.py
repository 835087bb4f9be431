#!/usr/bin/env python3
"""configs[4]: overlap join with hit materialisation (interval_join.py's path), 50M x 50M, CSR hit list in HBM.
Secondary measurement (not the bench line): prints one JSON object.  NT/NQ env vars scale it down."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import numpy as np
import torch

from bxmi import synth
from bxmi.intervals import IntervalIndex

NT = int(os.environ.get("NT", 50_000_000))
NQ = int(os.environ.get("NQ", 50_000_000))
(ts, te), (qs_h, qe_h) = synth.cfg5(NT, NQ)
MODE = os.environ.get("MODE", "random")  # sorted: the queries ordered by start, as a sorted BED file would give them
if MODE == "sorted":
    o = np.argsort(qs_h, kind="stable")
    qs_h, qe_h = qs_h[o], qe_h[o]
t0 = time.perf_counter()
ix = IntervalIndex()
ix.append(ts, te)
ix.seal()
build_s = time.perf_counter() - t0
qs, qe = torch.from_numpy(qs_h).cuda(), torch.from_numpy(qe_h).cuda()
offs = torch.empty(NQ + 1, dtype=torch.int64, device="cuda")
cap = int(NQ * 8)
hits = torch.empty(cap, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
rc, total = ix.find_dev(qs.data_ptr(), qe.data_ptr(), NQ, offs.data_ptr(), hits.data_ptr(), cap, stream)
assert rc == 0, rc
torch.cuda.synchronize()
reps = 3
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ix.find_dev(qs.data_ptr(), qe.data_ptr(), NQ, offs.data_ptr(), hits.data_ptr(), cap, stream)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
# size-independent checks on the full result
counts = torch.empty(NQ, dtype=torch.int32, device="cuda")
tot = torch.zeros(1, dtype=torch.int64, device="cuda")
ix.count_dev(qs.data_ptr(), qe.data_ptr(), NQ, counts.data_ptr(), tot.data_ptr(), stream)
torch.cuda.synchronize()
d_ts, d_te = torch.from_numpy(ts).cuda(), torch.from_numpy(te).cuda()
h = hits[:total].long()
rep = torch.repeat_interleave(torch.arange(NQ, device="cuda"), (offs[1:] - offs[:-1]))
ok_overlap = bool(((d_te[h] > qs[rep]) & (d_ts[h] < qe[rep])).all().item())
ok_counts = bool(torch.equal((offs[1:] - offs[:-1]).int(), counts)) and int(tot.item()) == total == int(offs[-1].item())
# hits of one query come in the tree's order: (start, insertion index) non-decreasing for proper intervals
key = d_ts[h].long() * (1 << 31) + h
same = rep[1:] == rep[:-1]
ok_order = bool((key[1:][same] >= key[:-1][same]).all().item())
alg = NQ * 16 + total * 4 + NT * 8
print(json.dumps(dict(workload="configs[4]: %d x %d join, G=2e9, len U[1,200], CSR in HBM, query order: %s" % (NQ, NT, MODE), ms=round(ms, 3),
                      mqueries_per_s=round(NQ / ms / 1e3, 1), mhits_per_s=round(total / ms / 1e3, 1), hits=total,
                      algorithmic_bytes=alg, achieved_gbs=round(alg / ms / 1e6, 1), frac_of_8tbs=round(alg / ms / 1e6 / 8000, 4),
                      index_build_s=round(build_s, 2), every_hit_overlaps=ok_overlap, counts_match_count_path=ok_counts,
                      hits_in_tree_order=ok_order)))
