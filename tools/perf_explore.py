#!/usr/bin/env python3
"""Kernel-time exploration of the count path on the GPU box (not a test, not the bench)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import numpy as np
import torch

from bxmi import _ffi, synth
from bxmi.intervals import IntervalIndex

NQ = int(os.environ.get("NQ", 100_000_000))
(ts, te), (qs_h, qe_h) = synth.cfg2(10_000_000, NQ)
ix = IntervalIndex()
ix.append(ts, te)
ix.seal()
stream = torch.cuda.current_stream().cuda_stream
counts = torch.empty(NQ, dtype=torch.int32, device="cuda")
total = torch.zeros(1, dtype=torch.int64, device="cuda")


def opt(k, v):
    _ffi.call("bxmi_set_option", k.encode(), int(v))


def run(qs, qe, reps=5):
    for _ in range(2):
        ix.count_dev(qs.data_ptr(), qe.data_ptr(), NQ, counts.data_ptr(), total.data_ptr(), stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ix.count_dev(qs.data_ptr(), qe.data_ptr(), NQ, counts.data_ptr(), total.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {}
qs, qe = torch.from_numpy(qs_h).cuda(), torch.from_numpy(qe_h).cuda()
for lds in (18688, 1024, 0):
    for grid in (0, 512, 1024, 2048):
        for gs in (0, 1):
            opt("ivl.lds_ints", lds), opt("ivl.count_grid", grid), opt("ivl.group_sum", gs)
            res["random lds=%d grid=%d sum=%s" % (lds, grid, "dpp" if gs == 0 else "shfl")] = round(run(qs, qe), 3)
opt("ivl.lds_ints", 18688), opt("ivl.count_grid", 0), opt("ivl.group_sum", 0)
# locality experiments: what a query partition would buy
order = np.argsort(qs_h, kind="stable")
qs_s, qe_s = torch.from_numpy(qs_h[order]).cuda(), torch.from_numpy(qe_h[order]).cuda()
res["fully sorted by qs"] = round(run(qs_s, qe_s), 3)
for bits in (4, 8, 12):
    shift = 28 - bits
    order = np.argsort(qs_h >> shift, kind="stable")
    a, b = torch.from_numpy(qs_h[order]).cuda(), torch.from_numpy(qe_h[order]).cuda()
    for lds in (18688, 1024):
        opt("ivl.lds_ints", lds)
        res["bucketed %d bits lds=%d" % (bits, lds)] = round(run(a, b), 3)
    opt("ivl.lds_ints", 18688)
for k, v in res.items():
    print("%-45s %8.3f ms  %8.1f Mq/s" % (k, v, NQ / v / 1e3))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "perf_explore.json"), "w"), indent=1)
