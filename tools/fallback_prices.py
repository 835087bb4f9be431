#!/usr/bin/env python3
"""What the fallback stages of the count path cost on the shapes that still select them BY DEFAULT (end):
  1. round 1's bucketed pass: a large batch that asks for the TOTAL only (no per-query counts) -- configs[1] without counts;
  2. the same pass forced under per-query counts (ivl.bitmap=0), for comparison with the default pass of the same call;
  3. key slices with 16 / 64 lanes per run: the count of configs[4] (50 M targets on 2 x 10^9 coordinates: buckets of 2^20
     coordinates, no image fits) -- the shape find()'s count half runs on;
  4. the direct kernel: an index with reversed targets, 20 M queries."""
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

stream = torch.cuda.current_stream().cuda_stream


def timed(ix, qs, qe, nq, counts, total, reps=5):
    for _ in range(2):
        ix.count_dev(qs.data_ptr(), qe.data_ptr(), nq, counts.data_ptr() if counts is not None else None, total.data_ptr(), stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    total.zero_()
    e0.record()
    for _ in range(reps):
        ix.count_dev(qs.data_ptr(), qe.data_ptr(), nq, counts.data_ptr() if counts is not None else None, total.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4), int(total.item()) // reps


out = {}
total = torch.zeros(1, dtype=torch.int64, device="cuda")
(ts, te), (qs_h, qe_h) = synth.cfg2(10_000_000, 100_000_000)
ix = IntervalIndex()
ix.append(ts, te)
ix.seal()
qs, qe = torch.from_numpy(qs_h).cuda(), torch.from_numpy(qe_h).cuda()
counts = torch.empty(len(qs_h), dtype=torch.int32, device="cuda")
ms, tot = timed(ix, qs, qe, len(qs_h), counts, total)
out["configs1_default_pass_with_counts"] = dict(ms=ms, total=tot)
ms, tot2 = timed(ix, qs, qe, len(qs_h), None, total)
out["configs1_total_only_round1_bucketed_pass"] = dict(ms=ms, total=tot2, same_total=tot2 == tot)
_ffi.call("bxmi_set_option", b"ivl.bitmap", 0)
ms, tot3 = timed(ix, qs, qe, len(qs_h), counts, total)
_ffi.call("bxmi_set_option", b"ivl.bitmap", -1)
out["configs1_round1_bucketed_pass_forced_with_counts"] = dict(ms=ms, same_total=tot3 == tot)
ix.close()
del qs, qe, counts
torch.cuda.empty_cache()

rng = np.random.default_rng(5)
n = 50_000_000
s = rng.integers(0, 2_000_000_000, size=n, dtype=np.int64)
e = s + rng.integers(1, 201, size=n)
q = rng.integers(0, 2_000_000_000, size=n, dtype=np.int64)
r = q + rng.integers(1, 201, size=n)
ix = IntervalIndex()
ix.append(s.astype(np.int32), np.minimum(e, 2**31 - 1).astype(np.int32))
ix.seal()
qs, qe = torch.from_numpy(q.astype(np.int32)).cuda(), torch.from_numpy(np.minimum(r, 2**31 - 1).astype(np.int32)).cuda()
counts = torch.empty(n, dtype=torch.int32, device="cuda")
ms, tot = timed(ix, qs, qe, n, counts, total)
out["configs4_count_default"] = dict(ms=ms, total=tot, slices=ix.slice_state()[0], sparse=ix.sparse_state()[0], flat=ix.flat_state()[0])
for lanes in (16, 64, 1):
    _ffi.call("bxmi_set_option", b"ivl.sl_lanes", lanes)
    ms, t2 = timed(ix, qs, qe, n, counts, total)
    out["configs4_count_slices_lanes_%d" % lanes] = dict(ms=ms, same_total=t2 == tot)
_ffi.call("bxmi_set_option", b"ivl.sl_lanes", 0)
ix.close()

m = 20_000_000
s2 = s[:5_000_000].astype(np.int32) // 8
e2 = (s2 + rng.integers(1, 201, size=len(s2))).astype(np.int32)
e2[7] = s2[7] - 5  # one reversed target: the index stays on the direct kernel
ix = IntervalIndex()
ix.append(s2, e2)
ix.seal()
qs2 = torch.from_numpy((q[:m] // 8).astype(np.int32)).cuda()
qe2 = qs2 + 100
counts = torch.empty(m, dtype=torch.int32, device="cuda")
ms, tot = timed(ix, qs2, qe2, m, counts, total, reps=3)
out["reversed_target_index_direct_kernel_20M_queries"] = dict(ms=ms, total=tot, has_reversed=ix.has_reversed)
print(json.dumps(out, indent=1))
