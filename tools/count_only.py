#!/usr/bin/env python3
"""Runs only the count kernel a few times (profiling target).  MODE=random|sorted|bucket8, NQ, REPS; NOTOTAL=1: no overlap total asked for; NOCOUNTS=1: the total only."""
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
MODE = os.environ.get("MODE", "random")
REPS = int(os.environ.get("REPS", 3))
for kv in os.environ.get("BXMI_OPTS", "").split(","):
    if "=" in kv:
        k, v = kv.split("=")
        _ffi.call("bxmi_set_option", k.encode(), int(v))
(ts, te), (qs_h, qe_h) = synth.cfg2(10_000_000, NQ)
if MODE == "sorted":
    o = np.argsort(qs_h, kind="stable")
    qs_h, qe_h = qs_h[o], qe_h[o]
elif MODE.startswith("bucket"):
    o = np.argsort(qs_h >> (28 - int(MODE[6:])), kind="stable")
    qs_h, qe_h = qs_h[o], qe_h[o]
ix = IntervalIndex()
ix.append(ts, te)
ix.seal()
qs, qe = torch.from_numpy(qs_h).cuda(), torch.from_numpy(qe_h).cuda()
counts = torch.empty(NQ, dtype=torch.int32, device="cuda")
total = torch.zeros(1, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
tptr = None if os.environ.get("NOTOTAL") else total.data_ptr()
cptr = None if os.environ.get("NOCOUNTS") else counts.data_ptr()  # NOCOUNTS=1: a total-only batch
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ix.count_dev(qs.data_ptr(), qe.data_ptr(), NQ, cptr, tptr, stream)
torch.cuda.synchronize()
e0.record()
for _ in range(REPS):
    ix.count_dev(qs.data_ptr(), qe.data_ptr(), NQ, cptr, tptr, stream)
e1.record()
torch.cuda.synchronize()
print("MODE=%s NQ=%d  %.3f ms/launch  total=%d" % (MODE, NQ, e0.elapsed_time(e1) / REPS, int(total.item()) // (REPS + 1)))
