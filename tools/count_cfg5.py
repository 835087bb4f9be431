#!/usr/bin/env python3
"""Count pass on the configs[4] shapes (50M x 50M, G=2e9): which large-batch stage serves it and how long it takes.
Secondary measurement: prints one JSON object.  NT/NQ env vars scale it down."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import torch

from bxmi import synth
from bxmi.intervals import IntervalIndex

NT = int(os.environ.get("NT", 50_000_000))
NQ = int(os.environ.get("NQ", 50_000_000))
(ts, te), (qs_h, qe_h) = synth.cfg5(NT, NQ)
ix = IntervalIndex()
ix.append(ts, te)
ix.seal()
qs, qe = torch.from_numpy(qs_h).cuda(), torch.from_numpy(qe_h).cuda()
counts = torch.empty(NQ, dtype=torch.int32, device="cuda")
total = torch.zeros(1, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
ix.count_dev(qs.data_ptr(), qe.data_ptr(), NQ, counts.data_ptr(), total.data_ptr(), stream)
torch.cuda.synchronize()
first_total = int(total.item())
reps = int(os.environ.get("REPS", 5))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ix.count_dev(qs.data_ptr(), qe.data_ptr(), NQ, counts.data_ptr(), total.data_ptr(), stream)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(json.dumps(dict(workload="configs[4] shapes, count only: %d x %d" % (NQ, NT), ms=round(ms, 3), total=first_total,
                      counts_sum=int(counts.sum(dtype=torch.int64).item()), flat_state=ix.flat_state(), slice_state=ix.slice_state())))
