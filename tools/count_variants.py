#!/usr/bin/env python3
"""A/B matrix of the large-batch count pass on configs[1] (100M x 10M) in ONE process: every variant is a set of
`bxmi_set_option` knobs; per variant REPS passes timed with events, the counts compared with the first variant's, and
a pause between variants so that a kernel trace of the run (rocprofv3 --kernel-trace) can be cut into segments by
tools/trace_segments.py.  VARIANTS="name:key=value+key=value,name2:..."; NQ, NT, REPS, ORDER=generated|sorted|clustered."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import numpy as np
import torch

from bxmi import _ffi, synth
from bxmi.intervals import IntervalIndex

DEFAULTS = _ffi.options()  # read before any variant turns a knob
NQ = int(os.environ.get("NQ", 100_000_000))
NT = int(os.environ.get("NT", 10_000_000))
REPS = int(os.environ.get("REPS", 5))
ORDER = os.environ.get("ORDER", "generated")
DEFAULT = "dense:,dense128k:ivl.bd_chunk=131072,dense64k:ivl.bd_chunk=65536,dense_t16k:ivl.bm_variant=0,pair:ivl.dense=0"
VARIANTS = []
for item in os.environ.get("VARIANTS", DEFAULT).split(","):
    name, _, kv = item.partition(":")
    VARIANTS.append((name, [(k, int(v)) for k, _, v in (x.partition("=") for x in kv.split("+") if x)]))

if ORDER == "clustered":
    (ts, te), (qs_h, qe_h) = synth.clustered(NT, NQ)
else:
    (ts, te), (qs_h, qe_h) = synth.cfg2(NT, NQ)
    if ORDER == "sorted":
        o = np.argsort(qs_h, kind="stable")
        qs_h, qe_h = qs_h[o], qe_h[o]
ix = IntervalIndex()
ix.append(ts, te)
t0 = time.perf_counter()
ix.seal()
seal_s = time.perf_counter() - t0
stream = torch.cuda.current_stream().cuda_stream
qs, qe = torch.from_numpy(qs_h).cuda(), torch.from_numpy(qe_h).cuda()
counts = torch.empty(NQ, dtype=torch.int32, device="cuda")
ref = None
total = torch.zeros(1, dtype=torch.int64, device="cuda")
print(json.dumps(dict(order=ORDER, nq=NQ, nt=NT, seal_s=round(seal_s, 4))), flush=True)
touched = set()
for name, opts in VARIANTS:
    for k, v in opts:
        _ffi.call("bxmi_set_option", k.encode(), v)
        touched.add(k)
    if any(k == "ivl.bd_unit_log2" for k, _ in opts) or os.environ.get("RESEAL"):
        ix.seal()  # the geometry of the images is decided when an index is prepared
    total.zero_()
    ix.count_dev(qs.data_ptr(), qe.data_ptr(), NQ, counts.data_ptr(), total.data_ptr(), stream)  # warm-up (builds the images)
    torch.cuda.synchronize()
    tot = int(total.item())
    if ref is None:
        ref = counts.clone()
        same = True
    else:
        same = bool(torch.equal(counts, ref))
    time.sleep(0.05)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        ix.count_dev(qs.data_ptr(), qe.data_ptr(), NQ, counts.data_ptr(), total.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / REPS
    print(json.dumps(dict(variant=name, opts=dict(opts), ms=round(ms, 4), gqps=round(NQ / ms / 1e6, 2), total=tot, same_as_first=same,
                          dense=ix.dense_state(), flat=ix.flat_state()[0], slices=ix.slice_state()[0])), flush=True)
    if not same:
        bad = torch.nonzero(counts != ref).flatten()
        print(json.dumps(dict(mismatches=int(bad.numel()), first=bad[:8].tolist(), got=counts[bad[:8]].tolist(), want=ref[bad[:8]].tolist(),
                              qs=qs[bad[:8]].tolist(), qe=qe[bad[:8]].tolist())), flush=True)
    time.sleep(0.05)
    for k, _ in opts:  # back to the library's defaults
        _ffi.call("bxmi_set_option", k.encode(), DEFAULTS[k])
    if any(k == "ivl.bd_unit_log2" for k, _ in opts):
        ix.seal()
