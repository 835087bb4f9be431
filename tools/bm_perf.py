#!/usr/bin/env python3
"""Timing matrix of the large-batch count passes on configs[1] (100M x 10M): the bucketed search pass against the
bitmap-cell pass in its tile shapes / unroll depths, shuffled and sorted queries.  One JSON line per measurement.
NQ, REPS, SORTED=0/1 from the environment."""
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
REPS = int(os.environ.get("REPS", 5))


def opt(k, v):
    _ffi.call("bxmi_set_option", k.encode(), int(v))


(ts, te), (qs_h, qe_h) = synth.cfg2(10_000_000, NQ)
ix = IntervalIndex()
ix.append(ts, te)
ix.seal()
stream = torch.cuda.current_stream().cuda_stream
counts = torch.empty(NQ, dtype=torch.int32, device="cuda")
ref = torch.empty(NQ, dtype=torch.int32, device="cuda")
total = torch.zeros(1, dtype=torch.int64, device="cuda")


def run(qs, qe, out):
    total.zero_()
    ix.count_dev(qs.data_ptr(), qe.data_ptr(), NQ, out.data_ptr(), total.data_ptr(), stream)


def timed(qs, qe, out):
    run(qs, qe, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        ix.count_dev(qs.data_ptr(), qe.data_ptr(), NQ, out.data_ptr(), total.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


ORDERS = os.environ.get("ORDERS", "generated,sorted").split(",")
# variant:u:pair:pipe
CONFIGS = [tuple(int(x) for x in c.split(":")) for c in os.environ.get("CONFIGS", "-1:2:1:1,0:2:1:1,2:2:1:1,2:4:1:1,0:4:0:0").split(",")]
EXP_PAIR = int(os.environ.get("EXP_PAIR", "1"))
EXPS = [int(x) for x in os.environ.get("EXPS", "").split(",") if x]
for order in ORDERS:
    qs, qe = torch.from_numpy(qs_h).cuda(), torch.from_numpy(qe_h).cuda()
    if order == "sorted":
        o = torch.argsort(qs, stable=True)
        qs, qe = qs[o].contiguous(), qe[o].contiguous()
        del o
    opt("ivl.bitmap", 0)
    ms = timed(qs, qe, ref)
    run(qs, qe, ref)
    torch.cuda.synchronize()
    want_total = int(total.item())
    print(json.dumps(dict(order=order, path="bucketed", ms=round(ms, 4), total=want_total)), flush=True)
    opt("ivl.bitmap", -1)
    opt("ivl.bm_pair", EXP_PAIR)
    for exp in EXPS:  # diagnostics: the search kernel without its stores (1), its record loads (2), both (3)
        opt("ivl.bm_exp", exp)
        print(json.dumps(dict(order=order, path="bitmap", exp=exp, ms=round(timed(qs, qe, counts), 4))), flush=True)
    opt("ivl.bm_exp", 0)
    for variant, u, pair, pipe in CONFIGS:
        if True:
            opt("ivl.bm_variant", variant)
            opt("ivl.bm_u", u)
            opt("ivl.bm_pair", pair)
            opt("ivl.bm_pipe", pipe & 1)
            opt("ivl.bm_nt", pipe >> 1)
            ms = timed(qs, qe, counts)
            run(qs, qe, counts)
            torch.cuda.synchronize()
            same = bool(torch.equal(counts, ref)) and int(total.item()) == want_total
            print(json.dumps(dict(order=order, path="bitmap", variant=variant, u=u, pair=pair, pipe=pipe, ms=round(ms, 4), same_as_bucketed=same,
                                  state=ix.bitmap_state())), flush=True)
            if not same:
                bad = torch.nonzero(counts != ref).flatten()
                print(json.dumps(dict(mismatches=int(bad.numel()), first=bad[:8].tolist(), got=counts[bad[:8]].tolist(),
                                      want=ref[bad[:8]].tolist(), qs=qs[bad[:8]].tolist(), qe=qe[bad[:8]].tolist())), flush=True)
    opt("ivl.bm_variant", -1)
    opt("ivl.bm_u", 2)
    opt("ivl.bm_pair", 1)
    opt("ivl.bm_pipe", 1)
    del qs, qe
