#!/usr/bin/env python3
"""The host-pointer count (bxmi_ivl_count on numpy arrays, configs[1]: 100 M queries x 10 M targets) under several chunk sizes,
into a fresh and into a touched output array.  NQ / NT shrink it; CHUNKS="0 4 8 16" (Mi queries, 0 = one piece)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
from bxmi import _ffi  # noqa: E402
from bxmi.intervals import IntervalIndex  # noqa: E402
from bxmi._ffi import call, ptr  # noqa: E402

NQ = int(os.environ.get("NQ", 100_000_000))
NT = int(os.environ.get("NT", 10_000_000))
G = 250_000_000
rng = np.random.default_rng(201)
ts = rng.integers(0, G, NT, dtype=np.int32)
te = ts + rng.integers(1, 1001, NT, dtype=np.int32)
rng = np.random.default_rng(202)
qs = rng.integers(0, G, NQ, dtype=np.int32)
qe = qs + rng.integers(1, 1001, NQ, dtype=np.int32)
ix = IntervalIndex()
ix.append(ts, te)
ix.seal()
ref_counts, ref_total = ix.count(qs, qe)  # also sizes the handle's device buffers


def one(counts):
    total = C.c_int64(0)
    t0 = time.perf_counter()
    call("bxmi_ivl_count", ix._h, ptr(qs), ptr(qe), NQ, ptr(counts), C.byref(total))
    dt = time.perf_counter() - t0
    return dt, total.value


out = {}
for cfg in os.environ.get("CHUNKS", "0 8:0 8:1 8:2 8:4 16:2 4:2").split():
    ch, touchers = (int(x) for x in (cfg + ":2").split(":")[:2])  # "Mi queries[:threads touching the output's pages]"
    call("bxmi_set_option", b"ivl.host_chunk", ch << 20)
    call("bxmi_set_option", b"ivl.host_touchers", touchers)
    fresh = []
    for _ in range(3):
        c = np.empty(NQ, dtype=np.int32)  # pages never touched: the download faults them in
        dt, tot = one(c)
        fresh.append(dt)
        ok = bool(tot == ref_total and np.array_equal(c, ref_counts))
        del c
    c = np.zeros(NQ, dtype=np.int32)
    c[:] = 1
    warm = [one(c)[0] for _ in range(3)]
    tonly = [one(None)[0] for _ in range(3)]
    out[cfg] = dict(fresh_out_ms=[round(x * 1e3, 2) for x in fresh], touched_out_ms=[round(x * 1e3, 2) for x in warm],
                                  total_only_ms=[round(x * 1e3, 2) for x in tonly], same_counts=ok)
    print("chunk %-5s: fresh %s  touched %s  total-only %s  ok=%s" % (cfg, out[cfg]["fresh_out_ms"], out[cfg]["touched_out_ms"],
                                                                      out[cfg]["total_only_ms"], ok), flush=True)
print(json.dumps(out))

# ---- find() on host arrays (configs[4]: 50 M x 50 M, ~250 M hits): upload, find, 0.4 GB of offsets + 1 GB of hits down
if os.environ.get("FIND", "1") == "1":
    del ix, ref_counts, qs, qe
    from bxmi import synth

    NF = int(os.environ.get("NF", 50_000_000))
    (ts, te), (qs, qe) = synth.cfg5(NF, NF)
    ix = IntervalIndex()
    ix.append(ts, te)
    ix.seal()
    off0, hits0 = ix.find(qs, qe, cap_hint=6 * NF)
    fout = {}
    for touchers in (0, 2, 4):
        call("bxmi_set_option", b"ivl.host_touchers", touchers)
        ms = []
        for _ in range(3):
            t0 = time.perf_counter()
            off, hits = ix.find(qs, qe, cap_hint=6 * NF)
            ms.append(round((time.perf_counter() - t0) * 1e3, 2))
            same = bool(np.array_equal(off, off0) and np.array_equal(hits, hits0))
            del off, hits
        fout["touchers_%d" % touchers] = dict(ms=ms, same=same)
        print("find host, %d touchers: %s ms  same=%s  (%d hits)" % (touchers, ms, same, len(hits0)), flush=True)
    print(json.dumps(dict(find_host=fout)))
