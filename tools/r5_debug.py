#!/usr/bin/env python3
"""Round 5 debugging aid: find() through the exchange (ivl.fx_fill) against the oracle on the smallest failing shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import numpy as np
from bxmi import _ffi
from bxmi.intervals import IntervalIndex
from oracle import oracle as O

def set_opt(k, v): _ffi.call("bxmi_set_option", k.encode(), int(v))

rng = np.random.default_rng(70)
n, span = 100_000, 30_000_000
s = rng.integers(1000, span, size=n); e = s + rng.integers(0, 1200, size=n)
NQ = 50_000
qs_all = rng.integers(0, span + 2000, size=NQ); qe_all = qs_all + rng.integers(1, 2500, size=NQ)
s, e, qs_all, qe_all = (a.astype(np.int32) for a in (s, e, qs_all, qe_all))
t = O.OracleIntervalTree(); t.insert_many_arrays(s, e)
ix = IntervalIndex(); ix.append(s, e); ix.seal()
set_opt("ivl.partition", 1); set_opt("ivl.bitmap_min", 1); set_opt("ivl.sorted_path", 0)
def run(nq, fx=1, variant=-1, f=-1, rep=0):
    qs, qe = qs_all[:nq], qe_all[:nq]
    w_off, w_hits = t.find_batch(qs, qe)
    set_opt("ivl.fx_fill", fx); set_opt("ivl.bm_variant", variant); set_opt("ivl.sl_f", f)
    off, hits = ix.find(qs, qe)
    ok_off = np.array_equal(off, w_off)
    badq = []
    if ok_off:
        neq = hits != w_hits
        if neq.any():
            badq = sorted(set((np.searchsorted(w_off, np.nonzero(neq)[0], side="right") - 1).tolist()))
    print("nq=%d fx=%d variant=%d f=%d rep=%d: offsets %s, bad queries %d %s" % (nq, fx, variant, f, rep, ok_off, len(badq), badq[:12]), flush=True)
    return off, hits, w_off, w_hits, badq
off, hits, w_off, w_hits, badq = run(NQ)
for q in badq[:6]:
    g, w = hits[w_off[q]:w_off[q + 1]], w_hits[w_off[q]:w_off[q + 1]]
    # is what came out the list of another query?  overlaps of the right query?
    ov = [(int(e[h]) > int(qs_all[q]) and int(s[h]) < int(qe_all[q])) if 0 <= h < n else None for h in g.tolist()]
    owner = None
    for q2 in range(max(0, q - 900), min(NQ, q + 900)):
        w2 = w_hits[w_off[q2]:w_off[q2 + 1]]
        if q2 != q and len(w2) and len(g) and g[0] in w2:
            owner = q2; break
    print(" q", q, "qs", qs_all[q], "qe", qe_all[q], "got", g.tolist(), "want", w.tolist(), "overlap", ov, "first got hit belongs to query", owner, flush=True)
for rep in (1, 2): run(NQ, rep=rep)
run(NQ, fx=0)
for nq in (49152, 16384, 16384 + 848, 848, 32768 + 848):
    run(nq)
for f in (0, 2, 4, 6):
    run(NQ, f=f)
run(NQ, variant=2)
run(NQ, variant=1)
