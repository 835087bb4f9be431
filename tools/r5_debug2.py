#!/usr/bin/env python3
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
want = {}
tot_bad = 0
for rep in range(3):
    for nq, f, variant in ((NQ, -1, -1), (49152, -1, -1), (NQ, 0, -1), (NQ, 2, -1), (NQ, -1, 2), (33616, 1, -1)):
        qs, qe = qs_all[:nq], qe_all[:nq]
        if nq not in want: want[nq] = t.find_batch(qs, qe)
        w_off, w_hits = want[nq]
        set_opt("ivl.bm_variant", variant); set_opt("ivl.sl_f", f)
        off, hits = ix.find(qs, qe)
        ok = np.array_equal(off, w_off)
        nb = int((hits != w_hits).sum()) if ok else -1
        tot_bad += abs(nb)
        print("  rep %d nq=%d f=%d variant=%d: offsets %s bad hits %d" % (rep, nq, f, variant, ok, nb), flush=True)
print("TOTAL BAD", tot_bad)
