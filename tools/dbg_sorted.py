#!/usr/bin/env python3
"""debug: the sorted-batch case of test_sorted_batches_skip_the_bucketing through the flat walk; prints where counts differ"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import numpy as np
from bxmi import _ffi
from bxmi.intervals import IntervalIndex
from oracle import oracle as O

def rc(rng, n, span, zero_frac=0.0, rev_frac=0.0, lmax=50):
    s = rng.integers(-span, span, size=n); ln = rng.integers(1, lmax + 1, size=n)
    ln[rng.random(n) < zero_frac] = 0; e = s + ln
    flip = rng.random(n) < rev_frac
    s, e = np.where(flip, e, s), np.where(flip, s, e)
    return s.astype(np.int32), e.astype(np.int32)

n, nq, span, lmax = 300000, 100000, 3_000_000, 2000
rng = np.random.default_rng(n + nq)
s, e = rc(rng, n, span, 0.05, 0, lmax)
t = O.OracleIntervalTree(); t.insert_many_arrays(s, e)
ix = IntervalIndex(); ix.append(s, e); ix.seal()
qs, qe = rc(rng, nq, span, 0.1, 0.05, lmax * 2)
order = np.argsort(qs, kind="stable"); qs, qe = qs[order].copy(), qe[order].copy()
qe[::101] = 2**31 - 1; qe[::103] = -(2**31); qs[-3:] = 2**31 - 1; qe[-3:] = [2**31 - 1, 0, -(2**31)]
want_c, want_t = t.count_batch(qs, qe)
def so(k, v): _ffi.call("bxmi_set_option", k.encode(), int(v))
so("ivl.partition", 1); so("ivl.sorted_path", 0)
for bw in (0, 1):
    so("ivl.bw", bw)
    c, tot = ix.count(qs, qe)
    bad = np.nonzero(c != want_c)[0]
    print("bw", bw, "mismatches", len(bad), "total ok", tot == want_t, "flat", ix.flat_state())
    if len(bad):
        print(" first", bad[:20].tolist(), "\n got", c[bad[:20]].tolist(), "\n want", want_c[bad[:20]].tolist(), "\n qs", qs[bad[:20]].tolist(), "\n qe", qe[bad[:20]].tolist())
        d = np.diff(bad); print(" runs of bad:", len(bad), "span", bad[0], bad[-1], "tiles", sorted(set((bad >> 14).tolist())))
