#!/usr/bin/env python3
"""Per-call find (bxmi_ivl_find_one: answer polled from host memory) against the batched find on the same queries:
a visibility bug between the kernel's hit stores and its completion word would show up as a differing hit list.
N env = number of calls (default 100000)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import numpy as np

from bxmi.intervals import IntervalIndex

rng = np.random.default_rng(3)
n = 300_000
s = rng.integers(0, 3_000_000, size=n).astype(np.int32)
e = (s + rng.integers(1, 4000, size=n)).astype(np.int32)  # ~250 hits per query: every wave of the kernel writes hits
ix = IntervalIndex()
ix.append(s, e)
ix.seal()
N = int(os.environ.get("N", 100_000))
qs = rng.integers(0, 3_000_000, size=N).astype(np.int32)
qe = (qs + rng.integers(1, 1500, size=N)).astype(np.int32)
off, hits = ix.find(qs, qe)
bad = 0
for i in range(N):
    got = ix.find_one_list(int(qs[i]), int(qe[i]))
    if got != hits[off[i]:off[i + 1]].tolist():
        bad += 1
        if bad < 4:
            print("MISMATCH at call", i, len(got), int(off[i + 1] - off[i]))
print("find_one stress: %d calls, %d mismatches, %.0f hits per call" % (N, bad, float(off[-1]) / N))
sys.exit(1 if bad else 0)
