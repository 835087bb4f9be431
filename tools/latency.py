#!/usr/bin/env python3
"""Per-call latency of the drop-in API (what an unmodified script pays per line)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import numpy as np
import bx.bitset, bx.intervals
rng = np.random.default_rng(1)
t = bx.intervals.IntervalTree()
s = rng.integers(0, 10_000_000, size=200_000); e = s + rng.integers(1, 1000, size=len(s))
t.insert_batch(s, e, list(range(len(s))))
t.find(1, 2)
q = rng.integers(0, 10_000_000, size=2000).tolist()
t0 = time.perf_counter()
for x in q: t.find(x, x + 500)
print("IntervalTree.find       %.1f us/call" % ((time.perf_counter() - t0) / len(q) * 1e6))
b = bx.bitset.BinnedBitSet()
for i in range(0, 20000): b.set_range(int(s[i]), int(e[i] - s[i]))
b.count_range(0, 10)
t0 = time.perf_counter()
for x in q: b.count_range(x, 500)
print("BinnedBitSet.count_range %.1f us/call" % ((time.perf_counter() - t0) / len(q) * 1e6))
t0 = time.perf_counter()
for i in range(20000, 40000): b.set_range(int(s[i]), int(e[i] - s[i]))
b.count_range(0, 1)
print("BinnedBitSet.set_range   %.2f us/call (queued, one flush)" % ((time.perf_counter() - t0) / 20000 * 1e6))
t0 = time.perf_counter(); end = 0; n = 0
while True:
    st = b.next_set(end)
    if st == b.size: break
    end = b.next_clear(st); n += 1
print("next_set/next_clear walk %.2f us/run over %d runs" % ((time.perf_counter() - t0) / max(n, 1) * 1e6, n))
