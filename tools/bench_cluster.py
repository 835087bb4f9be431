#!/usr/bin/env python3
"""ClusterTree on the device vs the CPU restatement: N intervals (env N, default 20M), one chromosome-like span."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import numpy as np

from bxmi.intervals import IntervalIndex

N = int(os.environ.get("N", 20_000_000))
rng = np.random.default_rng(3)
s = rng.integers(0, 2_000_000_000, size=N).astype(np.int32)
e = (s.astype(np.int64) + rng.integers(1, 200, size=N)).clip(max=2**31 - 1).astype(np.int32)
ids = np.arange(N, dtype=np.int32)
ix = IntervalIndex()
t0 = time.perf_counter()
ix.append(s, e)
ix.seal()
seal_s = time.perf_counter() - t0
ix.clusters(50, ids)
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    cs, ce, off, mem = ix.clusters(50, ids)
dev_s = (time.perf_counter() - t0) / reps
out = dict(n=N, max_dist=50, clusters=int(len(cs)), seal_s=round(seal_s, 3), clusters_s=round(dev_s, 4),
           m_intervals_per_s=round(N / dev_s / 1e6, 1), note="clusters_s includes the id upload and the D2H of all four result arrays")
if os.environ.get("CPU", "1") == "1":
    from oracle import oracle as O

    m = min(N, 2_000_000)
    t0 = time.perf_counter()
    want = O.cluster_regions(s[:m], e[:m], ids[:m], 50, 0)
    cpu_s = time.perf_counter() - t0
    sub = IntervalIndex()
    sub.append(s[:m], e[:m])
    a, b, o2, m2 = sub.clusters(50, ids[:m])
    out.update(cpu_port_m_intervals_per_s=round(m / cpu_s / 1e6, 2), cpu_sample=m,
               agrees_with_cpu=bool(a.tolist() == [w[0] for w in want] and m2.tolist() == [i for w in want for i in w[2]]))
print(json.dumps(out))
