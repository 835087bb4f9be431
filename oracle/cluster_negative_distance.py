#!/usr/bin/env python3
"""Does ClusterTree with a NEGATIVE max_dist have an answer to reproduce?  ("What's missing" 1.)

Runs the reference's own src/cluster.c (compiled in place by oracle/Makefile -> oracle/_ref/libcluster_ref.so; the same
binary bx.intervals.cluster wraps, lib/bx/intervals/cluster.pyx:57-73) on the same intervals
  (a) in different insertion orders with the node priorities fixed (srand(seed) before every tree: src/cluster.c:66-69
      draws them from the process-wide rand()), and
  (b) in ONE insertion order with different seeds,
and counts how many distinct getregions() results come out.  With max_dist >= 0 every run gives the same regions (the
connected components of "gap <= max_dist": what bxmi_ivl_clusters computes).  With max_dist < 0 an interval can lie
both "right of" and "left of" a cluster (src/cluster.c:224-232 tests start - max_dist > end first), merged clusters
stop being ordered, and the result depends on both the order and the priorities.

    python oracle/cluster_negative_distance.py            # prints a table; exits 1 if it finds NO dependence
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402

if not O.have_ref_cluster():
    sys.exit("oracle/_ref/libcluster_ref.so is missing: run `make -C oracle ref` where /root/reference exists")
libc = C.CDLL(None)


def regions(triples, max_dist, seed):
    libc.srand(seed)
    return tuple(sorted((s, e, tuple(ids)) for s, e, ids in O.ref_cluster_regions(triples, max_dist, 0)))


rng = np.random.default_rng(20260927)
rows = []
for max_dist in (5, 0, -1, -5, -20):
    by_order, by_seed, trials = 0, 0, 0
    for case in range(40):
        n = int(rng.integers(20, 120))
        s = rng.integers(0, 600, size=n)
        e = s + rng.integers(1, 60, size=n)
        base = [(int(a), int(b), i) for i, (a, b) in enumerate(zip(s, e))]
        seen_orders = {regions([base[i] for i in rng.permutation(n)], max_dist, 12345) for _ in range(12)}
        seen_seeds = {regions(base, max_dist, seed) for seed in range(12)}
        by_order += len(seen_orders) > 1
        by_seed += len(seen_seeds) > 1
        trials += 1
    rows.append((max_dist, trials, by_order, by_seed))
print("max_dist  interval sets  >1 result over 12 insertion orders (fixed priorities)  >1 result over 12 rand() seeds (fixed order)")
for max_dist, trials, by_order, by_seed in rows:
    print("%8d  %13d  %52d  %46d" % (max_dist, trials, by_order, by_seed))
bad_positive = any(r[2] or r[3] for r in rows if r[0] >= 0)
negative_depends = any(r[2] or r[3] for r in rows if r[0] < 0)
print("non-negative distances reproducible:", not bad_positive, "| negative distances depend on order / priorities:", negative_depends)
sys.exit(0 if (negative_depends and not bad_positive) else 1)
