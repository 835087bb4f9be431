#!/usr/bin/env python3
"""Per-call IntervalTree.find (the loop of scripts/interval_join.py:26-30) at several index sizes, the same workload on either side:
  PYTHONPATH=/tmp/bxref/lib python tools/per_call_crossover.py reference   (the real bx-python, in the build container)
  python tools/per_call_crossover.py dropin                                (this repo's bx package, on the GPU box)
Targets: uniform on a span of 2500 coordinates per target, at most 2e9 (0.3 hits per 500-wide query, more from 800 k targets on);
4000 timed calls after 200 warm ones.  Prints one JSON line {side, n: us_per_call}."""
import json
import os
import sys
import time

import numpy as np

side = sys.argv[1] if len(sys.argv) > 1 else "dropin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if side == "dropin":
    sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import bx.intervals.intersection as bi  # noqa: E402

sizes = [int(x) for x in os.environ.get("SIZES", "1000 10000 100000 1000000 4000000").split()]
out = {"side": side, "module": bi.__file__, "us_per_call": {}, "hits_per_call": {}, "build_s": {}}
for n in sizes:
    rng = np.random.default_rng(n)
    span = min(n * 2500, 2_000_000_000)
    s = rng.integers(0, span, size=n)
    e = s + rng.integers(1, 1000, size=n)
    t0 = time.perf_counter()
    t = bi.IntervalTree()
    ins = t.insert  # (row by row on both sides: what the unmodified script does)
    for a, b, i in zip(s.tolist(), e.tolist(), range(n)):
        ins(a, b, i)
    t.find(1, 2)
    out["build_s"][str(n)] = round(time.perf_counter() - t0, 3)
    q = rng.integers(0, span, size=4200).tolist()
    find = t.find
    for x in q[:200]:
        find(x, x + 500)
    hits = 0
    t0 = time.perf_counter()
    for x in q[200:]:
        hits += len(find(x, x + 500))
    dt = time.perf_counter() - t0
    out["us_per_call"][str(n)] = round(dt / 4000 * 1e6, 2)
    out["hits_per_call"][str(n)] = round(hits / 4000, 2)
    del t
print(json.dumps(out))
