#!/usr/bin/env python3
"""End-to-end wall time of the CLI counterparts on a big synthetic BED pair, bulk ingest vs per-line ingest."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import numpy as np
from bxmi import synth
N = int(os.environ.get("N", 2_000_000))
d = tempfile.mkdtemp()
fa, fb = os.path.join(d, "q.bed"), os.path.join(d, "t.bed")
for path, seed, tag in ((fa, 11, "q"), (fb, 12, "t")):
    rng = np.random.default_rng(seed)
    chroms = list(synth.HG19_SIZES)
    ch = rng.integers(0, len(chroms), size=N)
    s = rng.integers(0, 40_000_000, size=N)
    e = s + rng.integers(1, 1000, size=N)
    with open(path, "w") as f:
        f.writelines("%s\t%d\t%d\t%s%d\t0\t+\n" % (chroms[c], a, b, tag, i) for i, (c, a, b) in enumerate(zip(ch.tolist(), s.tolist(), e.tolist())))
env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "bx-python_amd") + os.pathsep + ROOT, PYTHONWARNINGS="ignore")
for mod, args in (("bed_coverage", [fb]), ("bed_intersect", [fa, fb]), ("bed_intersect_basewise", [fa, fb])):
    row = []
    for extra in ({}, {"BXMI_NO_FASTPARSE": "1"}):
        t0 = time.perf_counter()
        p = subprocess.run([sys.executable, "-m", "bxmi.cli." + mod] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(env, **extra))
        row.append((time.perf_counter() - t0, len(p.stdout), p.returncode))
    print("%-24s N=%d  bulk %.2f s   per-line %.2f s   (same output: %s)" % (mod, N, row[0][0], row[1][0], row[0][1:] == row[1][1:]))
