#!/usr/bin/env python3
"""One-off randomized differential run of the interval engine against the CPU oracle (not part of the test suite):
many target / query distributions, every count and find path forced on in turn (direct kernels, round 1's bucketed pass,
the large-batch pass on bucket images and on key slices with random tile shapes / unit sizes / run widths, find through
the exchange), sorted and unsorted batches.
ROUNDS env (default 20).  Prints the first disagreement and exits 1, or a summary line.
The CPU oracle dominates the run time and its cost grows with the number of HITS: every round is sized so that the
expected overlaps stay below ~20 M (an unbounded version of this script once spent a whole GPU allowance waiting for
the oracle) -- run it under `timeout` all the same."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import numpy as np

from bxmi import _ffi
from bxmi.intervals import IntervalIndex
from oracle import oracle as O


def opt(k, v):
    _ffi.call("bxmi_set_option", k.encode(), int(v))


rng = np.random.default_rng(int(os.environ.get("SEED", 12345)))
rounds = int(os.environ.get("ROUNDS", 20))
MAX_HITS = 20_000_000
checked = 0
sparse_served = 0
clumped_served = 0
was_clumped = False
for r in range(rounds):
    n = int(rng.choice([5000, 20000, 100000, 400000]))
    nq = int(rng.choice([3000, 20000, 70000]))
    span = int(rng.choice([2000, 10**5, 10**7, 2 * 10**9]))
    lmax = int(rng.choice([1, 5, 200, 5000, 10**6]))
    # expected overlaps per query ~ n * (target length + query length) / span: shrink the lengths until the round is affordable
    while n * nq * (1.5 * lmax + 1) / span > MAX_HITS and lmax > 1:
        lmax //= 2
    if n * nq * (1.5 * lmax + 1) / span > MAX_HITS:
        nq = max(1000, int(MAX_HITS * span / (n * (1.5 * lmax + 1))))
    clump = rng.random() < 0.3 and n * nq / 2 * 0.5 < MAX_HITS * 50
    s = rng.integers(-span // 2, span // 2, size=n)
    if clump:  # a tenth of the targets piled on a small stretch (queries there see all of them)
        s[: n // 10] = rng.integers(0, max(span // 1000, 2), size=n // 10)
    dups = not clump and rng.random() < 0.25
    if dups:  # duplicate-heavy targets (what real BED tracks look like): starts drawn from n / 8 coordinates around hot spots
        pool = np.sort(rng.integers(-span // 2, span // 2, size=max(n // 8, 16)))
        s = pool[rng.integers(0, len(pool), size=n)]
    ln = rng.integers(0, lmax + 1, size=n)
    if dups:
        ln = rng.integers(0, lmax + 1, size=12)[rng.integers(0, 12, size=n)]  # a dozen lengths: the ends pile up too
    e = np.minimum(s + ln, 2**31 - 1)
    qs = rng.integers(-span // 2 - 10, span // 2 + 10, size=nq)
    qe = np.minimum(qs + rng.integers(0, 2 * lmax + 2, size=nq), 2**31 - 1)
    flip = rng.random(nq) < 0.03
    qs, qe = np.where(flip, qe, qs), np.where(flip, qs, qe)
    if rng.random() < 0.4:
        o = np.argsort(qs, kind="stable")
        qs, qe = qs[o], qe[o]
    s, e, qs, qe = (a.astype(np.int32) for a in (s, e, qs, qe))
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    want_c, want_t = t.count_batch(qs, qe)
    ix = IntervalIndex()
    ix.append(s, e)
    opt("ivl.bitmap_min", 1)
    # (partition, -, bitmap, slice): direct kernel, round 1's pass, the large-batch pass
    # on images first / slices first / slices only, each with random tile shapes, unit sizes and run widths
    # sparse = 1: offset-cell images whatever the density (with ivl.bm_hard_ppm opened up: cells with more than five keys, their
    # lists and the searches behind them are then the rule, not the exception)
    # sparse = 2: offset cells in the CLUMPED layout (ivl.clumped = 1: a rank table per hard cell) wherever the tables fit the LDS
    for part, cells, bitmap, slices, flat, dense, sparse in ((0, 1, 0, 0, -1, -1, -1), (1, 1, 0, 0, -1, -1, -1), (1, 0, 0, 0, -1, -1, -1), (1, 1, -1, 0, 1, 1, -1),
                                                             (1, 1, -1, 0, 0, 1, -1), (1, 1, -1, -1, -1, -1, -1), (1, 1, -1, 1, 0, 0, -1), (1, 1, -1, 1, 0, 0, -1),
                                                             (1, 1, -1, -1, 1, 1, -1), (1, 1, -1, -1, -1, -1, 1), (1, 1, -1, -1, -1, -1, 1),
                                                             (1, 1, -1, -1, 0, -1, 2), (1, 1, -1, -1, -1, -1, 2)):
        knobs = dict(variant=int(rng.integers(-1, 3)), f=int(rng.integers(-1, 7)), lanes=int(rng.choice([0, 16, 64, 1])),
                     sorted_path=int(rng.integers(0, 2)), cell_log2=int(rng.choice([0, 6, 7, 8])))
        opt("ivl.clumped", 1 if sparse == 2 else -1)
        if sparse == 2:
            sparse = -1
            ix.seal()  # (the offset-cell images of a sealed index are built once, in one layout)
            was_clumped = True
        elif was_clumped:
            ix.seal()
            was_clumped = False
        opt("ivl.sparse", sparse)
        opt("ivl.bo_cell_log2", knobs["cell_log2"] if sparse == 1 else 0)
        opt("ivl.bm_hard_ppm", 1000000 if sparse == 1 and rng.random() < 0.7 else 2000)
        if sparse == 1:
            ix.seal()  # (the images are built once per sealed index: another width needs them again)
        opt("ivl.partition", part)
        opt("ivl.bitmap", bitmap)
        opt("ivl.slice", slices)
        opt("ivl.flat", flat)    # 1: cell images of units wherever the index qualifies (the persistent walk)
        opt("ivl.dense", dense)  # 1: dense unit images wherever it qualifies
        opt("ivl.bd_w8", int(rng.integers(-1, 2)))
        opt("ivl.bd_chunk", int(rng.choice([0, 1024, 20000])))
        opt("ivl.bm_variant", knobs["variant"])
        opt("ivl.sl_f", knobs["f"])
        opt("ivl.sl_lanes", knobs["lanes"])
        opt("ivl.sorted_path", knobs["sorted_path"])
        got_c, got_t = ix.count(qs, qe)
        clumped_served += ix.sparse_state()[0] == 2
        # the same batch asking for the TOTAL only (counts = NULL): on cell images the walk keeps the totals itself (ivl.tot_walk)
        opt("ivl.tot_walk", int(rng.integers(0, 2)))
        only_t = ix.count(qs, qe, want_counts=False)[1]
        opt("ivl.tot_walk", 1)
        if only_t != want_t:
            print("TOTAL-ONLY MISMATCH round", r, dict(n=n, nq=nq, span=span, lmax=lmax, clump=clump, part=part, cells=cells, bitmap=bitmap, slices=slices, flat=flat,
                                                       dense=dense, sparse=sparse, **knobs), ix.flat_state(), ix.sparse_state(), only_t, want_t)
            sys.exit(1)
        if not np.array_equal(got_c, want_c) or got_t != want_t:
            bad = np.nonzero(got_c != want_c)[0][:5]
            print("MISMATCH round", r, dict(n=n, nq=nq, span=span, lmax=lmax, clump=clump, part=part, cells=cells, bitmap=bitmap, slices=slices, flat=flat, dense=dense,
                                            sparse=sparse, **knobs),
                  ix.flat_state(), ix.dense_state(), ix.slice_state(), ix.sparse_state(), bad, qs[bad], qe[bad], got_c[bad], want_c[bad])
            sys.exit(1)
    opt("ivl.bitmap", -1)
    opt("ivl.flat", -1), opt("ivl.dense", -1), opt("ivl.bd_w8", -1), opt("ivl.bd_chunk", 0)
    opt("ivl.sparse", -1), opt("ivl.bo_cell_log2", 0), opt("ivl.bm_hard_ppm", 2000), opt("ivl.clumped", -1)
    sparse_served += ix.sparse_state()[0] == 1
    m = min(nq, 20000)
    w_off, w_hits = t.find_batch(qs[:m], qe[:m])
    # direct kernels, the bucketed find, find through the exchange (the fill straight into the list or through scratch and the copy,
    # other tile shapes and unit sizes); sorted batches take the staged kernels
    for part, sliced in ((0, 0), (1, 0), (1, 1), (1, 1), (1, 1)):
        knobs = dict(variant=int(rng.integers(-1, 3)), f=int(rng.integers(-1, 7)), lanes=int(rng.choice([0, 16, 64])), sorted_path=int(rng.integers(0, 2)),
                     direct=int(rng.integers(0, 2)))
        opt("ivl.partition", part)
        opt("ivl.find_sliced", sliced)
        opt("ivl.fx_direct", knobs["direct"])
        opt("ivl.slice", -1)
        opt("ivl.bm_variant", knobs["variant"])
        opt("ivl.sl_f", knobs["f"])
        opt("ivl.sl_lanes", knobs["lanes"])
        opt("ivl.sorted_path", knobs["sorted_path"])
        off, hits = ix.find(qs[:m], qe[:m])
        if not (np.array_equal(off, w_off) and np.array_equal(hits, w_hits)):
            print("FIND MISMATCH round", r, dict(n=n, nq=nq, span=span, lmax=lmax, clump=clump, part=part, sliced=sliced, **knobs), ix.slice_state())
            sys.exit(1)
    for k, v in (("ivl.partition", -1), ("ivl.find_sliced", 1), ("ivl.bm_variant", -1), ("ivl.sl_f", -1), ("ivl.sl_lanes", 0),
                 ("ivl.sorted_path", 1), ("ivl.fx_direct", -1)):
        opt(k, v)
    checked += 1
    ix.close()
print("fuzz: %d rounds (%d with offset-cell images, %d batches on the clumped layout), all counts and hit lists equal the oracle" % (checked, sparse_served, clumped_served))
