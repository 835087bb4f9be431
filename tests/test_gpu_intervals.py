"""
GPU parity tests of the interval path (run on the MI355X box with -m gpu).

Everything goes through the C ABI (libbxmi.so via bxmi._ffi); the checker is
the CPU oracle (oracle/ivtree.c) and the committed reference-generated vectors.
Bit-exact: counts, CSR offsets, hit order.
"""
import hashlib

import numpy as np
import pytest

from bxmi import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


@pytest.fixture(scope="module")
def IntervalIndex():
    from bxmi.intervals import IntervalIndex

    return IntervalIndex


def set_opt(key, value):
    from bxmi import _ffi

    _ffi.call("bxmi_set_option", key.encode(), int(value))


def _library_defaults():
    """every knob's value as the library starts with (bxmi_option_at): read at import, before any test turns one"""
    from bxmi import _ffi

    return _ffi.options()


DEFAULT_OPTS = _library_defaults()


def reset_opts():
    for k, v in DEFAULT_OPTS.items():
        set_opt(k, v)


def _erange_contract(ix, qs, qe, want_off):
    """bxmi_ivl_find with a hit buffer that is too small: BXMI_ERANGE, the offsets and the total valid, the buffer untouched."""
    import ctypes as C

    from bxmi import _ffi

    qs, qe = np.ascontiguousarray(qs, dtype=np.int32), np.ascontiguousarray(qe, dtype=np.int32)
    cap = max(1, int(want_off[-1]) // 3)
    offsets = np.full(len(qs) + 1, -7, dtype=np.int64)
    hits = np.full(cap, -5, dtype=np.int32)
    total = C.c_int64(0)
    rc = _ffi.call("bxmi_ivl_find", ix._h, _ffi.ptr(qs), _ffi.ptr(qe), len(qs), _ffi.ptr(offsets), _ffi.ptr(hits), cap, C.byref(total), allow=(_ffi.ERANGE,))
    assert rc == _ffi.ERANGE and total.value == int(want_off[-1])
    assert np.array_equal(offsets, want_off), np.nonzero(offsets != want_off)[0][:8]
    assert (hits == -5).all()


def make_index(IntervalIndex, starts, ends):
    ix = IntervalIndex()
    ix.append(starts, ends)
    ix.seal()
    return ix


# ------------------------------------------------------------------ golden --
def test_order_matches_reference_traverse(golden_trees, IntervalIndex):
    for case in golden_trees:
        ix = make_index(IntervalIndex, case["starts"], case["ends"])
        assert ix.order().tolist() == case["order"], (case["mode"], case["n"])


def test_count_and_find_match_reference_vectors(golden_trees, IntervalIndex):
    for _once in (0,):
        for case in golden_trees:
            ix = make_index(IntervalIndex, case["starts"], case["ends"])
            q = np.array(case["queries"], dtype=np.int32)
            counts, total = ix.count(q[:, 0], q[:, 1])
            want = [len(h) for h in case["hits"]]
            assert counts.tolist() == want, (case["mode"], case["n"])
            assert total == sum(want)
            set_opt("ivl.partition", 1)
            try:
                counts, total = ix.count(q[:, 0], q[:, 1])
                p_offs, p_hits = ix.find(q[:, 0], q[:, 1])
            finally:
                set_opt("ivl.partition", -1)
            assert counts.tolist() == want and total == sum(want), ("partitioned", case["mode"], case["n"])
            assert p_hits.tolist() == [x for h in case["hits"] for x in h], ("partitioned find", case["mode"], case["n"])
            assert p_offs.tolist() == np.concatenate([[0], np.cumsum(want)]).tolist()
            offs, hits = ix.find(q[:, 0], q[:, 1])
            assert offs.tolist() == np.concatenate([[0], np.cumsum(want)]).tolist()
            assert hits.tolist() == [x for h in case["hits"] for x in h], (case["mode"], case["n"])


def test_neighbours_match_reference_vectors(golden_trees, IntervalIndex):
    import operator

    n = 0
    for case in golden_trees:
        if not case["neighbours"]:
            continue
        ix = make_index(IntervalIndex, case["starts"], case["ends"])
        s, e = case["starts"], case["ends"]
        for kind, pos, k, md, want in case["neighbours"]:
            cand = ix.neighbors(pos, md, -1 if kind == "before" else +1).tolist()
            if len(cand) != k:  # intersection.pyx:242-245 / :257-260
                cand = sorted(cand, key=(lambda i: e[i]) if kind == "before" else (lambda i: s[i]), reverse=kind == "before")[:k]
            assert cand == want, (kind, pos, k, md)
            n += 1
    assert n > 300


def test_before_with_reversed_targets_reports_every_interval_whose_end_qualifies(IntervalIndex):
    """An index holding intervals with start > end: the reference's left() prunes by subtree (`minstart > position`,
    intersection.pyx:196-197), so whether a reversed interval whose start lies right of the position is reported depends
    on the treap's random shape.  The engine reports every interval whose END qualifies (0 <= position - 1 - end <
    max_dist) -- the union of what the reference can report -- and after() is untouched."""
    rng = np.random.default_rng(44)
    n = 3000
    s = rng.integers(0, 100_000, size=n).astype(np.int32)
    e = (s + rng.integers(1, 400, size=n)).astype(np.int32)
    flip = rng.random(n) < 0.1
    s2, e2 = np.where(flip, e, s).astype(np.int32), np.where(flip, s, e).astype(np.int32)
    ix = make_index(IntervalIndex, s2, e2)
    assert ix.has_reversed
    for pos, md in ((50_000, 2500), (10, 100), (99_000, 50_000), (70_123, 1)):
        got = sorted(ix.neighbors(pos, md, -1).tolist())
        p = pos - 1
        want = sorted(np.nonzero((p - e2.astype(np.int64) >= 0) & (p - e2.astype(np.int64) < md))[0].tolist())
        assert got == want, (pos, md, len(got), len(want))
        got = sorted(ix.neighbors(pos, md, +1).tolist())
        q = pos + 1
        want = sorted(np.nonzero((s2.astype(np.int64) - q >= 0) & (s2.astype(np.int64) - q < md))[0].tolist())
        assert got == want, ("after", pos, md)


def test_find_one_beyond_its_buffer(IntervalIndex):
    """More hits than the 16 KiB host-visible result buffer holds: falls over to the batched path."""
    n = 10000
    s = np.arange(n, dtype=np.int32)
    ix = make_index(IntervalIndex, s, s + 100000)
    assert ix.find_one(50, 60).tolist() == list(range(60))
    assert ix.find_one(0, 200000).tolist() == list(range(n))
    assert ix.find_one(5, 5).tolist() == list(range(5)) and ix.find_one(7, 3).tolist() == list(range(3))


def test_empty_index(IntervalIndex):
    ix = IntervalIndex()
    ix.seal()
    assert ix.find_one(1, 10).tolist() == []
    c, t = ix.count([1, 5], [10, 5])
    assert c.tolist() == [0, 0] and t == 0
    offs, hits = ix.find([1], [10])
    assert offs.tolist() == [0, 0] and len(hits) == 0
    assert ix.order().tolist() == []


# ------------------------------------------------------- oracle differential --
def _random_case(rng, n, span, zero_frac=0.0, rev_frac=0.0, lmax=50):
    s = rng.integers(-span, span, size=n)
    ln = rng.integers(1, lmax + 1, size=n)
    ln[rng.random(n) < zero_frac] = 0
    e = s + ln
    flip = rng.random(n) < rev_frac
    s, e = np.where(flip, e, s), np.where(flip, s, e)
    return s.astype(np.int32), e.astype(np.int32)


@pytest.mark.parametrize(
    "n,span,zero,rev,lmax",
    [(1, 10, 0, 0, 5), (31, 40, 0.2, 0, 8), (32, 40, 0, 0, 8), (33, 40, 0, 0.2, 8), (1023, 500, 0.1, 0, 30),
     (1025, 100, 0.3, 0.1, 10), (40000, 100000, 0.05, 0, 200), (200000, 3000, 0.1, 0, 20), (70000, 10**9, 0, 0, 10**6),
     (5000, 2**31 - 1100, 0.05, 0, 1000), (3000, 5, 0.5, 0, 2)],  # the whole int32 line; everything piled on 10 coordinates
)
def test_random_differential(O, IntervalIndex, n, span, zero, rev, lmax):
    rng = np.random.default_rng(n * 7 + span)
    s, e = _random_case(rng, n, span, zero, rev, lmax)
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    ix = make_index(IntervalIndex, s, e)
    assert ix.has_reversed == bool((e < s).any())
    assert ix.order().tolist() == t.traverse().tolist()
    nq = 3000
    qs, qe = _random_case(rng, nq, span, 0.1, 0.1, lmax * 2)
    qs[:5] = [-(2**31), 2**31 - 1, -(2**31), 2**31 - 1, 0]  # extreme coordinates
    qe[:5] = [2**31 - 1, 2**31 - 1, -(2**31), -(2**31), 0]
    want_c, want_t = t.count_batch(qs, qe)
    got_c, got_t = ix.count(qs, qe)
    bad = np.nonzero(got_c != want_c)[0]
    assert len(bad) == 0, (bad[:5], qs[bad[:5]], qe[bad[:5]], got_c[bad[:5]], want_c[bad[:5]])
    assert got_t == want_t
    set_opt("ivl.partition", 1)  # same batch through the large-batch paths
    try:
        fl_c, fl_t = ix.count(qs, qe)  # the flat walk on cell images where the index qualifies, else as the next line
        set_opt("ivl.flat", 0)
        dn_c, dn_t = ix.count(qs, qe)  # dense unit images where the index qualifies, else as the next line
        set_opt("ivl.dense", 0)
        bm_c, bm_t = ix.count(qs, qe)  # neither kind of image: key slices where they fit (else identical to the next line)
        set_opt("ivl.bitmap", 0)       # the bucketed search pass
        got_c, got_t = ix.count(qs, qe)
        tot_only = ix.count(qs, qe, want_counts=False)[1]
        if not ix.has_reversed:
            p_off, p_hits = ix.find(qs, qe)
    finally:
        reset_opts()
    bad = np.nonzero(fl_c != want_c)[0]
    assert len(bad) == 0 and fl_t == want_t, ("flat walk on cells", ix.flat_state(), bad[:5], qs[bad[:5]], qe[bad[:5]], fl_c[bad[:5]], want_c[bad[:5]])
    bad = np.nonzero(dn_c != want_c)[0]
    assert len(bad) == 0 and dn_t == want_t, ("dense pass", ix.dense_state(), bad[:5], qs[bad[:5]], qe[bad[:5]], dn_c[bad[:5]], want_c[bad[:5]])
    bad = np.nonzero(bm_c != want_c)[0]
    assert len(bad) == 0 and bm_t == want_t, ("slices / first-generation pass", ix.slice_state(), bad[:5], qs[bad[:5]], qe[bad[:5]], bm_c[bad[:5]], want_c[bad[:5]])
    if not ix.has_reversed:
        w_off, w_hits = t.find_batch(qs, qe)
        assert np.array_equal(p_off, w_off) and np.array_equal(p_hits, w_hits), "partitioned find"
    bad = np.nonzero(got_c != want_c)[0]
    assert len(bad) == 0, ("partitioned", bad[:5], qs[bad[:5]], qe[bad[:5]], got_c[bad[:5]], want_c[bad[:5]])
    assert got_t == want_t == tot_only
    want_off, want_hits = t.find_batch(qs, qe)
    got_off, got_hits = ix.find(qs, qe)
    assert np.array_equal(got_off, want_off)
    assert np.array_equal(got_hits, want_hits)
    for i in list(range(8)) + list(range(100, 160)):  # the one-query latency path (per-call find() of the drop-in)
        assert ix.find_one(int(qs[i]), int(qe[i])).tolist() == want_hits[want_off[i]:want_off[i + 1]].tolist(), i


@pytest.mark.parametrize("n,nq,span,lmax", [(5000, 40000, 200000, 300), (300000, 100000, 3_000_000, 2000), (300000, 50000, 2**31 - 3_000_000, 10**6),
                                            (40, 33000, 1000, 50)])
def test_sorted_batches_skip_the_bucketing(O, IntervalIndex, n, nq, span, lmax):
    """Queries whose starts are non-decreasing take the one-pass local kernel (detected on the device); a single
    descent sends the same batch down the bucketed path.  Both must equal the reference, as must the direct kernel."""
    rng = np.random.default_rng(n + nq)
    s, e = _random_case(rng, n, span, 0.05, 0, lmax)
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    ix = make_index(IntervalIndex, s, e)
    qs, qe = _random_case(rng, nq, span, 0.1, 0.05, lmax * 2)
    order = np.argsort(qs, kind="stable")
    qs, qe = qs[order].copy(), qe[order].copy()
    qe[::101] = 2**31 - 1   # far-away ends leave the staged slice
    qe[::103] = -(2**31)
    qs[-3:] = 2**31 - 1     # still sorted; INT_MAX starts
    qe[-3:] = [2**31 - 1, 0, -(2**31)]
    want_c, want_t = t.count_batch(qs, qe)
    set_opt("ivl.partition", 1)
    try:
        loc_c, loc_t = ix.count(qs, qe)  # the order check of the bitmap-cell pass hands a sorted batch to the local kernel
        set_opt("ivl.sorted_path", 0)
        fl_c, fl_t = ix.count(qs, qe)  # the flat walk on cell images (where the index qualifies): long runs, the cooperative finish
        set_opt("ivl.flat", 0)
        dn_c, dn_t = ix.count(qs, qe)  # the same on dense unit images
        set_opt("ivl.dense", 0)
        bm_c, bm_t = ix.count(qs, qe)  # bitmap-cell pass (where the index qualifies): long runs, one bucket per tile
        set_opt("ivl.sorted_path", 1)
        set_opt("ivl.bitmap", 0)
        got_c, got_t = ix.count(qs, qe)
        tot_only = ix.count(qs, qe, want_counts=False)[1]
        f_off, f_hits = ix.find(qs, qe)  # windows in query order, no bucketing
        set_opt("ivl.sorted_path", 0)  # same batch, bucketed
        b_c, b_t = ix.count(qs, qe)
        g_off, g_hits = ix.find(qs, qe)  # bucketed find of a sorted batch (wave-aggregated ranks in the scatter)
        set_opt("ivl.sorted_path", 1)
        qs2, qe2 = qs.copy(), qe.copy()  # one descent in the middle: not sorted any more
        m = nq // 2
        qs2[m], qs2[m + 1] = qs[m + 1] + 1, qs[m] - 1
        u_c, u_t = ix.count(qs2, qe2)
    finally:
        reset_opts()
    assert np.array_equal(fl_c, want_c) and fl_t == want_t, ("flat walk on a sorted batch", ix.flat_state())
    assert np.array_equal(dn_c, want_c) and dn_t == want_t, ("dense pass on a sorted batch", ix.dense_state())
    bad = np.nonzero(bm_c != want_c)[0]
    assert len(bad) == 0 and bm_t == want_t, ("slices / first-generation pass", ix.slice_state(), bad[:5], qs[bad[:5]], qe[bad[:5]], bm_c[bad[:5]], want_c[bad[:5]])
    assert np.array_equal(loc_c, want_c) and loc_t == want_t, "sorted batch behind the bitmap pass's order check"
    bad = np.nonzero(got_c != want_c)[0]
    assert len(bad) == 0, ("sorted path", bad[:5], qs[bad[:5]], qe[bad[:5]], got_c[bad[:5]], want_c[bad[:5]])
    assert got_t == want_t == tot_only
    assert np.array_equal(b_c, want_c) and b_t == want_t
    w_off, w_hits = t.find_batch(qs, qe)
    assert np.array_equal(f_off, w_off) and np.array_equal(f_hits, w_hits), "find on a sorted batch"
    assert np.array_equal(g_off, w_off) and np.array_equal(g_hits, w_hits), "bucketed find on a sorted batch"
    want2_c, want2_t = t.count_batch(qs2, qe2)
    assert np.array_equal(u_c, want2_c) and u_t == want2_t


@pytest.mark.parametrize("stage", ["flat", "sparse"])
def test_sorted_batches_on_cell_images(O, IntervalIndex, stage):
    """A sorted batch on an index with cell images is answered straight from the images, stretch by stretch (bs_plan / bs_walk):
    a batch crowded into one unit (stretches cut into several items), queries left and right of the grid, longer than a record
    holds, zero-length and reversed ones, a pile of 70 000 identical targets (counts beyond 16 bits), a length that is not a
    multiple of four, totals without counts -- against the oracle treap, and against the first-generation kernel for sorted
    batches (ivl.sorted_cells = 0)."""
    rng = np.random.default_rng(1234 + (stage == "sparse"))
    n, span = (300_000, 6_000_000) if stage == "flat" else (200_000, 80_000_000)
    s = rng.integers(1000, span, size=n)
    e = s + rng.integers(1, 1500, size=n)
    s[:70_000] = span // 3
    e[:70_000] = span // 3 + 40
    nq = 400_003
    qs = rng.integers(-5000, span + 20_000, size=nq)
    qs[: nq // 2] = rng.integers(span // 3 - 3000, span // 3 + 3000, size=nq // 2)  # half of the batch in one unit, on the pile
    qe = qs + rng.integers(1, 3000, size=nq)
    qe[::7] = qs[::7]                                            # zero-length
    qe[::11] = qs[::11] - rng.integers(1, 50, size=len(qs[::11]))  # reversed
    qe[::13] = qs[::13] + rng.integers(20_000, 3_000_000, size=len(qs[::13]))  # longer than a record holds
    o = np.argsort(qs, kind="stable")
    qs, qe = qs[o], qe[o]
    s, e, qs, qe = (np.clip(a, -(2**31), 2**31 - 1).astype(np.int32) for a in (s, e, qs, qe))
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    pick = np.arange(0, nq, 3)  # (a query on the pile costs the treap 70 000 steps: every third one is checked against it)
    want, _ = t.count_batch(qs[pick], qe[pick])
    ix = make_index(IntervalIndex, s, e)
    set_opt("ivl.partition", 1)
    if stage == "flat":
        set_opt("ivl.flat", 1)
        set_opt("ivl.bm_hard_ppm", 10**6)
    else:
        set_opt("ivl.sparse", 1)
    try:
        results = []
        for chunk, sorted_cells in ((0, 1), (4096, 1), (50_000, 1), (0, 0)):
            set_opt("ivl.bd_chunk", chunk)
            set_opt("ivl.sorted_cells", sorted_cells)
            got, got_total = ix.count(qs, qe)
            state = ix.flat_state() if stage == "flat" else ix.sparse_state()
            assert state[0] == 1, state
            bad = np.nonzero(got[pick] != want)[0]
            assert len(bad) == 0, (stage, chunk, sorted_cells, bad[:8], qs[pick][bad[:8]], qe[pick][bad[:8]], got[pick][bad[:8]], want[bad[:8]])
            assert got_total == int(got.sum(dtype=np.int64))
            assert ix.count(qs, qe, want_counts=False)[1] == got_total
            results.append(got)
        assert all(np.array_equal(results[0], r) for r in results[1:])
        assert int(results[0].max()) >= 70_000
        assert ix.order_state()[0] >= 0  # (the pass reported what its order check found)
    finally:
        reset_opts()


def test_partitioned_counts_beyond_16_bits(O, IntervalIndex):
    """Counts ride back to query order as 16 bits; a pile-up of >= 65535 overlapping targets takes the escape
    (recomputed from the index in the gather) -- regular, zero-length and reversed queries, both search variants."""
    rng = np.random.default_rng(21)
    pile = 70_000
    s = np.concatenate([np.full(pile, 1000), rng.integers(0, 3_000_000, size=40_000)])
    e = np.concatenate([np.full(pile, 2000), s[pile:] + rng.integers(1, 500, size=40_000)])
    s[:10] = 1500  # a few different starts inside the pile
    qs = rng.integers(0, 3_000_000, size=50_000)
    qe = qs + rng.integers(0, 800, size=50_000)
    qs[:6] = [1500, 1999, 1500, 1700, 999, 2000]
    qe[:6] = [1600, 2001, 1500, 1600, 1001, 2100]  # inside, edge, zero-length, reversed, touching both ends
    qs[100:200] = rng.integers(1000, 2000, size=100)
    qe[100:200] = qs[100:200] + rng.integers(0, 5, size=100)
    s, e, qs, qe = (a.astype(np.int32) for a in (s, e, qs, qe))
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    want, want_total = t.count_batch(qs, qe)
    ix = make_index(IntervalIndex, s, e)
    set_opt("ivl.partition", 1)
    try:
        fl, fl_total = ix.count(qs, qe)  # the flat walk on cell images: the pile is one hard cell per array, counts of 65535 and more escape
        fstate = ix.flat_state()
        set_opt("ivl.flat", 0)
        dn, dn_total = ix.count(qs, qe)  # dense unit images refuse the index (70 000 keys in one block: more than 15 bits of rank)
        dstate = ix.dense_state()
        set_opt("ivl.dense", 0)
        bm, bm_total = ix.count(qs, qe)  # neither kind of image: key slices, or the first-generation pass
        set_opt("ivl.bitmap", 0)
        got, got_total = ix.count(qs, qe)
    finally:
        reset_opts()
    assert int(want.max()) >= pile
    assert dstate[0] == -1 and dstate[1][0] >= pile, dstate
    assert fstate[0] == 1 and fstate[1] >= 2, fstate
    bad = np.nonzero(fl != want)[0]
    assert len(bad) == 0 and fl_total == want_total, ("flat walk", bad[:5], qs[bad[:5]], qe[bad[:5]], fl[bad[:5]], want[bad[:5]])
    bad = np.nonzero(dn != want)[0]
    assert len(bad) == 0 and dn_total == want_total, ("dense pass", bad[:5], qs[bad[:5]], qe[bad[:5]], dn[bad[:5]], want[bad[:5]])
    bad = np.nonzero(bm != want)[0]
    assert len(bad) == 0 and bm_total == want_total, ("bitmap pass", bad[:5], qs[bad[:5]], qe[bad[:5]], bm[bad[:5]], want[bad[:5]])
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, (bad[:5], qs[bad[:5]], qe[bad[:5]], got[bad[:5]], want[bad[:5]])
    assert got_total == want_total


def test_partitioned_path_dense_bucket_is_sampled(O, IntervalIndex):
    """250k of 300k targets sit inside one coordinate bucket: its slices exceed LDS and are staged sampled
    (every stride-th key) with a short finishing search -- counts must still be exact."""
    rng = np.random.default_rng(77)
    s = np.concatenate([rng.integers(0, 10_000_000, size=50_000), rng.integers(5_000_000, 5_004_000, size=250_000)])
    ln = rng.integers(0, 300, size=len(s))
    s, e = s.astype(np.int32), (s + ln).astype(np.int32)
    qs = np.concatenate([rng.integers(0, 10_000_000, size=10_000), rng.integers(4_999_000, 5_005_000, size=15_000)])
    qe = qs + rng.integers(0, 2000, size=len(qs))
    qe[::97] = qs[::97] - 5            # a few reversed queries
    qe[::89] += 3_000_000              # and some far longer than a bucket (leave the staged slice)
    qs, qe = qs.astype(np.int32), qe.astype(np.int32)
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    want, want_total = t.count_batch(qs, qe)
    ix = make_index(IntervalIndex, s, e)
    set_opt("ivl.partition", 1)
    try:
        set_opt("ivl.bm_hard_ppm", 10**6)  # keep the cell images although the dense stretch is all hard cells
        set_opt("ivl.dense", 0)            # (250k keys on 4000 coordinates: far more duplicates than a unit's overflow list holds)
        bm, bm_total = ix.count(qs, qe)
        state = ix.flat_state()
        set_opt("ivl.flat", 0)
        set_opt("ivl.bitmap", 0)
        got, got_total = ix.count(qs, qe)
        p_off, p_hits = ix.find(qs[:12000], qe[:12000])
    finally:
        reset_opts()
    assert state[0] == 1 and state[1] > 100, state
    bad = np.nonzero(bm != want)[0]
    assert len(bad) == 0 and bm_total == want_total, ("cell images, hard cells", bad[:5], qs[bad[:5]], qe[bad[:5]], bm[bad[:5]], want[bad[:5]])
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, (bad[:5], qs[bad[:5]], qe[bad[:5]], got[bad[:5]], want[bad[:5]])
    assert got_total == want_total
    w_off, w_hits = t.find_batch(qs[:12000], qe[:12000])
    assert np.array_equal(p_off, w_off) and np.array_equal(p_hits, w_hits)


def test_long_target_spanning_everything(O, IntervalIndex):
    """One chromosome-long interval inserted first: every later window starts at it."""
    rng = np.random.default_rng(5)
    s, e = _random_case(rng, 20000, 10**6, 0, 0, 100)
    s = np.concatenate([[-(10**6) - 5], s]).astype(np.int32)
    e = np.concatenate([[10**6 + 500], e]).astype(np.int32)
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    ix = make_index(IntervalIndex, s, e)
    qs, qe = _random_case(rng, 500, 10**6, 0.1, 0, 300)
    want_off, want_hits = t.find_batch(qs, qe)
    got_off, got_hits = ix.find(qs, qe)
    assert np.array_equal(got_off, want_off) and np.array_equal(got_hits, want_hits)
    assert ix.count(qs, qe)[0].tolist() == np.diff(want_off).tolist()


def test_incremental_append_reseals(O, IntervalIndex):
    rng = np.random.default_rng(9)
    s, e = _random_case(rng, 5000, 20000, 0.1, 0, 60)
    ix = IntervalIndex()
    t = O.OracleIntervalTree()
    for lo in range(0, 5000, 1250):
        ix.append(s[lo:lo + 1250], e[lo:lo + 1250])
        t.insert_many_arrays(s[lo:lo + 1250], e[lo:lo + 1250])
        qs, qe = _random_case(rng, 400, 20000, 0, 0, 100)
        assert ix.count(qs, qe)[0].tolist() == t.count_batch(qs, qe)[0].tolist()
        assert np.array_equal(ix.find(qs, qe)[1], t.find_batch(qs, qe)[1])


@pytest.mark.parametrize("stage", ["slices", "dense", "flat", "sparse", "clumped"])
@pytest.mark.parametrize("shape", ["uniform", "sorted", "one_bucket", "messy", "ragged_tail", "dups"])
def test_bitmap_pass_differential(O, IntervalIndex, shape, stage):
    """The large-batch count pass (count_bitmap.hpp, and its search stages count_slices.hpp and count_dense.hpp) against the oracle
    treap: shuffled, sorted and clumped batches, zero-length / reversed / off-grid / very long queries (escapes), tiles
    that are not full, targets whose coordinates carry duplicates (duplicate descriptors and hard cells; cells with
    thousands of keys for the slices), all tile shapes, unroll depths, unit sizes and run widths."""
    rng = np.random.default_rng(7 + ["uniform", "sorted", "one_bucket", "messy", "ragged_tail", "dups"].index(shape))
    n, span = 120_000, 40_000_000  # bucket width 2^15
    s = rng.integers(1000, span, size=n)
    if shape == "dups":
        s[: n // 2] = rng.choice(s[n // 2:], size=n // 2)          # half of the starts repeat another one
        s[:2000] = rng.integers(5_000_000, 5_000_064, size=2000)   # and two cells' worth of piled-up coordinates
    e = s + rng.integers(0, 1500, size=n)
    nq = {"ragged_tail": 16384 * 3 + 17}.get(shape, 70_000)
    qs = rng.integers(0, span + 2000, size=nq)
    qe = qs + rng.integers(1, 3000, size=nq)
    if shape == "sorted":
        o = np.argsort(qs, kind="stable")
        qs, qe = qs[o], qe[o]
    elif shape == "one_bucket":
        qs = rng.integers(20_000_000, 20_030_000, size=nq)
        qe = qs + rng.integers(1, 3000, size=nq)
    elif shape == "messy":
        k = nq // 10
        qe[:k] = qs[:k]                                       # zero-length
        qe[k:2 * k] = qs[k:2 * k] - rng.integers(1, 50, size=k)   # reversed
        qe[2 * k:3 * k] = qs[2 * k:3 * k] + rng.integers(32766, 5_000_000, size=k)  # longer than a record holds
        qs[3 * k:4 * k] = rng.integers(-(2**31), 1000, size=k)    # left of the grid
        qe[3 * k:4 * k] = qs[3 * k:4 * k] + rng.integers(1, 2000, size=k)
        qs[4 * k:5 * k] = rng.integers(span + 2000, 2**31 - 5000, size=k)  # right of it
        qe[4 * k:5 * k] = qs[4 * k:5 * k] + rng.integers(1, 2000, size=k)
        qs[5 * k:5 * k + 4] = [-(2**31), 2**31 - 1, 900, 999]
        qe[5 * k:5 * k + 4] = [2**31 - 1, 2**31 - 1, 1001, 1000]
        p = rng.permutation(nq)
        qs, qe = qs[p], qe[p]
    s, e, qs, qe = (np.clip(a, -(2**31), 2**31 - 1).astype(np.int32) for a in (s, e, qs, qe))
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    want, want_total = t.count_batch(qs, qe)
    ix = make_index(IntervalIndex, s, e)
    ix_blocks = [(0, 0)]
    set_opt("ivl.partition", 1)
    try:
        if stage in ("dense", "flat"):
            # units of 16 (8) buckets; tile shapes, work items of every size (several waves' batches / several items per
            # unit / one item per unit), record pipelines of every depth, both lookup styles and both rank bases of the
            # dense images, sorted batches through the exchange (long runs)
            set_opt("ivl.flat", 1 if stage == "flat" else 0)
            set_opt("ivl.dense", 1)
            if shape == "dups" and stage == "flat":
                set_opt("ivl.bm_hard_ppm", 10**6)  # (two cells' worth of piled-up coordinates and 60 000 repeated starts: keep the cells anyway)
            # w8: 8-bit counts between the search and the un-permute kernel (cell images): forced on (counts of 255 and more
            # come back as "ask again" and are recomputed), off, or left to the density + feedback
            # tf: duplicated coordinates from which a cell of the dense images gets a rank table (0 = two where the LDS has the room, else six)
            for k, (variant, chunk, blocks, w8, tf) in enumerate(((0, 0, 0, 0, 0), (1, 4096, 1, 0, 6), (2, 1 << 20, 1, 0, 6), (-1, 20000, 0, 0, 2),
                                                                  (0, 1024, 0, 0, 2), (1, 65536, 0, 0, 3), (2, 4096, 0, 0, 3), (2, 0, 0, 1, 0),
                                                                  (1, 8192, 0, 1, 0), (-1, 0, 0, -1, 0), (2, 0, 1, 0, 2))):
                set_opt("ivl.sorted_path", k % 2)
                set_opt("ivl.bm_variant", variant)
                set_opt("ivl.bd_chunk", chunk)
                set_opt("ivl.bd_w8", w8)
                if stage == "dense" and (blocks, tf) != ix_blocks[0]:
                    set_opt("ivl.bd_blocks", blocks)
                    set_opt("ivl.bd_table_from", tf)
                    ix.seal()  # (the rank base and the tables of the images are decided when the index is prepared)
                    ix_blocks[0] = (blocks, tf)
                got, got_total = ix.count(qs, qe)
                state = ix.dense_state() if stage == "dense" else ix.flat_state()
                assert state[0] == 1 and ix.slice_state()[0] == 0, (state, ix.slice_state())
                assert (ix.flat_state()[0] == 0) == (stage == "dense")
                bad = np.nonzero(got != want)[0]
                assert len(bad) == 0, (shape, stage, variant, chunk, blocks, w8, state, bad[:8], qs[bad[:8]], qe[bad[:8]], got[bad[:8]], want[bad[:8]])
                assert got_total == want_total
            if shape == "dups":
                assert (state[1][1] > 100) if stage == "dense" else (state[1] > 0)  # overflow entries / hard cells
            return
        if stage == "clumped":
            # offset cells in the CLUMPED layout (round 6: cells of 64 coordinates, every hard cell with a rank table -- bytes, or 16
            # bits for the two piled-up cells of "dups" -- one 1024-thread workgroup per CU): tile shapes, work items of every
            # size, both count widths, totals only
            set_opt("ivl.clumped", 1)
            set_opt("ivl.flat", 0)
            for k, (variant, chunk, w8) in enumerate(((0, 0, 0), (1, 4096, 0), (2, 1 << 20, 1), (-1, 20000, -1), (2, 0, -1), (1, 65536, 1))):
                set_opt("ivl.sorted_path", k % 2)
                set_opt("ivl.bm_variant", variant)
                set_opt("ivl.bd_chunk", chunk)
                set_opt("ivl.bd_w8", w8)
                got, got_total = ix.count(qs, qe)
                state = ix.sparse_state()
                assert state[0] == 2 and state[2] == 6 and ix.slice_state()[0] == 0 and ix.dense_state()[0] == 0, (state, ix.slice_state(), ix.dense_state())
                bad = np.nonzero(got != want)[0]
                assert len(bad) == 0, (shape, stage, variant, chunk, w8, state, bad[:8], qs[bad[:8]], qe[bad[:8]], got[bad[:8]], want[bad[:8]])
                assert got_total == want_total
                assert ix.count(qs, qe, want_counts=False)[1] == want_total
            if shape == "dups":
                assert state[1] > 0  # hard cells: every one of them tabled
            return
        if stage == "sparse":
            # offset-cell images (one target per 333 coordinates: cells of 256 by the density; narrower ones forced): tile shapes,
            # work items of every size, both count widths; queries longer than the record's length field (4095 at cells of
            # 256) are escapes; "dups": cells with more than five keys -- lists in LDS, and two cells with a thousand keys each
            # that are finished by a search in the sorted array
            set_opt("ivl.sparse", 1)
            if shape == "dups":
                set_opt("ivl.bm_hard_ppm", 10**6)
            width = [0]
            for k, (variant, chunk, w8, cell) in enumerate(((0, 0, 0, 0), (1, 4096, 0, 7), (2, 1 << 20, 1, 6), (-1, 20000, -1, 8), (0, 1024, 1, 0),
                                                            (2, 0, -1, 7), (1, 65536, 0, 6))):
                set_opt("ivl.sorted_path", k % 2)
                set_opt("ivl.bm_variant", variant)
                set_opt("ivl.bd_chunk", chunk)
                set_opt("ivl.bd_w8", w8)
                if cell != width[0]:
                    set_opt("ivl.bo_cell_log2", cell)
                    ix.seal()  # (the cell width is decided when the index is prepared)
                    width[0] = cell
                got, got_total = ix.count(qs, qe)
                state = ix.sparse_state()
                assert state[0] == 1 and state[2] == (cell or 8) and ix.slice_state()[0] == 0 and ix.flat_state()[0] == 0, (state, ix.slice_state())
                bad = np.nonzero(got != want)[0]
                assert len(bad) == 0, (shape, stage, variant, chunk, w8, cell, state, bad[:8], qs[bad[:8]], qe[bad[:8]], got[bad[:8]], want[bad[:8]])
                assert got_total == want_total
            if shape == "dups":
                assert state[1] > 0  # hard cells
            return
        set_opt("ivl.dense", 0)
        set_opt("ivl.flat", 0)
        if stage == "slices":
            set_opt("ivl.slice", 1)
            # lanes: 16 / 64 per (tile, unit) run, 1 = the flat walk over the item's runs, 0 = by expected run length
            for k, (variant, f, lanes) in enumerate(((0, -1, 0), (1, 0, 16), (2, 2, 64), (0, 6, 16), (2, 6, 64), (1, 3, 0), (0, 1, 64), (-1, -1, 0),
                                                     (0, 6, 1), (2, 0, 1), (1, 4, 1))):
                set_opt("ivl.sorted_path", k % 2)
                set_opt("ivl.bm_variant", variant)
                set_opt("ivl.sl_f", f)
                set_opt("ivl.sl_lanes", lanes)
                set_opt("ivl.sl_flat", (k >> 1) & 1)  # the flat 16-byte walk of count_dense.hpp, or the lanes-per-run kernels
                got, got_total = ix.count(qs, qe)
                state = ix.slice_state()
                assert state[0] == 1, state
                bad = np.nonzero(got != want)[0]
                assert len(bad) == 0, (shape, variant, f, lanes, state, bad[:8], qs[bad[:8]], qe[bad[:8]], got[bad[:8]], want[bad[:8]])
                assert got_total == want_total
            assert state[1][0] > 0 and all(a <= b for a, b in zip(state[1], state[1][1:]))  # keys per unit grow with the unit
            return
        raise AssertionError("unknown stage " + stage)
    finally:
        reset_opts()


@pytest.mark.parametrize("shape", ["uniform", "sorted", "messy", "ragged_tail", "dups", "long_target", "pile", "pile70k"])
def test_find_through_the_exchange_differential(O, IntervalIndex, shape):
    """find() on large unsorted batches (count_slices.hpp: count half; find_exchange.hpp: CSR offsets, the fill on LDS windows of
    half-bucket pieces, hits back to query order) against the oracle treap's find: same
    offsets, same hits in the same order.  Escapes (zero-length / reversed / off-grid / over-long queries), duplicated
    coordinates (queries with more hits than a wave's LDS image holds), a tile that is not full, one target spanning
    everything (walks that leave the staged window), a pile of targets larger than a piece's window, all tile shapes, unit
    sizes and run widths, with and without the copy; and the bucketed find of the first generation on the same input.
    pile70k: 70 000 long targets under every query, 65535 hits and more per query -- the count does not fit the word the count half packs for the fill (0xFFFF: read again
    from the 32-bit counts, a hand-issued load since round 6)."""
    shapes = ["uniform", "sorted", "messy", "ragged_tail", "dups", "long_target", "pile", "pile70k"]
    rng = np.random.default_rng(70 + shapes.index(shape))
    n, span = 100_000, 30_000_000
    s = rng.integers(1000, span, size=n)
    if shape == "dups":
        s[: n // 2] = rng.choice(s[n // 2:], size=n // 2)
        s[:1500] = rng.integers(5_000_000, 5_000_040, size=1500)
    if shape == "pile":
        s[:25_000] = rng.integers(5_001_000, 5_003_000, size=25_000)  # one half bucket holds more pairs than an LDS window
    e = s + rng.integers(0, 1200, size=n)
    if shape == "pile70k":  # 70 000 long targets that every query meets
        s[:70_000] = rng.integers(1000, 2_000_000, size=70_000)
        e[:70_000] = s[:70_000] + rng.integers(27_000_000, 28_000_000, size=70_000)
    if shape == "long_target":
        s[0], e[0] = 2000, span - 5  # every query meets it, and the walk down from hi passes thousands of candidates
        s[1], e[1] = 15_000_000, 15_400_000
    nq = {"ragged_tail": 16384 * 2 + 311, "pile70k": 260}.get(shape, 50_000)
    qs = rng.integers(0, span + 2000, size=nq)
    qe = qs + rng.integers(1, 2500, size=nq)
    if shape == "pile70k":
        qs = rng.integers(3_000_000, 26_000_000, size=nq)
        qe = qs + rng.integers(1, 2500, size=nq)
    if shape == "pile":
        qs[:300] = rng.integers(4_999_000, 5_004_000, size=300)
        qe[:300] = qs[:300] + rng.integers(1, 400, size=300)
    if shape == "sorted":
        o = np.argsort(qs, kind="stable")
        qs, qe = qs[o], qe[o]
    elif shape == "messy":
        k = nq // 10
        qe[:k] = qs[:k]
        qe[k:2 * k] = qs[k:2 * k] - rng.integers(1, 50, size=k)
        qe[2 * k:3 * k] = qs[2 * k:3 * k] + rng.integers(32766, 300_000, size=k)
        qs[3 * k:4 * k] = rng.integers(-(2**31), 1000, size=k)
        qe[3 * k:4 * k] = qs[3 * k:4 * k] + rng.integers(1, 2000, size=k)
        qs[4 * k:5 * k] = rng.integers(span + 2000, 2**31 - 5000, size=k)
        qe[4 * k:5 * k] = qs[4 * k:5 * k] + rng.integers(1, 2000, size=k)
        p = rng.permutation(nq)
        qs, qe = qs[p], qe[p]
    s, e, qs, qe = (np.clip(a, -(2**31), 2**31 - 1).astype(np.int32) for a in (s, e, qs, qe))
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    want_off, want_hits = t.find_batch(qs, qe)
    ix = make_index(IntervalIndex, s, e)
    set_opt("ivl.partition", 1)
    set_opt("ivl.bitmap_min", 1)
    try:
        for k, (variant, f, lanes, sorted_path) in enumerate(((0, -1, 0, 0), (1, 0, 16, 0), (2, 2, 64, 0), (0, 6, 16, 0), (2, 6, 64, 0), (-1, -1, 0, 1))):
            # the fill on LDS windows straight into the CSR list (1), or into scratch and then the copy (0)
            for fx_fill in (1, 0):
                set_opt("ivl.fx_direct", fx_fill)
                set_opt("ivl.sorted_path", sorted_path)
                set_opt("ivl.bm_variant", variant)
                set_opt("ivl.sl_f", f)
                set_opt("ivl.sl_lanes", lanes)
                off, hits = ix.find(qs, qe)
                assert ix.slice_state()[0] == 1
                assert np.array_equal(off, want_off), (shape, variant, f, lanes, fx_fill, np.nonzero(np.diff(off) != np.diff(want_off))[0][:8])
                bad = np.nonzero(hits != want_hits)[0]
                assert len(bad) == 0, (shape, variant, f, lanes, fx_fill, bad[:8], hits[bad[:8]], want_hits[bad[:8]])
        set_opt("ivl.fx_direct", 1)
        set_opt("ivl.find_sliced", 0)
        off, hits = ix.find(qs, qe)
        assert np.array_equal(off, want_off) and np.array_equal(hits, want_hits), "bucketed find"
        # a buffer that is too small is reported with the total (the wrapper then retries with the exact size)
        set_opt("ivl.find_sliced", 1)
        off, hits = ix.find(qs, qe, cap_hint=max(1, int(want_off[-1]) // 2 - 1))
        assert np.array_equal(off, want_off) and np.array_equal(hits, want_hits), "after BXMI_ERANGE"
        # ... and BXMI_ERANGE itself comes with the offsets and the total valid and the hit buffer untouched (include/bxmi.h)
        for direct in (1, 0):
            set_opt("ivl.fx_direct", direct)
            _erange_contract(ix, qs, qe, want_off)
        set_opt("ivl.fx_direct", 1)
    finally:
        reset_opts()


def test_bitmap_pass_is_refused_where_it_does_not_fit(O, IntervalIndex):
    """Spans beyond 2^28, reversed targets and heavily duplicated coordinates keep the bucketed search pass when the slice
    stage is off; with it on (the default) the wide span and the duplicates are served by slices, and a bucket that
    holds more keys than the LDS refuses the slices too."""
    rng = np.random.default_rng(9)
    cases = {}
    s = rng.integers(0, 2**30, size=50_000)
    cases["wide"] = (s, s + rng.integers(0, 500, size=50_000))
    s = rng.integers(0, 10_000_000, size=50_000)
    e = s + rng.integers(0, 500, size=50_000)
    e[7] = s[7] - 3
    cases["reversed"] = (s, e)
    s = rng.integers(0, 12_500, size=50_000) * 8  # four occupied coordinates per cell, each several times over: all cells hard
    cases["duplicates"] = (s, s + 8 * rng.integers(0, 10, size=50_000))
    qs = rng.integers(0, 10_000_000, size=20_000).astype(np.int32)
    qe = (qs + rng.integers(1, 800, size=20_000)).astype(np.int32)
    s = rng.integers(0, 2**29, size=150_000)
    s[:90_000] = rng.integers(1_000_000, 1_001_000, size=90_000)  # one bucket with 90k keys: more than a workgroup stages
    cases["pile"] = (s, s + rng.integers(0, 500, size=150_000))
    set_opt("ivl.partition", 1)
    set_opt("ivl.dense", 0)  # (the dense images and the flat walk have their own limits: test_dense_pass_limits)
    set_opt("ivl.flat", 0)
    try:
        for name, (s, e) in cases.items():
            s, e = s.astype(np.int32), e.astype(np.int32)
            t = O.OracleIntervalTree()
            t.insert_many_arrays(s, e)
            want, want_total = t.count_batch(qs, qe)
            for slices in (0, -1):
                set_opt("ivl.slice", slices)
                ix = make_index(IntervalIndex, s, e)
                assert ix.slice_state()[0] == 0
                got, got_total = ix.count(qs, qe)
                assert np.array_equal(got, want) and got_total == want_total, (name, slices)
                sl = ix.slice_state()[0]
                if name == "reversed" or slices == 0:
                    assert sl == 0, (name, sl)
                elif name == "pile":
                    assert sl == -1, (name, sl)
                else:
                    assert sl == 1, (name, sl)
                ix.close()
    finally:
        reset_opts()


def test_dense_pass_limits(O, IntervalIndex):
    """What the dense unit images (count_dense.hpp) hold and what they refuse: a span of 2^30 (buckets of 2^19, one per
    unit) is served when asked for; more than 32767 keys in one 2^17-coordinate block, more duplicated coordinates than a
    unit's overflow list holds, and reversed targets are refused -- the batch then takes the other stages, same counts."""
    rng = np.random.default_rng(19)
    cases = {}
    s = rng.integers(0, 2**30, size=50_000)
    cases["wide"] = (s, s + rng.integers(0, 500, size=50_000), 1)
    s = rng.integers(0, 2**29, size=150_000)
    s[:90_000] = rng.integers(1_000_000, 1_001_000, size=90_000)  # 90k keys on 1000 coordinates: one block overflows its 15-bit ranks
    cases["pile"] = (s, s + rng.integers(0, 500, size=150_000), -1)
    s = rng.integers(0, 125_000, size=500_000) * 8  # 125 000 coordinates, each several times over: 4096 of them per unit of 2^15
    cases["duplicates"] = (s, s + 8 * rng.integers(0, 10, size=500_000), -1)
    s = rng.integers(0, 400_000, size=60_000)
    s[:3000] = rng.choice(s[3000:], size=3000)  # a few thousand duplicated coordinates in a few units: fits
    cases["some_duplicates"] = (s, s + rng.integers(0, 300, size=60_000), 1)
    s = rng.integers(0, 10_000_000, size=50_000)
    e = s + rng.integers(0, 500, size=50_000)
    e[7] = s[7] - 3
    cases["reversed"] = (s, e, 0)
    set_opt("ivl.partition", 1)
    set_opt("ivl.dense", 1)
    set_opt("ivl.flat", 0)
    try:
        for name, (s, e, expect) in cases.items():
            s, e = s.astype(np.int32), e.astype(np.int32)
            hi = int(s.max()) + 1000
            qs = rng.integers(-500, hi, size=40_000).astype(np.int32)
            qe = (qs + rng.integers(1, 900, size=40_000)).astype(np.int32)
            t = O.OracleIntervalTree()
            t.insert_many_arrays(s, e)
            want, want_total = t.count_batch(qs, qe)
            ix = make_index(IntervalIndex, s, e)
            got, got_total = ix.count(qs, qe)
            state = ix.dense_state()
            bad = np.nonzero(got != want)[0]
            assert len(bad) == 0 and got_total == want_total, (name, state, bad[:8], qs[bad[:8]], qe[bad[:8]], got[bad[:8]], want[bad[:8]])
            assert state[0] == expect, (name, state)
            if name == "pile":
                assert state[1][0] > 32767
            if name == "duplicates":
                assert state[1][1] > 5632
            # the flat walk on cell images: spans up to 2^28 (units of at least two buckets: the runs are padded; the pile's
            # index spans 2^29), coordinates duplicated all over are too many of them (a key on every 7th coordinate and a duplicate on every
            # 60th: a tenth of the cells hold two duplicated coordinates -- the dense images' overflow lists take those)
            set_opt("ivl.flat", 1)
            ix.seal()
            got, got_total = ix.count(qs, qe)
            fstate = ix.flat_state()
            set_opt("ivl.flat", 0)
            bad = np.nonzero(got != want)[0]
            assert len(bad) == 0 and got_total == want_total, ("flat", name, fstate, bad[:8], qs[bad[:8]], qe[bad[:8]], got[bad[:8]], want[bad[:8]])
            assert fstate[0] == {"wide": -1, "pile": -1, "duplicates": -1, "some_duplicates": -1, "reversed": 0}[name], (name, fstate)
            ix.close()
    finally:
        reset_opts()


# --------------------------------------------------------- scale / golden hash --
def test_scale_1M_hash(golden_scale, IntervalIndex):
    pt = golden_scale["1M x 200k"]
    (ts, te), _ = synth.cfg2(pt["n_targets"], 1)
    qs, qe = synth.uniform_intervals(pt["n_queries_total"], 202)
    qs, qe = qs[:: pt["stride"]].copy(), qe[:: pt["stride"]].copy()
    ix = make_index(IntervalIndex, ts, te)
    counts, total = ix.count(qs, qe)
    assert total == pt["total"]
    assert hashlib.sha256(counts.tobytes()).hexdigest() == pt["counts_sha256"]
    set_opt("ivl.partition", 1)
    try:
        set_opt("ivl.flat", 1)  # the flat walk on cell images although the index is sparse
        for variant in (0, 1, 2):
            set_opt("ivl.bm_variant", variant)
            counts, total = ix.count(qs, qe)
            assert ix.flat_state()[0] == 1
            assert total == pt["total"] and hashlib.sha256(counts.tobytes()).hexdigest() == pt["counts_sha256"], ("flat walk", variant)
        set_opt("ivl.flat", 0)
        set_opt("ivl.dense", 1)  # dense unit images
        for variant in (0, 1, 2):
            set_opt("ivl.bm_variant", variant)
            counts, total = ix.count(qs, qe)
            assert ix.dense_state()[0] == 1
            assert total == pt["total"] and hashlib.sha256(counts.tobytes()).hexdigest() == pt["counts_sha256"], ("dense stage", variant)
        set_opt("ivl.dense", 0)
        for variant in (0, 1, 2, 3):  # the slice stage (1M targets on 250M coordinates are sparse): tile shapes, unit sizes, run widths
            set_opt("ivl.bm_variant", variant % 3)
            set_opt("ivl.sl_f", (-1, 0, 3, 6)[variant])
            set_opt("ivl.sl_lanes", (0, 64, 16, 1)[variant])
            counts, total = ix.count(qs, qe)
            assert ix.slice_state()[0] == 1
            assert total == pt["total"] and hashlib.sha256(counts.tobytes()).hexdigest() == pt["counts_sha256"], ("slice stage", variant)
        set_opt("ivl.bitmap", 0)
        counts, total = ix.count(qs, qe)
    finally:
        reset_opts()
    assert total == pt["total"] and hashlib.sha256(counts.tobytes()).hexdigest() == pt["counts_sha256"], "partitioned"
    offs, hits = ix.find(qs, qe)
    assert offs[-1] == pt["total"] and np.array_equal(np.diff(offs), counts)
    # every reported hit really overlaps, and hits of one query come in tree order
    rep = np.repeat(np.arange(len(qs)), counts)
    assert (te[hits] > qs[rep]).all() and (ts[hits] < qe[rep]).all()


def test_scale_cfg2_full_size_properties(golden_scale, IntervalIndex):
    """BASELINE configs[1]: 100M queries x 10M targets, count-only.  The reference hash pins the
    1M-query subsample; the rest is checked through size-independent identities."""
    key = "10M x 1M (cfg2 subsample)"
    (ts, te), (qs, qe) = synth.cfg2()
    ix = make_index(IntervalIndex, ts, te)
    counts, total = ix.count(qs, qe)
    assert total == int(counts.sum(dtype=np.int64))
    assert key in golden_scale, "tests/golden/scale.json lost its %r point: the reference check of configs[1] must not vanish silently" % key
    pt = golden_scale[key]
    sub = counts[:: pt["stride"]]
    assert int(sub.sum(dtype=np.int64)) == pt["total"]
    assert hashlib.sha256(np.ascontiguousarray(sub).tobytes()).hexdigest() == pt["counts_sha256"]
    assert ix.flat_state()[0] == 1 and ix.dense_state()[0] == 0  # the full batch above went through the flat walk on cell images
    # the direct tree kernel and the large-batch passes agree (first 8M queries through the direct kernel)
    set_opt("ivl.partition", 0)
    try:
        direct, _ = ix.count(qs[:8_000_000], qe[:8_000_000])
        set_opt("ivl.partition", -1)
        set_opt("ivl.flat", 0)    # the same pass on dense unit images
        img, img_total = ix.count(qs, qe)
        assert ix.dense_state()[0] == 1
        assert np.array_equal(img, counts) and img_total == total
        del img
        set_opt("ivl.bitmap", 0)  # the bucketed search pass on the whole batch
        old, old_total = ix.count(qs, qe)
    finally:
        reset_opts()
    assert np.array_equal(direct, counts[:8_000_000])
    assert np.array_equal(old, counts) and old_total == total
    del old
    # additivity: counting two halves separately gives the same per-query numbers
    h = len(qs) // 2
    c2, t2 = ix.count(qs[h:], qe[h:])
    assert np.array_equal(c2, counts[h:])
    # a query covering the whole genome sees every target; an empty one sees none
    c3, _ = ix.count(np.array([-5, 7], np.int32), np.array([2**31 - 1, 7], np.int32))
    assert c3[0] == len(ts)
    # monotonicity: widening a query never loses hits
    wide, _ = ix.count(qs[:1_000_000] - 100, qe[:1_000_000] + 100)
    assert (wide >= counts[:1_000_000]).all()


@pytest.mark.parametrize("stage", ["dense", "flat", "sparse"])
def test_padded_runs_on_many_full_tiles(O, IntervalIndex, stage):
    """The padded layout of the flat walk (every unit's run of a tile on whole 16-byte slots, ring of record loads) at the
    shape the headline uses: 32768-query tiles, all of them full but the last, batches of 64 tiles per wave, several rounds
    of the ring per batch -- against the oracle treap, with 16-bit and 8-bit counts."""
    rng = np.random.default_rng(77)
    n, span = 300_000, 60_000_000
    s = rng.integers(1000, span, size=n)
    e = s + rng.integers(1, 1500, size=n)
    nq = 32768 * 150 + 4321
    qs = rng.integers(0, span + 2000, size=nq)
    qe = qs + rng.integers(1, 3000, size=nq)
    s, e, qs, qe = (a.astype(np.int32) for a in (s, e, qs, qe))
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    want, want_total = t.count_batch(qs, qe)
    ix = make_index(IntervalIndex, s, e)
    set_opt("ivl.partition", 1)
    set_opt("ivl.bm_variant", 2)
    if stage == "sparse":  # offset cells of 128 coordinates, units of 2^19: runs of ~280 records, batches of 32 tiles per wave
        set_opt("ivl.sparse", 1)
    else:
        set_opt("ivl.flat", 1 if stage == "flat" else 0)
        set_opt("ivl.dense", 1)
    try:
        for w8 in (0, 1, -1):
            set_opt("ivl.bd_w8", w8)
            got, got_total = ix.count(qs, qe)
            state = ix.dense_state() if stage == "dense" else (ix.flat_state() if stage == "flat" else ix.sparse_state())
            assert state[0] == 1, state
            bad = np.nonzero(got != want)[0]
            assert len(bad) == 0 and got_total == want_total, (stage, w8, bad[:8], qs[bad[:8]], qe[bad[:8]], got[bad[:8]], want[bad[:8]])
    finally:
        reset_opts()


def test_sorted_batches_after_shuffled_ones_with_the_order_check_kept(O, IntervalIndex):
    """With the exact order check on every pass (ivl.order_skip = 0) shuffled batches stand the sorted-batch kernel down and a
    sorted batch right after them is answered by it: same counts every time."""
    rng = np.random.default_rng(93)
    n, span, nq = 200_000, 50_000_000, 4096 * 4200 + 77
    s = rng.integers(1000, span, size=n)
    e = s + rng.integers(1, 1500, size=n)
    qs = rng.integers(0, span, size=nq)
    qe = qs + rng.integers(1, 2500, size=nq)
    s, e, qs, qe = (a.astype(np.int32) for a in (s, e, qs, qe))
    ix = make_index(IntervalIndex, s, e)
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    pick = rng.integers(0, nq, size=300_000)
    want, _ = t.count_batch(qs[pick], qe[pick])
    set_opt("ivl.order_skip", 0)  # (or the check itself would be dropped after the shuffled batches)
    first, first_total = ix.count(qs, qe)
    assert np.array_equal(first[pick], want)
    for _ in range(2):
        got, got_total = ix.count(qs, qe)
        assert np.array_equal(got, first) and got_total == first_total
    o = np.argsort(qs, kind="stable")
    for _ in range(3):
        got, got_total = ix.count(qs[o], qe[o])
        assert np.array_equal(got, first[o]) and got_total == first_total
    got, got_total = ix.count(qs, qe)
    reset_opts()
    assert np.array_equal(got, first) and got_total == first_total


@pytest.mark.parametrize("stage", ["flat", "dense", "slices"])
def test_order_check_is_dropped_and_comes_back(O, IntervalIndex, stage):
    """After two shuffled batches in a row the order check is no longer launched: a probe of 8192 starts rides on the
    parameter kernel and reports through host memory.  A batch without a descent in the probe -- sorted, or sorted but for
    its last two queries -- brings the exact check (and the kernel for sorted batches) back; met without the check it
    goes through the exchange.  Same counts at every step, for every search stage."""
    rng = np.random.default_rng(95)
    n, span, nq = 150_000, 40_000_000, 32768 * 66 + 1234
    s = rng.integers(1000, span, size=n)
    e = s + rng.integers(1, 1500, size=n)
    qs = rng.integers(0, span, size=nq)
    qe = qs + rng.integers(1, 2500, size=nq)
    s, e, qs, qe = (a.astype(np.int32) for a in (s, e, qs, qe))
    ix = make_index(IntervalIndex, s, e)
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    want, want_total = t.count_batch(qs, qe)
    o = np.argsort(qs, kind="stable")
    almost = o.copy()
    almost[-1], almost[-3] = almost[-3], almost[-1]  # a descent at the very end of the batch: not in the probe
    set_opt("ivl.partition", 1)
    set_opt("ivl.flat", 1 if stage == "flat" else 0)
    set_opt("ivl.dense", 1 if stage in ("flat", "dense") else 0)
    set_opt("ivl.slice", 1 if stage == "slices" else 0)

    def shuffled(times):
        for k in range(times):
            got, got_total = ix.count(qs, qe)
            assert np.array_equal(got, want) and got_total == want_total, ("shuffled", k)

    def ordered(order, what):
        got, got_total = ix.count(qs[order], qe[order])
        assert np.array_equal(got, want[order]) and got_total == want_total, what

    try:
        assert ix.order_state()[0] == 0
        shuffled(3)
        skipping, seen = ix.order_state()
        assert skipping == 1 and seen >= 2, (skipping, seen)
        ordered(almost, "almost sorted, no check")       # through the exchange; the probe sees no descent
        shuffled(1)                                      # ... so this one is checked again
        assert ix.order_state()[0] == 0
        shuffled(2)
        assert ix.order_state()[0] == 1
        ordered(o, "sorted, no check")                   # through the exchange
        ordered(o, "sorted, checked")                    # its report has arrived: the check is back
        assert ix.order_state()[0] == 0
        ordered(almost, "almost sorted, checked")        # the exact check finds the descent
        shuffled(1)
    finally:
        reset_opts()


def test_total_only_batches(O, IntervalIndex):
    """bxmi_ivl_count_dev / _multi_dev with counts = NULL (scripts/bed_count_overlapping.py consumes len(find()) only; configs[3]'s
    "all-reduce on counts"): the same pass, nothing stored per query.  On 8-bit counts the un-permute kernel sums a tile's count
    bytes without reading a slot when every 0xFF byte is the tile sort's padding, and takes the ordinary path for a tile where a
    real query came back as "ask again" -- escapes (reversed, over-long, off-grid queries) and counts of 255 and more.  Totals
    against the oracle on every search stage that serves the index, shuffled and sorted."""
    from bxmi import _ffi

    rng = np.random.default_rng(77)
    span = 30_000_000
    s = np.concatenate([rng.integers(1000, span, size=400_000), rng.integers(5_000_000, 5_400_000, size=150_000)])
    e = s + rng.integers(1, 300, size=len(s))
    s, e = s.astype(np.int32), e.astype(np.int32)
    ix = make_index(IntervalIndex, s, e)
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    nq = 32768 * 66 + 77
    batches = {}
    qs = rng.integers(6_000_000, span, size=nq)
    batches["plain"] = (qs, qs + rng.integers(1, 800, size=nq))            # no tile holds an escape: sums of bytes only
    qe = qs + rng.integers(1, 800, size=nq)
    k = np.arange(0, nq, 40_000)                                          # an escape in every other tile or so
    qe = qe.copy()
    qe[k[0::3]] = qs[k[0::3]] - 4
    qe[k[1::3]] = qs[k[1::3]] + 50_000
    qs2 = qs.copy()
    qs2[k[2::3]] = -5000
    qe[k[2::3]] = -4000
    batches["escapes"] = (qs2, qe)
    qs = rng.integers(5_000_000, 5_400_000, size=nq)                       # the crowd: counts of 255 and more in every tile
    batches["crowd"] = (qs, qs + rng.integers(900, 1200, size=nq))
    total = _ffi.DeviceArray(8)
    # (the oracle's answer does not depend on the stage: one count_batch per batch and order, not per stage)
    cases = []
    for name, (a, b) in batches.items():
        for order in ("shuffled", "sorted"):
            if order == "sorted":
                o = np.argsort(a, kind="stable")
                a, b = a[o], b[o]
            a32, b32 = a.astype(np.int32), b.astype(np.int32)
            cases.append((name, order, a32, b32, t.count_batch(a32, b32)[1]))
    set_opt("ivl.partition", 1)
    try:
        # (on cell images -- bitmap cells and offset cells -- the persistent walk keeps the totals itself, ivl.tot_walk: no slots, no
        # count stores, no un-permute kernel, the queries behind escape records answered by bm_escape_totals_kernel; with it off,
        # and on the other stages, the counts pass without its stores)
        for stage in (("ivl.flat", 1, 1), ("ivl.flat", 1, 0), ("ivl.sparse", 1, 1), ("ivl.sparse", 1, 0), ("ivl.dense", 1, 1), ("ivl.slice", 1, 1)):
            reset_opts()
            set_opt("ivl.partition", 1)
            set_opt("ivl.bm_hard_ppm", 10**6)
            set_opt(stage[0], stage[1])
            set_opt("ivl.tot_walk", stage[2])
            if stage[0] != "ivl.flat":
                set_opt("ivl.flat", 0)
            if stage[0] in ("ivl.slice", "ivl.sparse"):
                set_opt("ivl.dense", 0)
            if stage[0] == "ivl.sparse":
                ix.seal()  # (offset-cell images are built per sealed index: asked for now)
            for name, order, a32, b32, want in cases:
                if True:
                    dq, de = _ffi.DeviceArray.from_numpy(a32), _ffi.DeviceArray.from_numpy(b32)
                    for rep in range(2):  # (the second pass of the crowd may already run on 16-bit counts)
                        total.zero()
                        ix.count_dev(dq.ptr, de.ptr, nq, None, total.ptr, None)
                        _ffi.call("bxmi_synchronize", None)
                        got = int(total.to_numpy(np.int64, 1)[0])
                        assert got == want, (stage, name, order, rep, got, want)
            if stage[0] == "ivl.sparse":
                assert ix.sparse_state()[0] == 1, ix.sparse_state()  # (the offset-cell walk did serve the index)
            if stage[0] == "ivl.flat":
                assert ix.flat_state()[0] == 1, ix.flat_state()
    finally:
        reset_opts()


def test_host_count_in_chunks(O, IntervalIndex):
    """bxmi_ivl_count on host arrays, the chunked pipeline (upload of chunk k+1 / pass on k / download of k-1 at once,
    ivl_count_host_chunks): the same counts and total as the oracle whatever the chunk size and however many threads touch
    the output's pages -- ragged last chunk, chunks below and above the batch-pass threshold, shuffled and sorted, counts
    wanted or the total only, into a fresh and into a written output array (the touchers must leave neither zeros nor stale
    values behind).  intersection.pyx:400-406 per query."""
    import ctypes as C

    from bxmi import _ffi

    rng = np.random.default_rng(4242)
    span = 40_000_000
    s = rng.integers(0, span, size=300_000).astype(np.int32)
    e = (s + rng.integers(1, 2000, size=len(s))).astype(np.int32)
    ix = make_index(IntervalIndex, s, e)
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    nq = 5 * 1_000_000 + 12_345
    qs = rng.integers(-1000, span + 1000, size=nq).astype(np.int32)
    qe = (qs + rng.integers(0, 3000, size=nq)).astype(np.int32)
    try:
        for order in ("shuffled", "sorted"):
            if order == "sorted":
                o = np.argsort(qs, kind="stable")
                qs, qe = np.ascontiguousarray(qs[o]), np.ascontiguousarray(qe[o])
            want_counts, want_total = t.count_batch(qs, qe)
            for chunk, touchers in ((4096, 2), (1 << 20, 0), (1 << 20, 3), (2_500_000, 1), (0, 2)):
                reset_opts()
                set_opt("ivl.host_chunk", chunk)
                set_opt("ivl.host_touchers", touchers)
                for fresh in (True, False):
                    counts = np.empty(nq, dtype=np.int32) if fresh else np.full(nq, -3, dtype=np.int32)
                    total = C.c_int64(-1)
                    _ffi.call("bxmi_ivl_count", ix._h, _ffi.ptr(qs), _ffi.ptr(qe), nq, _ffi.ptr(counts), C.byref(total))
                    assert total.value == want_total, (order, chunk, touchers, fresh)
                    assert np.array_equal(counts, want_counts), (order, chunk, touchers, fresh, np.nonzero(counts != want_counts)[0][:8])
                total = C.c_int64(-1)
                _ffi.call("bxmi_ivl_count", ix._h, _ffi.ptr(qs), _ffi.ptr(qe), nq, None, C.byref(total))
                assert total.value == want_total, (order, chunk, touchers, "total only")
    finally:
        reset_opts()


def test_count_width_feedback(O, IntervalIndex):
    """8-bit counts between the search and the un-permute kernel of the flat walk.  The index is sparse as a whole (the host
    starts with 8 bits) but 150 000 of its targets crowd into 400 000 coordinates; queries elsewhere have small counts,
    queries inside the crowd collect 255 targets and more: each of those comes back as "ask again" and is recomputed
    (exact), and the running total mirrored to the host makes the index fall back to 16-bit counts within a few passes."""
    rng = np.random.default_rng(91)
    span = 30_000_000
    s = np.concatenate([rng.integers(1000, span, size=400_000), rng.integers(5_000_000, 5_400_000, size=150_000)])
    e = s + rng.integers(1, 300, size=len(s))
    nq = 32768 * 70 + 99  # (the large-batch pass takes batches of 2 Mi queries and more)
    s, e = s.astype(np.int32), e.astype(np.int32)
    ix = make_index(IntervalIndex, s, e)
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    set_opt("ivl.partition", 1)
    set_opt("ivl.flat", 1)
    set_opt("ivl.bm_hard_ppm", 10**6)  # (the crowd has cells with several duplicated coordinates: keep the cell images anyway)
    try:
        qs = rng.integers(6_000_000, span, size=nq).astype(np.int32)   # away from the crowd
        qe = (qs + rng.integers(1, 800, size=nq)).astype(np.int32)
        want, want_total = t.count_batch(qs, qe)
        got, got_total = ix.count(qs, qe)
        assert ix.flat_state()[0] == 1
        assert np.array_equal(got, want) and got_total == want_total
        assert ix.count_width() == (8, 0)
        qs = rng.integers(5_000_000, 5_400_000, size=nq).astype(np.int32)  # inside it: ~ 0.4 targets per coordinate
        qe = (qs + rng.integers(900, 1200, size=nq)).astype(np.int32)
        want, want_total = t.count_batch(qs, qe)
        assert (want >= 255).mean() > 0.9
        for _ in range(10):  # (the mirror in host memory is a pass or two behind the kernels that write it)
            got, got_total = ix.count(qs, qe)
            assert np.array_equal(got, want) and got_total == want_total
            if ix.count_width()[0] == 16:
                break
        bits, wide = ix.count_width()
        assert bits == 16 and wide > 4096 and wide * 64 > nq, (bits, wide)  # (what makes the host give the 8 bits up)
        got, got_total = ix.count(qs, qe)
        assert np.array_equal(got, want) and got_total == want_total
    finally:
        reset_opts()


def test_clustered_distribution_differential(O, IntervalIndex):
    """bxmi.synth.clustered (everything around hot spots, heavily duplicated coordinates) at 4 M queries x 1 M targets
    against the oracle treap, through whatever large-batch stage serves such an index (bitmap cells refuse it -- too many
    duplicated coordinates per cell; offset cells in the clumped layout take it since round 6, the dense unit images with
    ivl.clumped = 0), in generated order and sorted by start, and through the first-generation pass."""
    (ts, te), (qs, qe) = synth.clustered(1_000_000, 4_000_000, hot_spots=2_000, genome=25_000_000)
    t = O.OracleIntervalTree()
    t.insert_many_arrays(ts, te)
    # (the treap answers a query of this distribution in ~27 us -- mean count 700: every tenth query is checked against it,
    # all of them through the passes' agreement with each other and the totals)
    pick = np.arange(0, len(qs), 10)
    want, _ = t.count_batch(qs[pick], qe[pick])
    ix = make_index(IntervalIndex, ts, te)
    got, got_total = ix.count(qs, qe)
    stages = (ix.flat_state()[0], ix.dense_state()[0], ix.sparse_state()[0], ix.slice_state()[0])
    bad = np.nonzero(got[pick] != want)[0]
    assert len(bad) == 0 and got_total == int(got.sum(dtype=np.int64)), (stages, bad[:8], qs[pick][bad[:8]], qe[pick][bad[:8]], got[pick][bad[:8]], want[bad[:8]])
    # one of the exchange's search stages served it (round 6: offset cells in the clumped layout -- a rank table per hard cell)
    assert stages[3] == 1 or stages[0] == 1 or stages[1] == 1 or stages[2] == 2, stages
    o = np.argsort(qs, kind="stable")
    got_s, got_s_total = ix.count(qs[o], qe[o])
    assert np.array_equal(got_s, got[o]) and got_s_total == got_total, "sorted by start"
    try:
        set_opt("ivl.clumped", 0)
        got_d, got_d_total = ix.count(qs, qe)
        assert ix.dense_state()[0] == 1 or ix.slice_state()[0] == 1, (ix.dense_state(), ix.slice_state())
        set_opt("ivl.bitmap", 0)
        got_1, got_1_total = ix.count(qs, qe)
    finally:
        reset_opts()
    assert np.array_equal(got_d, got) and got_d_total == got_total, "dense unit images / key slices"
    assert np.array_equal(got_1, got) and got_1_total == got_total, "first-generation pass"


def test_c_abi_allreduce_world_of_one():
    """bxmi_comm_* / bxmi_allreduce_i64 (RCCL opened at first use): a communicator of one rank on this GPU; the sum
    all-reduce of int64 totals in place is then the identity, stream-ordered behind the kernel that produced them.  (Two
    ranks need two GPUs: RCCL refuses a second rank on the same device.  The 2-rank bookkeeping is the gloo test.)"""
    from bxmi import _ffi, shard

    comm = shard.Comm(0, 1, lambda raw: raw)
    vals = np.array([5, -7, 2**40, 0, 123456789012], dtype=np.int64)
    buf = _ffi.DeviceArray.from_numpy(vals)
    comm.allreduce_i64(buf.ptr, len(vals))
    _ffi.call("bxmi_synchronize", None)
    assert buf.to_numpy(np.int64, len(vals)).tolist() == vals.tolist()
    comm.allreduce_i64(buf.ptr, 0)
    comm.close()
    with pytest.raises(_ffi.BxmiError):
        _ffi.call("bxmi_comm_create", _ffi.C.byref(_ffi.vp()), None, 0, 1)


def test_genome_sharded_count_single_rank():
    """configs[3] shape at reduced size: 24 chromosomes, per-chromosome indexes, counts summed (world size 1 here;
    the 2-rank gloo run of the same driver is tests/test_host_logic.py)."""
    from bxmi import shard

    rng = np.random.default_rng(404)
    targets, queries = {}, {}
    for chrom, size in synth.HG19_SIZES.items():
        nt, nq = max(50, size // 20000), max(50, size // 5000)
        s = rng.integers(0, size - 1000, size=nt).astype(np.int32)
        targets[chrom] = (s, (s + rng.integers(1, 1001, size=nt)).astype(np.int32))
        q = rng.integers(0, size - 1000, size=nq).astype(np.int32)
        queries[chrom] = (q, (q + rng.integers(1, 1001, size=nq)).astype(np.int32))
    totals, per_query = shard.count_genome(targets, queries, rank=0, world=1)
    for chrom in synth.HG19_SIZES:
        ts, te = targets[chrom]
        qs, qe = queries[chrom]
        want = np.searchsorted(np.sort(ts), qe, "left") - np.searchsorted(np.sort(te), qs, "right")  # proper intervals only
        assert np.array_equal(per_query[chrom], want.astype(np.int32)), chrom
        assert totals[chrom] == int(want.sum())


def test_count_multi_on_sorted_batches(O, IntervalIndex):
    """A sorted BED file against a genome (scripts/interval_join.py:21-28 loops over the chromosomes): bxmi_ivl_count_multi_dev
    with every chromosome's queries sorted by start is answered by the walk on cell images as the queries lie (count_dense.hpp:
    bs_check_multi_kernel / bs_plan_multi_kernel / bs_walk_kernel over segments) -- against the oracle, for sparse indexes on
    offset cells and dense ones on bitmap cells, with escapes, a chromosome without queries, queries left and right of the grid,
    totals only; one chromosome in reversed or partly shuffled order sends the whole batch through the exchange, same answers."""
    from bxmi import _ffi

    rng = np.random.default_rng(515)
    for density, opt in (("sparse", ("ivl.sparse", 1)), ("dense", ("ivl.flat", 1))):
        specs = [(150_000, 50_000_000, 400_001), (260_000, 90_000_000, 700_000), (90_000, 30_000_000, 0), (200_000, 70_000_000, 250_000),
                 (120_000, 40_000_000, 16384 * 9)]
        if density == "dense":
            specs = [(n * 4, span // 3, nq) for n, span, nq in specs]  # (configs[1]'s density: one target per ~28 coordinates)
        ixs, host, want = [], [], []
        for k, (n, span, nq) in enumerate(specs):
            s = rng.integers(1000, span, size=n)
            e = s + rng.integers(1, 1000, size=n)
            qs = np.sort(rng.integers(-5000, span + 8000, size=nq))
            qe = qs + rng.integers(0, 1500, size=nq)
            if nq:
                qe[::97] = qs[::97] + 40_000  # longer than a record holds
                qe[::89] = qs[::89] - 3       # reversed
            s, e, qs, qe = (a.astype(np.int32) for a in (s, e, qs, qe))
            t = O.OracleIntervalTree()
            t.insert_many_arrays(s, e)
            want.append(t.count_batch(qs, qe))
            ixs.append(make_index(IntervalIndex, s, e))
            host.append((qs, qe))
        totals = _ffi.DeviceArray(8 * len(specs))

        def run(queries, with_counts=True):
            dev = [(_ffi.DeviceArray.from_numpy(a), _ffi.DeviceArray.from_numpy(b), _ffi.DeviceArray(4 * max(len(a), 4))) for a, b in queries]
            totals.zero()
            IntervalIndex.count_multi_dev(ixs, [d[0].ptr for d in dev], [d[1].ptr for d in dev], [len(q[0]) for q in queries],
                                          [d[2].ptr if with_counts else None for d in dev], [totals.ptr + 8 * i for i in range(len(specs))], None)
            _ffi.call("bxmi_synchronize", None)
            return [d[2].to_numpy(np.int32, len(q[0])) for d, q in zip(dev, queries)], totals.to_numpy(np.int64, len(specs))

        set_opt("ivl.partition", 1)
        set_opt("ivl.bitmap_min", 1)
        set_opt(*opt)
        try:
            for variant in (-1, 1):
                set_opt("ivl.bm_variant", variant)
                got, tot = run(host)  # every chromosome sorted
                for k, (wc, wt) in enumerate(want):
                    bad = np.nonzero(got[k] != wc)[0]
                    assert len(bad) == 0 and int(tot[k]) == wt, (density, "sorted", variant, k, bad[:5], got[k][bad[:5]], wc[bad[:5]], int(tot[k]), wt)
                _, tot = run(host, with_counts=False)  # totals only
                assert [int(x) for x in tot] == [wt for _, wt in want], (density, "totals only", variant)
            set_opt("ivl.bm_variant", -1)
            # one chromosome reversed, then one with a shuffled stretch in its middle: the exchange answers the whole batch
            for what in ("reversed", "partly"):
                queries, perm = list(host), None
                qs, qe = host[3]
                if what == "reversed":
                    perm = np.arange(len(qs))[::-1].copy()
                else:
                    perm = np.arange(len(qs))
                    perm[100_000:100_050] = perm[100_000:100_050][::-1]
                queries[3] = (np.ascontiguousarray(qs[perm]), np.ascontiguousarray(qe[perm]))
                got, tot = run(queries)
                for k, (wc, wt) in enumerate(want):
                    w = wc[perm] if k == 3 else wc
                    bad = np.nonzero(got[k] != w)[0]
                    assert len(bad) == 0 and int(tot[k]) == wt, (density, what, k, bad[:5], got[k][bad[:5]], w[bad[:5]])
            # and the sorted walk switched off (every batch through the exchange) agrees
            set_opt("ivl.sorted_cells", 0)
            got, tot = run(host)
            for k, (wc, wt) in enumerate(want):
                assert np.array_equal(got[k], wc) and int(tot[k]) == wt, (density, "sorted_cells off", k)
        finally:
            reset_opts()
        if density == "sparse":
            assert [ix.sparse_state()[0] for ix in ixs if ix is not ixs[2]] == [1, 1, 1, 1], [ix.sparse_state() for ix in ixs]
        else:
            assert [ix.flat_state()[0] for ix in ixs if ix is not ixs[2]] == [1, 1, 1, 1], [ix.flat_state() for ix in ixs]


def test_count_multi_equals_one_index_at_a_time(O, IntervalIndex):
    """bxmi_ivl_count_multi_dev (one fused bitmap-cell pass over several indexes = a dict of per-chromosome trees) against
    the oracle and against per-index calls: indexes of different spans (bucket widths 2^10 .. 2^17), one with reversed
    targets and one too wide for the bitmap cells (both answered by the per-index path inside the same call), ragged
    batch sizes incl. an empty one, escapes."""
    from bxmi import _ffi

    rng = np.random.default_rng(31)
    specs = [(60_000, 2_000_000, 300_001), (9_000, 250_000_000, 70_000), (200_000, 6_000_000, 16384 * 5), (5_000, 900_000, 0),
             (30_000, 2**30, 50_000), (20_000, 5_000_000, 40_000)]
    ixs, dev, want = [], [], []
    for k, (n, span, nq) in enumerate(specs):
        s = rng.integers(0, span, size=n)
        e = s + rng.integers(0, 2000, size=n)
        if k == 5:
            e[3] = s[3] - 7  # reversed target: this index stays on the direct kernel
        qs = rng.integers(-1000, span + 3000, size=nq)
        qe = qs + rng.integers(0, 4000, size=nq)
        if nq:
            qe[::97] = qs[::97] + 40_000  # longer than a record holds
            qe[::89] = qs[::89] - 3       # reversed
        s, e, qs, qe = (a.astype(np.int32) for a in (s, e, qs, qe))
        t = O.OracleIntervalTree()
        t.insert_many_arrays(s, e)
        want.append(t.count_batch(qs, qe))
        ixs.append(make_index(IntervalIndex, s, e))
        dev.append((_ffi.DeviceArray.from_numpy(qs), _ffi.DeviceArray.from_numpy(qe), _ffi.DeviceArray(4 * max(nq, 4)), nq))
    totals = _ffi.DeviceArray(8 * len(specs))
    set_opt("ivl.partition", 1)
    try:
        # all defaults: indexes 0 and 2 (dense) ride one pass on dense unit images, 1 and 4 (sparse / wide) one on slices
        totals.zero()
        IntervalIndex.count_multi_dev(ixs, [d[0].ptr for d in dev], [d[1].ptr for d in dev], [d[3] for d in dev], [d[2].ptr for d in dev],
                                      [totals.ptr + 8 * i for i in range(len(specs))], None)
        _ffi.call("bxmi_synchronize", None)
        tot = totals.to_numpy(np.int64, len(specs))
        for k, (wc, wt) in enumerate(want):
            got = dev[k][2].to_numpy(np.int32, dev[k][3])
            bad = np.nonzero(got != wc)[0]
            assert len(bad) == 0 and int(tot[k]) == wt, ("defaults", k, ixs[k].dense_state(), bad[:5], got[bad[:5]], wc[bad[:5]], int(tot[k]), wt)
        assert [ix.flat_state()[0] for ix in ixs] == [1, 0, 1, 0, 0, 0], [ix.flat_state() for ix in ixs]
        assert [ix.slice_state()[0] for ix in ixs] == [0, 1, 0, 0, 1, 0]
        set_opt("ivl.sparse", 1)  # the sparse ones on offset-cell images however small their batches (one pass for the two of them)
        totals.zero()
        IntervalIndex.count_multi_dev(ixs, [d[0].ptr for d in dev], [d[1].ptr for d in dev], [d[3] for d in dev], [d[2].ptr for d in dev],
                                      [totals.ptr + 8 * i for i in range(len(specs))], None)
        _ffi.call("bxmi_synchronize", None)
        tot = totals.to_numpy(np.int64, len(specs))
        for k, (wc, wt) in enumerate(want):
            got = dev[k][2].to_numpy(np.int32, dev[k][3])
            bad = np.nonzero(got != wc)[0]
            assert len(bad) == 0 and int(tot[k]) == wt, ("sparse", k, ixs[k].sparse_state(), bad[:5], got[bad[:5]], wc[bad[:5]], int(tot[k]), wt)
        assert [ix.sparse_state()[0] for ix in ixs] == [0, 1, 0, 0, 1, 0], [ix.sparse_state() for ix in ixs]
        set_opt("ivl.sparse", -1)
        set_opt("ivl.flat", 0)  # the same on dense unit images
        totals.zero()
        IntervalIndex.count_multi_dev(ixs, [d[0].ptr for d in dev], [d[1].ptr for d in dev], [d[3] for d in dev], [d[2].ptr for d in dev],
                                      [totals.ptr + 8 * i for i in range(len(specs))], None)
        _ffi.call("bxmi_synchronize", None)
        tot = totals.to_numpy(np.int64, len(specs))
        for k, (wc, wt) in enumerate(want):
            got = dev[k][2].to_numpy(np.int32, dev[k][3])
            bad = np.nonzero(got != wc)[0]
            assert len(bad) == 0 and int(tot[k]) == wt, ("dense", k, ixs[k].dense_state(), bad[:5], got[bad[:5]], wc[bad[:5]], int(tot[k]), wt)
        assert [ix.dense_state()[0] for ix in ixs] == [1, 0, 1, 0, 0, 0], [ix.dense_state() for ix in ixs]
        for ix in ixs:
            ix.seal()  # (forget the stages chosen so far: the loop below asserts on what each setting prepares)
        set_opt("ivl.dense", 0)
        for variant, slices in ((-1, 0), (0, -1), (2, -1), (1, 1)):  # no image stage left: key slices where allowed, else one index at a time
            set_opt("ivl.bm_variant", variant)
            set_opt("ivl.slice", slices)
            totals.zero()
            IntervalIndex.count_multi_dev(ixs, [d[0].ptr for d in dev], [d[1].ptr for d in dev], [d[3] for d in dev], [d[2].ptr for d in dev],
                                          [totals.ptr + 8 * i for i in range(len(specs))], None)
            _ffi.call("bxmi_synchronize", None)
            tot = totals.to_numpy(np.int64, len(specs))
            for k, (wc, wt) in enumerate(want):
                got = dev[k][2].to_numpy(np.int32, dev[k][3])
                bad = np.nonzero(got != wc)[0]
                assert len(bad) == 0 and int(tot[k]) == wt, (variant, slices, k, ixs[k].slice_state(), bad[:5], got[bad[:5]], wc[bad[:5]], int(tot[k]), wt)
            if slices == 0:
                assert [ix.slice_state()[0] for ix in ixs] == [0] * 6
        assert [ix.slice_state()[0] for ix in ixs] == [1, 1, 1, 0, 1, 0]
    finally:
        reset_opts()


def test_genome_cfg4_full_size_golden(golden_scale_doc, IntervalIndex):
    """BASELINE configs[3] at full size on one GPU: synth.cfg4 (24 chromosomes, 10M targets x 100M queries by chromosome
    length), one index per chromosome.  Every 100th count of every chromosome against the reference treap's hash
    (tests/golden/scale.json "cfg4_genome", made by oracle/gen_golden.py --only genome), totals = sums."""
    g = golden_scale_doc.get("cfg4_genome")
    assert g, "tests/golden/scale.json lacks cfg4_genome"
    paths = set()
    grand = 0
    for chrom, pt in g["chroms"].items():
        (ts, te), (qs, qe) = synth.cfg4_chrom(chrom)
        assert len(ts) == pt["n_targets"] and len(qs) == pt["n_queries_total"]
        ix = make_index(IntervalIndex, ts, te)
        counts, total = ix.count(qs, qe)
        assert total == int(counts.sum(dtype=np.int64)), chrom
        sub = np.ascontiguousarray(counts[:: g["stride"]])
        assert int(sub.sum(dtype=np.int64)) == pt["total"], chrom
        assert hashlib.sha256(sub.tobytes()).hexdigest() == pt["counts_sha256"], chrom
        paths.add((ix.flat_state()[0], ix.slice_state()[0], ix.sparse_state()[0]))
        grand += total
        ix.close()
    # big chromosomes through the large-batch pass -- on offset-cell images, a chromosome has one target per ~300 coordinates --
    # small ones (< 2 Mi queries) through the direct kernel
    assert paths == {(0, 0, 1), (0, 0, 0)}, paths
    assert grand > 0


GENOME_WORKER = r"""
import os, sys, json, hashlib
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "bx-python_amd"))
import numpy as np, torch, torch.distributed as dist
from bxmi import shard, synth, _ffi
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")     # the collective of this test; both ranks drive the SAME GPU through libbxmi
_ffi.call("bxmi_set_device", 0)
scale = int(sys.argv[2])
g = json.load(open(os.path.join(sys.argv[1], "tests", "golden", "scale.json")))["cfg4_genome"]
chroms = list(synth.HG19_SIZES)
weights = {c: synth.cfg4_sizes(10_000_000 // scale)[c] + synth.cfg4_sizes(100_000_000 // scale)[c] for c in chroms}
if len(sys.argv) > 3 and sys.argv[3] == "heaviest-of-8":
    # only the chromosomes the 8-rank deal hands its heaviest rank, dealt again to the two ranks of this test
    deal8 = shard.lpt_assign(weights, 8)
    heavy = max(deal8, key=lambda cs: sum(weights[c] for c in cs))
    chroms = [c for c in chroms if c in heavy]
    weights = {c: weights[c] for c in chroms}
    assert len(chroms) >= 2, chroms
mine = shard.lpt_assign(weights, world)[rank]
tg = {c: synth.cfg4_chrom(c, 10_000_000 // scale, 100_000_000 // scale)[0] for c in mine}
qr = {c: synth.cfg4_chrom(c, 10_000_000 // scale, 100_000_000 // scale)[1] for c in mine}
# every rank passes only what it owns; count_genome deals by the same LPT rule, so ownership must agree
tg_all = {c: tg.get(c, (np.zeros(0, np.int32),) * 2) for c in chroms}
qr_all = {c: qr.get(c, (np.zeros(0, np.int32),) * 2) for c in chroms}
totals, per_query = shard.count_genome(tg_all, qr_all, rank, world, weights=weights)
assert sorted(per_query) == sorted(mine)
for c in mine:
    ts, te = tg[c]; qs, qe = qr[c]
    want = (np.searchsorted(np.sort(ts), qe, "left") - np.searchsorted(np.sort(te), qs, "right")).astype(np.int32)   # proper intervals
    assert np.array_equal(per_query[c], want), c
    if scale == 1:
        sub = np.ascontiguousarray(per_query[c][:: g["stride"]])
        assert hashlib.sha256(sub.tobytes()).hexdigest() == g["chroms"][c]["counts_sha256"], c
# the reduced vector holds every chromosome's total on every rank
obj = [None] * world
dist.all_gather_object(obj, {c: int(per_query[c].sum(dtype=np.int64)) for c in mine})
full = {}
for d in obj: full.update(d)
assert totals == {c: full[c] for c in chroms}, (rank, totals, full)
dist.barrier(); dist.destroy_process_group()
sys.stdout.write("rank%d-ok %d chromosomes %d overlaps hashes-checked=%d\n" % (rank, len(mine), sum(totals.values()), len(mine) if scale == 1 else 0))
"""


def test_genome_two_ranks_real_engine(tmp_path):
    """configs[3]'s multi-GPU shape with the REAL engine: two processes (world size 2, gloo all-reduce), both on cuda:0,
    each builds and queries the chromosomes LPT deals it (1/4 of the full genome size), totals reduced across ranks."""
    import subprocess
    import sys as _sys

    script = tmp_path / "genome_worker.py"
    script.write_text(GENOME_WORKER)
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    cmd = [_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           str(script), root, "4"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "rank0-ok" in p.stdout and "rank1-ok" in p.stdout, p.stdout[-2000:]


def test_genome_two_ranks_full_size_heaviest_share(tmp_path):
    """The same two-rank run at FULL size (scale 1) on the chromosomes the 8-rank LPT deal hands its heaviest rank
    (bxmi/shard.py; scripts/interval_join.py:21-28 is the dict of per-chromosome trees being dealt): every 100th count of each
    chromosome against the reference treap's hash (tests/golden/scale.json "cfg4_genome") on the count_genome path, totals
    all-reduced across the two ranks."""
    import re
    import subprocess
    import sys as _sys

    script = tmp_path / "genome_worker.py"
    script.write_text(GENOME_WORKER)
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    cmd = [_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29543",
           str(script), root, "1", "heaviest-of-8"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    checked = [int(x) for x in re.findall(r"rank\d-ok \d+ chromosomes \d+ overlaps hashes-checked=(\d+)", p.stdout)]
    assert len(checked) == 2 and min(checked) >= 1 and sum(checked) >= 2, p.stdout[-2000:]


def test_find_on_sorted_batches_flat_fill(O, IntervalIndex):
    """find() on a batch sorted by start: the fill that stages a wave's window of pairs and its stretch of the hit list in LDS
    (part_fill_pipe_kernel) against the oracle's hit lists --
    ordinary stretches, queries on a pile (a wave's stretch beyond the LDS image: direct stores), a few very long targets far
    below the queries' windows (lanes that leave the staged pairs, walks handed to the whole wave), queries without hits."""
    rng = np.random.default_rng(4242)
    n, span = 400_000, 8_000_000
    s = rng.integers(0, span, size=n)
    e = s + rng.integers(1, 200, size=n)
    s[:3000] = span // 2 + rng.integers(0, 40, size=3000)      # a pile: queries there see thousands of hits
    e[:3000] = s[:3000] + 100
    e[3000:3040] = s[3000:3040] + rng.integers(500_000, 4_000_000, size=40)  # long targets: hits far below hi
    nq = 300_001
    qs = np.sort(rng.integers(-1000, span + 1000, size=nq))
    qe = qs + rng.integers(0, 300, size=nq)                     # (zero-length ones included)
    qe[::17] = qs[::17] - 5                                     # reversed
    s, e, qs, qe = (a.astype(np.int32) for a in (s, e, qs, qe))
    t = O.OracleIntervalTree()
    t.insert_many_arrays(s, e)
    w_off, w_hits = t.find_batch(qs, qe)
    ix = make_index(IntervalIndex, s, e)
    set_opt("ivl.partition", 1)
    try:
        got = ix.find(qs, qe)
        assert np.array_equal(got[0], w_off), ("offsets", np.nonzero(got[0] != w_off)[0][:8])
        bad = np.nonzero(got[1] != w_hits)[0]
        assert len(bad) == 0, ("hits", bad[:8], got[1][bad[:8]], w_hits[bad[:8]])
        assert int(np.diff(w_off).max()) > 3000 and len(w_hits) > 5 * nq  # the pile and the long targets are really in play
        # a buffer that is too small: the total comes back, the wrapper retries
        got = ix.find(qs, qe, cap_hint=len(w_hits) // 3)
        assert np.array_equal(got[0], w_off) and np.array_equal(got[1], w_hits), "after BXMI_ERANGE"
        _erange_contract(ix, qs, qe, w_off)
        got = ix.find(qs, qe, cap_hint=len(w_hits))  # exactly enough
        assert np.array_equal(got[0], w_off) and np.array_equal(got[1], w_hits)
    finally:
        reset_opts()


def test_find_almost_sorted_batch_stands_down(IntervalIndex):
    """A large batch that passes the order PROBE (two stretches of 4096 starts) and fails the exact check: the sorted find's chain
    is launched behind the check without waiting for the host (round 6), every kernel of it stands down on the check's word, and the
    batch is answered through the exchange -- same lists as the sorted batch's, with the two swapped queries' lists swapped."""
    rng = np.random.default_rng(777)
    n, span = 300_000, 30_000_000
    s = rng.integers(0, span, size=n)
    e = s + rng.integers(1, 400, size=n)
    nq = (1 << 21) + 77
    qs = np.sort(rng.integers(0, span, size=nq))
    qe = qs + rng.integers(1, 300, size=nq)
    s, e, qs, qe = (a.astype(np.int32) for a in (s, e, qs, qe))
    ix = make_index(IntervalIndex, s, e)
    off_s, hits_s = ix.find(qs, qe)
    for at in (5, nq - 9):  # far from the probe's stretches (at nq / 3 and 2 nq / 3)
        assert qs[at] < qs[at + 1]
        q2s, q2e = qs.copy(), qe.copy()
        q2s[[at, at + 1]] = q2s[[at + 1, at]]
        q2e[[at, at + 1]] = q2e[[at + 1, at]]
        off_w, hits_w = ix.find(q2s, q2e)
        cnt_s, cnt_w = np.diff(off_s), np.diff(off_w)
        want = cnt_s.copy()
        want[[at, at + 1]] = want[[at + 1, at]]
        assert np.array_equal(cnt_w, want) and off_w[-1] == off_s[-1]
        lo = off_s[at]
        assert np.array_equal(hits_w[:lo], hits_s[:lo]) and np.array_equal(hits_w[off_s[at + 2]:], hits_s[off_s[at + 2]:])
        assert np.array_equal(hits_w[off_w[at]:off_w[at + 1]], hits_s[off_s[at + 1]:off_s[at + 2]])
        assert np.array_equal(hits_w[off_w[at + 1]:off_w[at + 2]], hits_s[off_s[at]:off_s[at + 1]])
    # and the sorted batch again behind it: the same lists as before
    off_a, hits_a = ix.find(qs, qe)
    assert np.array_equal(off_a, off_s) and np.array_equal(hits_a, hits_s)
    c, total = ix.count(qs, qe)
    assert np.array_equal(c, np.diff(off_s)) and total == off_s[-1]


def test_find_join_scale_properties(IntervalIndex):
    """configs[4] shape at 4M x 4M (the 50M x 50M run is tools/bench_find.py): CSR consistency, every hit overlaps,
    hits of a query in tree order, and agreement with the count path."""
    (ts, te), (qs, qe) = synth.cfg5(4_000_000, 4_000_000)
    ix = make_index(IntervalIndex, ts, te)
    set_opt("ivl.partition", 0)
    try:
        d_offs, d_hits = ix.find(qs, qe, cap_hint=8 * len(qs))  # direct tree kernels
    finally:
        set_opt("ivl.partition", 1)
    try:
        offs, hits = ix.find(qs, qe, cap_hint=8 * len(qs))      # through the exchange, the fill on LDS windows (244 tiles, several tile chunks per piece)
        set_opt("ivl.fx_direct", 0)
        o_offs, o_hits = ix.find(qs, qe, cap_hint=8 * len(qs))  # the fill into scratch, then the copy
        assert np.array_equal(o_offs, offs) and np.array_equal(o_hits, hits)
        set_opt("ivl.find_sliced", 0)
        o_offs, o_hits = ix.find(qs, qe, cap_hint=8 * len(qs))  # the bucketed find of the first generation
        assert np.array_equal(o_offs, offs) and np.array_equal(o_hits, hits)
        del o_offs, o_hits
    finally:
        reset_opts()
    assert np.array_equal(d_offs, offs) and np.array_equal(d_hits, hits)
    counts, total = ix.count(qs, qe)
    assert offs[-1] == total == len(hits) and np.array_equal(np.diff(offs).astype(np.int32), counts)
    rep = np.repeat(np.arange(len(qs)), counts)
    assert (te[hits] > qs[rep]).all() and (ts[hits] < qe[rep]).all()
    key = ts[hits].astype(np.int64) * (1 << 31) + hits
    same = rep[1:] == rep[:-1]
    assert (key[1:][same] >= key[:-1][same]).all()
    # brute-force spot check of 200 queries
    for i in np.random.default_rng(1).integers(0, len(qs), size=200):
        want = np.nonzero((te > qs[i]) & (ts < qe[i]))[0]
        got = hits[offs[i]:offs[i + 1]]
        assert sorted(got.tolist()) == want.tolist()


def test_find_join_cfg5_full_size_golden(golden_scale_doc, IntervalIndex):
    """BASELINE configs[4] at full size: 50M x 50M (synth.cfg5), CSR hit lists.  Every 500th query's hit LIST (insertion
    indices in the reference's order) against the reference treap over all 50M targets (tests/golden/scale.json
    "cfg5_join", made by oracle/gen_golden.py --only join), both for the batch as generated (bucketed find) and sorted by
    start (local find); the whole result against the count pass."""
    g = golden_scale_doc.get("cfg5_join")
    assert g, "tests/golden/scale.json has no cfg5_join point: the reference check of configs[4] must not vanish silently"
    (ts, te), (qs, qe) = synth.cfg5(g["n_targets"], g["n_queries_total"])
    ix = make_index(IntervalIndex, ts, te)
    offs, hits = ix.find(qs, qe, cap_hint=6 * len(qs))
    lens = np.diff(offs)
    st = g["stride"]
    sub = np.arange(0, len(qs), st)
    assert len(sub) == g["n_queries"]
    sub_counts = lens[sub].astype(np.int32)
    assert int(sub_counts.sum(dtype=np.int64)) == g["total"]
    assert hashlib.sha256(np.ascontiguousarray(sub_counts).tobytes()).hexdigest() == g["counts_sha256"]
    sub_hits = np.concatenate([hits[offs[i]:offs[i + 1]] for i in sub.tolist()]).astype(np.int32)
    assert sub_hits[:16].tolist() == g["first_hits"]
    assert hashlib.sha256(np.ascontiguousarray(sub_hits).tobytes()).hexdigest() == g["hits_sha256"], "hit lists differ from the reference's"
    counts, total = ix.count(qs, qe)
    assert total == offs[-1] == len(hits) and np.array_equal(counts, lens.astype(np.int32))
    # the same queries sorted by start: every query must get the same list
    order = np.argsort(qs, kind="stable")
    s_offs, s_hits = ix.find(qs[order], qe[order], cap_hint=6 * len(qs))
    assert np.array_equal(np.diff(s_offs), lens[order])
    inv = np.empty(len(order), dtype=np.int64)
    inv[order] = np.arange(len(order))
    s_sub = np.concatenate([s_hits[s_offs[j]:s_offs[j + 1]] for j in inv[sub].tolist()]).astype(np.int32)
    assert np.array_equal(s_sub, sub_hits)


# -------------------------------------------------------------- compat API --
def test_compat_intervaltree_known_answers():
    """lib/bx/intervals/intersection_tests.py:158-201 and the doctests intersection.pyx:335-376."""
    from bx.intervals.intersection import Intersecter, Interval, IntervalTree

    assert Intersecter is IntervalTree
    iv = IntervalTree()
    n = 0
    for i in range(1, 1000, 80):
        iv.insert(i, i + 10, {"value": i * i})
        iv.add(i + 20, i + 30, {"astr": str(i * i)})
        iv.insert_interval(Interval(i + 40, i + 50, value={"astr": str(i * i)}))
        iv.add_interval(Interval(i + 60, i + 70, value={"astr": str(i * i)}))
        n += 4
    assert len(iv.find(100, 200)) == 5
    a = []
    iv.traverse(a.append)
    assert len(a) == n
    iv.traverse(lambda node: node.interval)
    e = IntervalTree()
    assert e.find(100, 300) == [] and e.after(100) == [] and e.before(100) == []
    assert e.after_interval(100) == [] and e.before_interval(100) == []
    assert e.upstream_of_interval(100) == [] and e.downstream_of_interval(100) == []
    assert e.traverse(lambda x: x.append(1)) is None

    t = IntervalTree()
    t.insert(0, 10, "food")
    t.insert(3, 7, dict(foo="bar"))
    assert t.find(2, 5) == ["food", {"foo": "bar"}]
    t = IntervalTree()
    for a_, b_ in ((0, 10), (3, 7), (3, 40), (13, 50)):
        t.insert_interval(Interval(a_, b_))
    assert repr(t.find(30, 50)) == "[Interval(3, 40), Interval(13, 50)]"
    assert t.find(100, 200) == []
    assert repr(t.before_interval(Interval(10, 20))) == "[Interval(3, 7)]"
    assert t.before_interval(Interval(5, 20)) == []
    assert repr(t.upstream_of_interval(Interval(11, 12))) == "[Interval(0, 10)]"
    assert repr(t.upstream_of_interval(Interval(11, 12, strand="-"))) == "[Interval(13, 50)]"
    assert repr(t.upstream_of_interval(Interval(1, 2, strand="-"), num_intervals=3)) == "[Interval(3, 7), Interval(3, 40), Interval(13, 50)]"
    with pytest.raises(OverflowError):
        t.find(2**31, 2**31 + 5)
    with pytest.raises(TypeError):
        t.find("a", 5)
    assert repr(t.find(2.5, 4.5)) == "[Interval(0, 10), Interval(3, 7), Interval(3, 40)]"  # floats truncate to (2, 4)


def test_compat_intervalnode_neighbours():
    """intersection_tests.py:17-54 (NeighborTestCase) and :104-141 (LotsaTestCase, reduced to what stays fast)."""
    from bx.intervals.intersection import Interval, IntervalNode

    iv = IntervalNode(50, 59, Interval(50, 59))
    for i in range(0, 110, 10):
        if i == 50:
            continue
        f = Interval(i, i + 9)
        iv = iv.insert(f.start, f.end, f)
    assert str(iv.left(60, n=2)) == str([Interval(50, 59), Interval(40, 49)])
    for i in range(10, 100, 10):
        assert iv.left(i, max_dist=10, n=1)[0].end == i - 1
    assert len(iv.left(60, n=200)) == 6
    for i in range(10, 100, 10):
        r = iv.right(i + 1, n=1)
        assert len(r) == 1 and r[0].start == i + 10
    for i in range(0, 100, 10):
        assert iv.right(i - 1, max_dist=10, n=1)[0].start == i

    big = IntervalNode(1, 2, Interval(1, 2))
    for i in range(0, 1000000, 10):
        big = big.insert(i, i, Interval(i, i))
    for i in range(600):
        big = big.insert(0, 1, Interval(0, 1))
    assert len(big.right(1, n=33)) == 33
    assert len(big.left(1, n=33)) == 1
    assert len(big.right(1, n=9999)) == 250
    assert len(big.right(1, n=9999, max_dist=99999)) == 9999
    assert len(big.right(1, max_dist=0, n=10)) == 0
    for n, d in enumerate(range(10, 1000, 100)):
        assert len(big.right(1, max_dist=d, n=10000)) == 10 * n + 1
    for (qs, qe) in ((1000, 5000), (123456, 130000)):
        for feat in big.find(qs, qe):
            assert (qs <= feat.end <= qe) or (qs <= feat.start <= qe)
    assert repr(big) .startswith("IntervalNode(")
