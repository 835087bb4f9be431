"""
GPU parity tests of the bitset path (run on the MI355X box with -m gpu).

Everything goes through the C ABI (libbxmi.so); the checker is the CPU oracle
(oracle/binbits.c, itself pinned to the reference) and the committed
reference-generated op sequences.  Bit-exact.
"""
import numpy as np
import pytest

from bitset_replay import replay
from bxmi import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


@pytest.fixture(scope="module")
def B():
    import bx.bitset

    return bx.bitset


# ------------------------------------------------------------------ golden --
def test_compat_binnedbitset_matches_reference_vectors(golden_bitsets, B):
    for case in golden_bitsets["cases"]:
        replay(case, B.BinnedBitSet)


def test_compat_binnedbitset_big_sizes(golden_bitsets, B):
    for case in golden_bitsets["big"]:
        replay(case, B.BinnedBitSet, check_final=False)


def test_compat_ctor_limits(golden_bitsets, B):
    assert B.MAX == golden_bitsets["MAX"]
    for size, want in golden_bitsets["ctor"]:
        try:
            got = ["ok", B.BinnedBitSet(size).size]
        except ValueError as ex:
            got = ["ValueError", str(ex)]
        assert got == want


@pytest.mark.parametrize("kind", ["BitSet", "BinnedBitSet"])
def test_reference_known_answers(B, kind):
    """lib/bx/bitset_tests.py:12-119, both classes."""

    def new(size):
        return B.BitSet(size) if kind == "BitSet" else B.BinnedBitSet(size, size % 11)

    def bits_of(b):
        return [b[i] for i in range(b.size)]

    with pytest.raises(ValueError):
        new(4000000000)
    b = new(100)
    with pytest.raises(IndexError):
        b.set(-5)
    with pytest.raises(IndexError):
        b.set(110)
    want = [0] * 100
    assert bits_of(b) == want
    for pos in (11, 14, 70, 16):
        b.set(pos)
        want[pos] = 1
    for pos in (14, 80, 16):
        b.clear(pos)
        want[pos] = 0
    assert bits_of(b) == want
    b = new(100)
    want = [0] * 100
    for s, e in ((11, 14), (20, 75), (90, 99)):
        b.set_range(s, e - s)
        want[s:e] = [1] * (e - s)
    assert bits_of(b) == want
    b = new(100)
    for s, e in ((11, 14), (20, 75), (90, 100)):
        b.set_range(s, e - s)
    assert [b.count_range(0, 0), b.count_range(0, 20), b.count_range(25, 25), b.count_range(80, 20), b.count_range(0, 100)] == [0, 3, 25, 10, 68]
    assert [b.next_set(0), b.next_set(13), b.next_set(15)] == [11, 13, 20]
    assert [b.next_clear(0), b.next_clear(11), b.next_clear(20), b.next_clear(92)] == [0, 14, 75, 100]
    x, y = new(100), new(100)
    x.set_range(20, 40), y.set_range(50, 25)
    x.iand(y)
    assert bits_of(x) == [1 if 50 <= i < 60 else 0 for i in range(100)]
    x, y = new(100), new(100)
    x.set_range(20, 40), y.set_range(50, 25)
    x.ior(y)
    assert bits_of(x) == [1 if 20 <= i < 75 else 0 for i in range(100)]
    z = new(100)
    z.set_range(20, 40)
    z.invert()
    assert bits_of(z) == [0 if 20 <= i < 60 else 1 for i in range(100)]


def test_flat_bitset_extras(B):
    """BitSet-only surface: ixor, clone, count_range defaults, next_set(start, end), operators (bitset.pyx:125-173)."""
    a, b = B.BitSet(200), B.BitSet(200)
    a.set_range(10, 50), b.set_range(40, 60)
    c = a.clone()
    c.ixor(b)
    assert [c[i] for i in range(200)] == [1 if (10 <= i < 40 or 60 <= i < 100) else 0 for i in range(200)]
    assert a.count_range() == 50 and a.count_range(30) == 30
    assert a.next_set(0, 5) == 5 and a.next_set(0) == 10 and a.next_clear(10, 30) == 30 and a.next_clear(10) == 60
    a &= b
    assert a.count_range(0, 200) == 20
    a |= b
    assert a.count_range(0, 200) == 60
    a = ~a
    assert a.count_range(0, 200) == 140 and a.count_range(40, 60) == 0  # no ALL_ONE arithmetic on a flat set
    with pytest.raises(IndexError):
        a.next_set(5, 3)
    with pytest.raises(TypeError):
        a.iand(B.BinnedBitSet(200))


# ------------------------------------------------------- oracle differential --
@pytest.mark.parametrize("size,gran", [(100, 1), (997, 7), (5000, 64), (70000, 1024), (1 << 20, 1024), (12345, 10), (4097, 4097), (1 << 22, 3)])
def test_random_differential_batch(O, size, gran):
    from bxmi.bitset import DeviceBitSet

    rng = np.random.default_rng(size + gran)
    dev = [DeviceBitSet(size, gran) for _ in range(2)]
    ora = [O.OracleBinnedBitSet(size, gran) for _ in range(2)]
    assert (dev[0].bin_size, dev[0].nbins) == (ora[0].bin_size, ora[0].nbins)
    for step in range(40):
        w = int(rng.integers(0, 2))
        op = int(rng.integers(0, 9))
        m = int(rng.integers(1, 300))
        s = rng.integers(0, size, size=m)
        if step % 5 == 0:  # long ranges too: many words, many bins
            n = rng.integers(0, size, size=m)
        else:
            n = rng.integers(0, max(2, min(size, 3000)), size=m)
        n = np.minimum(n, size - s)
        if op <= 2:
            dev[w].set_ranges(s, n), ora[w].set_ranges(s, n)
        elif op == 3:
            assert dev[w].count_ranges(s, n).tolist() == ora[w].count_ranges(s, n).tolist(), (size, gran, step)
        elif op == 4:
            for p in s[:6].tolist():
                assert dev[w].next(p, 1) == ora[w].next_set(p), (size, gran, step, p)
                assert dev[w].next(p, 0) == ora[w].next_clear(p), (size, gran, step, p)
        elif op == 5:
            for p in s[:4].tolist():
                dev[w].set(p), ora[w].set(p)
                dev[1 - w].clear(p), ora[1 - w].clear(p)
                assert dev[w].get(p) == 1 and dev[1 - w].get(p) == 0
        elif op == 6 and rng.random() < 0.5:
            dev[w].invert(), ora[w].invert()
        elif op == 7:
            dev[w].iand(dev[1 - w]), ora[w].iand(ora[1 - w])
        elif op == 8:
            dev[w].ior(dev[1 - w]), ora[w].ior(ora[1 - w])
        for k in range(2):
            assert dev[k].count_range(0, size) == ora[k].count_range(0, size), (size, gran, step, k)
    for k in range(2):
        assert dev[k].bin_states().tolist() == ora[k].states().tolist()
        assert np.array_equal(dev[k].to_bits(), ora[k].unpack())
        rs, re = dev[k].runs()
        ors, ore = ora[k].runs()
        assert np.array_equal(rs, ors) and np.array_equal(re, ore)
        mid = size // 3
        rs2, re2 = dev[k].runs(mid)
        keep = ore > mid
        assert np.array_equal(re2, ore[keep]) and np.array_equal(rs2, np.maximum(ors[keep], mid))


def test_and_count_fused(O):
    from bxmi.bitset import DeviceBitSet

    size = 3_000_001
    rng = np.random.default_rng(11)
    a, b = DeviceBitSet(size), DeviceBitSet(size)
    oa, ob = O.OracleBinnedBitSet(size), O.OracleBinnedBitSet(size)
    for d, o, seed in ((a, oa, 1), (b, ob, 2)):
        s = rng.integers(0, size - 2000, size=5000)
        n = rng.integers(1, 2000, size=5000)
        d.set_ranges(s, n), o.set_ranges(s, n)
    b.invert(), ob.invert()  # padding bits of the last bin become ones: they must not be counted
    got = a.and_count(b)
    oa.iand(ob)
    assert got == oa.count_range(0, size)
    assert np.array_equal(a.to_bits(), oa.unpack())


def test_group_ops_match_per_member_ops(O):
    """One launch over many bitsets == the per-chromosome loop (sizes chosen to straddle chunk edges)."""
    from bxmi.bitset import BitSetGroup, DeviceBitSet

    rng = np.random.default_rng(21)
    sizes = [100, 524288 + 77, 3_000_001, 4096 * 128, 65, 9_999_999]
    A = [DeviceBitSet(s) for s in sizes]
    B = [DeviceBitSet(s) for s in sizes]
    OA = [O.OracleBinnedBitSet(s) for s in sizes]
    OB = [O.OracleBinnedBitSet(s) for s in sizes]
    for dev, ora in ((A, OA), (B, OB)):
        for d, o in zip(dev, ora):
            m = 400
            s = rng.integers(0, d.size, size=m)
            n = np.minimum(rng.integers(0, max(2, d.size // 50), size=m), d.size - s)
            d.set_ranges(s, n), o.set_ranges(s, n)
    B[2].invert(), OB[2].invert()  # padding bits + ALL_ONE tags inside one member
    ga, gb = BitSetGroup(A), BitSetGroup(B)
    assert ga.popcounts().tolist() == [o.count_range(0, o.size) for o in OA]
    counts = ga.iand(gb, want_counts=True)
    for oa, ob in zip(OA, OB):
        oa.iand(ob)
    assert counts.tolist() == [o.count_range(0, o.size) for o in OA]
    for d, o in zip(A, OA):
        assert d.bin_states().tolist() == o.states().tolist()
        assert np.array_equal(d.to_bits(), o.unpack())
    ga.ior(gb)
    for oa, ob in zip(OA, OB):
        oa.ior(ob)
    assert ga.popcounts().tolist() == [int(o.unpack().sum()) for o in OA]
    for d, o in zip(A, OA):
        assert d.bin_states().tolist() == o.states().tolist()
        assert d.count_range(0, d.size) == o.count_range(0, o.size)


def test_group_launches_are_gated_on_bytes(B, monkeypatch):
    """bxmi.builders.group_coverage / group_iand (what bed_coverage / bed_intersect_basewise / bed_subtract_basewise call):
    a group launch makes every member allocate its whole word array (64 MiB for a default-sized set), so the gate is the
    bytes that would become resident against the device memory free right now (bxmi_mem_info) -- 100 default-sized sets with
    a few ranges each must NOT grow to 6.4 GB when memory is tight (here: the allowed fraction turned down), give the same
    answers through either path, and an out-of-memory group falls back to the per-set loop."""
    import gc

    from bxmi import _ffi, builders

    def free_bytes():
        free = _ffi.i64(0)
        _ffi.call("bxmi_mem_info", _ffi.C.byref(free), None)
        return free.value

    rng = np.random.default_rng(77)
    gc.collect()
    B.BinnedBitSet(100).count_range(0, 10)

    def make(n):
        out = []
        for _ in range(n):
            b = B.BinnedBitSet()
            for s, c in zip(rng.integers(0, 2_000_000, size=4).tolist(), rng.integers(1, 500, size=4).tolist()):
                b.set_range(s, c)
            out.append(b)
        return out

    xs, ys = make(100), make(100)
    want_cov = sum(b.count_range(0, b.size) for b in xs)
    assert builders.group_bytes(xs, ys) == 200 * ((xs[0].size + 7) // 8)
    free0 = free_bytes()
    monkeypatch.setattr(builders, "GROUP_MAX_FRACTION_OF_FREE", 1e-4)  # "memory is tight"
    assert not builders.group_fits(xs) and not builders.group_fits(xs, ys)
    assert builders.group_coverage(xs) == want_cov
    ands = [B.BinnedBitSet() for _ in xs]
    for a, x in zip(ands, xs):
        a.ior(x)
    builders.group_iand(ands, ys)
    want_and = [a.count_range(0, a.size) for a in ands]
    assert free0 - free_bytes() < (1 << 30), "the per-set loop must leave untouched bins unallocated"
    monkeypatch.setattr(builders, "GROUP_MAX_FRACTION_OF_FREE", 0.25)
    assert builders.group_fits(xs, ys)  # 12.8 GB against a 288 GB device
    assert builders.group_coverage(xs) == want_cov
    ands2 = [B.BinnedBitSet() for _ in xs]
    for a, x in zip(ands2, xs):
        a.ior(x)
    builders.group_iand(ands2, ys)
    assert [a.count_range(0, a.size) for a in ands2] == want_and

    # a group that runs out of memory falls back to the loop
    def boom(*a, **k):
        raise _ffi.BxmiError(_ffi.ENOMEM, "no room (test)")

    monkeypatch.setattr(builders, "as_group", boom)
    assert builders.group_coverage(xs) == want_cov
    assert not builders.group_fits([B.BinnedBitSet(64)] * (builders.GROUP_MAX_MEMBERS + 1))


def test_thousands_of_default_sized_sets_stay_small(O, B):
    """bitset_builders.py:31-45 creates one BinnedBitSet(MAX) per sequence name; a scaffold-level assembly has thousands.
    The reference allocates 64 KiB bins on first touch; here the dense words grow lazily to the highest bit needed, so
    4000 default-sized sets (256 GB if they were allocated whole) with a few low ranges each must fit -- and every
    operation the scripts use must see the untouched rest as zeros: count_range over the whole set, next_set / next_clear
    runs, iand / ior between sets of different extents, reads far beyond anything set."""
    import ctypes

    from bxmi import _ffi

    def free_bytes():
        # through the C ABI, i.e. the HIP runtime libbxmi itself is bound to (a dlopen of "libamdhip64.so" by name brings a SECOND
        # runtime into the process when torch's bundled copy was loaded first -- it sees no device)
        free = ctypes.c_int64(0)
        _ffi.call("bxmi_mem_info", ctypes.byref(free), None)
        return free.value

    import gc

    rng = np.random.default_rng(12)
    gc.collect()  # handles of earlier tests release their device memory now, not in the middle of the measurement
    B.BinnedBitSet(100).count_range(0, 10)  # the runtime's own start-up allocations happen before the first reading
    free0 = free_bytes()
    n_sets = 4000
    sets, oracles = [], []
    for k in range(n_sets):
        b = B.BinnedBitSet()  # MAX bits
        sets.append(b)
        if k < 40:
            oracles.append(O.OracleBinnedBitSet())
    hi = [int(rng.integers(1_000, 3_000_000)) for _ in range(n_sets)]
    for k, b in enumerate(sets):
        st = rng.integers(0, hi[k], size=5)
        ln = rng.integers(1, 900, size=5)
        for s, c in zip(st.tolist(), ln.tolist()):
            b.set_range(s, c)
            if k < 40:
                oracles[k].set_range(s, c)
    tot = sum(b.count_range(0, b.size) for b in sets)  # bed_coverage.py:27-29 (flushes every set's queued ranges)
    assert tot > 0
    used = free0 - free_bytes()
    # ~4 MiB per set: the device allocator hands out 2 MiB blocks and a set owns a few buffers (words, tags, range staging);
    # 64 MiB each if the words were allocated whole
    assert used < 24 << 30, "lazy words: %d MiB in use for %d sets" % (used >> 20, n_sets)
    for k in range(40):
        b, o = sets[k], oracles[k]
        assert b.count_range(0, b.size) == o.count_range(0, o.size)
        assert b.count_range(hi[k] + 5000, 10_000_000) == 0 and b.count_range(B.MAX - 10, 10) == 0
        assert b.next_set(B.MAX - 5) == o.next_set(B.MAX - 5) == B.MAX
        assert b.next_clear(400_000_000) == o.next_clear(400_000_000) == 400_000_000
        pos, runs, oruns = 0, [], []
        while True:  # bed_intersect_basewise.py:32-38
            s = b.next_set(pos)
            if s == b.size:
                break
            e = b.next_clear(s)
            runs.append((s, e))
            pos = e
        pos = 0
        while True:
            s = o.next_set(pos)
            if s == o.size:
                break
            e = o.next_clear(s)
            oruns.append((s, e))
            pos = e
        assert runs == oruns, k
        assert b[hi[k] + 100_000] == 0 and b[B.MAX - 1] == 0
    # sets of different extents against each other
    for k in range(0, 40, 2):
        a, b, oa, ob = sets[k], sets[k + 1], oracles[k], oracles[k + 1]
        if k % 4 == 0:
            a.iand(b), oa.iand(ob)
        else:
            a.ior(b), oa.ior(ob)
        assert a.count_range(0, a.size) == oa.count_range(0, oa.size), k
        w = rng.integers(0, 3_500_000, size=50).astype(np.int32)
        n = rng.integers(0, 5000, size=50).astype(np.int32)
        assert [a.count_range(int(x), int(y)) for x, y in zip(w[:8], n[:8])] == [oa.count_range(int(x), int(y)) for x, y in zip(w[:8], n[:8])]
    # a bit far up makes that one set grow, and only that one
    sets[100].set_range(B.MAX - 1000, 500)
    assert sets[100].count_range(B.MAX - 2000, 2000) == 500 and sets[100].next_set(3_000_000 + 1000) == B.MAX - 1000
    inv = sets[101]
    before = inv.count_range(0, inv.size)
    inv.invert()
    assert inv.count_range(0, 100) in (100 - k for k in range(101)) and sets[102].count_range(0, sets[102].size) >= 0
    inv.invert()
    assert inv.count_range(0, inv.size) == before


def test_batch_errors_match_reference_messages():
    from bxmi.bitset import DeviceBitSet

    d = DeviceBitSet(1000, 10)
    with pytest.raises(IndexError, match=r"Count \(-3\) must be non-negative\."):
        d.set_ranges([5, 10, 20], [5, -3, 5])
    assert d.count_range(0, 1000) == 5  # the valid prefix was applied, like a per-line loop would have
    with pytest.raises(IndexError, match=r"End \(1001\) is larger than the size of this BinnedBitSet \(1000\)\."):
        d.count_ranges([0, 999], [10, 2])
    with pytest.raises(IndexError, match=r"1000 is larger than the size of this BitSet \(1000\)\."):
        d.count_ranges([1000], [0])
    with pytest.raises(IndexError, match=r"BitSet index \(-1\) must be non-negative\."):
        d.set_ranges([-1], [1])
    with pytest.raises(ValueError, match="BitSets must have the same size"):
        d.iand(DeviceBitSet(999, 10))


# --------------------------------------------------------- full-size properties --
def _runs_sha(rs, re):
    import hashlib

    return hashlib.sha256(np.concatenate([np.asarray(rs, dtype=np.int64), np.asarray(re, dtype=np.int64)]).tobytes()).hexdigest()


def test_cfg3_genome_default_max_sizes(golden_scale_doc):
    """BASELINE configs[2] the way the reference's builders size it when no `lens` is given (lib/bx/bitset_builders.py:31-45:
    BinnedBitSet() = MAX = 512 Mi bits per chromosome, bitset.pyx:196-203; the variant bench.py quotes as
    bitset.default_MAX_sizes).  EVERY chromosome against the real bx.bitset (tests/golden/scale.json
    "cfg3_bitsets_default_max", oracle/gen_golden.py --only bitsets_genome_default): bin_size, popcounts of A, B, A & B, A | B,
    the run list of A & B, and count_range on the INVERTED intersection (whole set, the chromosome's length, eight windows that
    start and end inside bins) -- through the per-set calls, then popcounts / iand / ior through BitSetGroup."""
    from bxmi.bitset import MAX, BitSetGroup, DeviceBitSet

    g = golden_scale_doc.get("cfg3_bitsets_default_max")
    assert g, "tests/golden/scale.json has no cfg3_bitsets_default_max point: the reference check must not vanish silently"
    gold = g["chroms"]
    assert list(gold) == list(synth.HG19_SIZES)
    ra = synth.genome_ranges(1_500_000, 301)
    rb = synth.genome_ranges(1_500_000, 302)
    chroms = list(synth.HG19_SIZES)
    A, B, A2 = [], [], []
    for chrom in chroms:
        want = gold[chrom]
        assert want["size"] == MAX
        a, b, a2 = DeviceBitSet(), DeviceBitSet(), DeviceBitSet()
        assert (a.size, a.bin_size) == (MAX, want["bin_size"])
        a.set_ranges(*ra[chrom]), b.set_ranges(*rb[chrom]), a2.set_ranges(*ra[chrom])
        A.append(a), B.append(b), A2.append(a2)
    # one launch per genome first (the sets are still the plain A and B)
    gA, gB, gA2 = BitSetGroup(A), BitSetGroup(B), BitSetGroup(A2)
    assert gA.popcounts().tolist() == [gold[c]["pop_a"] for c in chroms]
    assert gB.popcounts().tolist() == [gold[c]["pop_b"] for c in chroms]
    gA2.ior(gB)
    assert gA2.popcounts().tolist() == [gold[c]["pop_or"] for c in chroms]
    for a2 in A2:  # back to A for the per-set pass below
        a2.close()
    # per-set calls of the drop-in classes
    for chrom, a, b in zip(chroms, A, B):
        want = gold[chrom]
        assert (a.count_range(0, MAX), b.count_range(0, MAX)) == (want["pop_a"], want["pop_b"]), chrom
        a2 = DeviceBitSet()
        a2.set_ranges(*ra[chrom])
        a2.ior(b)
        assert a2.count_range(0, MAX) == want["pop_or"], chrom
        a2.iand(a)  # (A | B) & A == A
        assert a2.count_range(0, MAX) == want["pop_a"], chrom
        a2.iand(b)  # ... & B == A & B, through the per-set iand
        assert a2.count_range(0, MAX) == want["pop_and"], chrom
        rs, re = a2.runs()
        assert len(rs) == want["n_runs"] and _runs_sha(rs, re) == want["runs_sha256"], chrom
        a2.invert()
        assert a2.count_range(0, MAX) == want["inverted_and_count_all"], chrom
        assert a2.count_range(0, want["chrom_len"]) == want["inverted_and_count_chrom"], chrom
        w = np.array(want["inverted_and_windows"], dtype=np.int64)
        assert a2.count_ranges(w[:, 0], w[:, 1]).tolist() == w[:, 2].tolist(), chrom
        for s, n, c in want["inverted_and_windows"][:2]:
            assert a2.count_range(s, n) == c, (chrom, s, n)
        a2.close()
    # the group iand with counts, then the run lists and the inverted counts of what it left
    assert gA.iand(gB, want_counts=True).tolist() == [gold[c]["pop_and"] for c in chroms]
    assert gB.popcounts().tolist() == [gold[c]["pop_b"] for c in chroms]
    for chrom, a in zip(chroms, A):
        want = gold[chrom]
        rs, re = a.runs()
        assert len(rs) == want["n_runs"] and _runs_sha(rs, re) == want["runs_sha256"], chrom
        a.invert()
        assert a.count_range(0, want["chrom_len"]) == want["inverted_and_count_chrom"], chrom
    for d in A + B:
        d.close()


def test_cfg3_genome_scale_properties(O, golden_scale_doc):
    """BASELINE configs[2]: two hg19-sized (3.1 Gbp, 24 chromosomes) bitsets, iand + count_range.
    EVERY chromosome against the real bx.bitset.BinnedBitSet (tests/golden/scale.json "cfg3_bitsets", made by
    oracle/gen_golden.py --only bitsets_genome: popcounts of A, B, A & B, A | B and the sha256 of the run list of A & B),
    once through the per-chromosome calls of the drop-in classes and once through BitSetGroup (one launch per genome: the
    path bench.py quotes); chr21 and chrY also bit-for-bit against the oracle, every chromosome through identities."""
    from bxmi.bitset import BitSetGroup, DeviceBitSet

    g = golden_scale_doc.get("cfg3_bitsets")
    assert g, "tests/golden/scale.json has no cfg3_bitsets point: the reference check of configs[2] must not vanish silently"
    gold = g["chroms"]
    assert list(gold) == list(synth.HG19_SIZES)
    ra = synth.genome_ranges(1_500_000, 301)
    rb = synth.genome_ranges(1_500_000, 302)
    tot_a = tot_b = tot_and = tot_or = 0
    for chrom, size in synth.HG19_SIZES.items():
        want = gold[chrom]
        assert want["size"] == size
        a, b, a2 = DeviceBitSet(size), DeviceBitSet(size), DeviceBitSet(size)
        a.set_ranges(*ra[chrom]), b.set_ranges(*rb[chrom]), a2.set_ranges(*ra[chrom])
        ca, cb = a.count_range(0, size), b.count_range(0, size)
        assert (ca, cb) == (want["pop_a"], want["pop_b"]), chrom
        a2.ior(b)
        c_or = a2.count_range(0, size)
        c_and = a.and_count(b)  # fused iand + popcount
        assert (c_and, c_or) == (want["pop_and"], want["pop_or"]), chrom
        assert c_and == a.count_range(0, size)
        assert c_and + c_or == ca + cb  # inclusion-exclusion
        rs, re = a.runs()
        assert int((re - rs).sum()) == c_and and (rs[1:] > re[:-1]).all()
        assert len(rs) == want["n_runs"] and _runs_sha(rs, re) == want["runs_sha256"], chrom
        # per-range counts against the full count: consecutive windows tile the chromosome
        edges = np.linspace(0, size, 2001).astype(np.int64)
        win = a.count_ranges(edges[:-1], np.diff(edges))
        assert int(win.sum()) == c_and
        a.invert()
        assert a.count_range(0, size) <= size - c_and  # ALL_ONE first-bin arithmetic can only subtract
        a.invert()
        assert a.count_range(0, size) == c_and
        if chrom in ("chr21", "chrY"):
            oa, ob = O.OracleBinnedBitSet(size), O.OracleBinnedBitSet(size)
            oa.set_ranges(*ra[chrom]), ob.set_ranges(*rb[chrom])
            assert (ca, cb) == (oa.count_range(0, size), ob.count_range(0, size))
            oa.iand(ob)
            assert c_and == oa.count_range(0, size)
            ors, ore = oa.runs()
            assert np.array_equal(rs, ors) and np.array_equal(re, ore)
        tot_a, tot_b, tot_and, tot_or = tot_a + ca, tot_b + cb, tot_and + c_and, tot_or + c_or
        for d in (a, b, a2):
            d.close()
    assert tot_and + tot_or == tot_a + tot_b
    assert 0.30 < tot_a / 3_095_677_412 < 0.45  # ~38 % coverage per set
    # the same genome through ONE launch per operation (BitSetGroup): popcounts, iand with counts, ior, the run lists
    chroms = list(synth.HG19_SIZES)
    A = [DeviceBitSet(synth.HG19_SIZES[c]) for c in chroms]
    B = [DeviceBitSet(synth.HG19_SIZES[c]) for c in chroms]
    A2 = [DeviceBitSet(synth.HG19_SIZES[c]) for c in chroms]
    for c, a, b, a2 in zip(chroms, A, B, A2):
        a.set_ranges(*ra[c]), b.set_ranges(*rb[c]), a2.set_ranges(*ra[c])
    gA, gB, gA2 = BitSetGroup(A), BitSetGroup(B), BitSetGroup(A2)
    assert gA.popcounts().tolist() == [gold[c]["pop_a"] for c in chroms]
    assert gB.popcounts().tolist() == [gold[c]["pop_b"] for c in chroms]
    gA2.ior(gB)
    assert gA2.popcounts().tolist() == [gold[c]["pop_or"] for c in chroms]
    assert gA.iand(gB, want_counts=True).tolist() == [gold[c]["pop_and"] for c in chroms]
    assert gA.popcounts().tolist() == [gold[c]["pop_and"] for c in chroms]
    assert gB.popcounts().tolist() == [gold[c]["pop_b"] for c in chroms]  # the other operand is untouched
    for c, a in zip(chroms, A):
        rs, re = a.runs()
        assert len(rs) == gold[c]["n_runs"] and _runs_sha(rs, re) == gold[c]["runs_sha256"], c
    for grp in (gA, gB, gA2):
        grp.close()
    for d in A + B + A2:
        d.close()
