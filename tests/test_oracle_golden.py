"""
Pins the CPU oracle (oracle/*.c) to the reference:
  * golden vectors produced by the real Cython reference (oracle/gen_golden.py),
  * the known answers of the reference's own tests
    (lib/bx/intervals/intersection_tests.py, lib/bx/bitset_tests.py,
     doctests intersection.pyx:335-376),
  * op-for-op equality with oracle/_ref (the reference's C compiled in place),
    when that library is present.
CPU only.
"""
import hashlib
import os

import numpy as np
import pytest

from bitset_replay import replay
from bxmi import synth
from oracle import oracle as O


# ---------------------------------------------------------------- intervals --
def build_tree(case):
    t = O.OracleIntervalTree()
    t.insert_many(case["starts"], case["ends"])
    return t


def test_find_matches_reference_vectors(golden_trees):
    nq = 0
    for case in golden_trees:
        t = build_tree(case)
        assert t.traverse().tolist() == case["order"], (case["mode"], case["n"])
        for (qs, qe), want in zip(case["queries"], case["hits"]):
            assert t.find(qs, qe).tolist() == want, (case["mode"], case["n"], qs, qe)
            nq += 1
    assert nq > 2000


def test_order_key_model(golden_trees):
    """In-order == sort by (start, end<=start first, -i / +i).

    SURVEY A.1 states the flag as end == start; the reversed-target vectors show it is
    end <= start (intersection.pyx:112-116 compares `end` with the node's *start*).
    This is the sort key the device index is built with."""
    for case in golden_trees:
        s, e = case["starts"], case["ends"]
        key = sorted(range(len(s)), key=lambda i: (s[i], 0 if e[i] <= s[i] else 1, -i if e[i] <= s[i] else i))
        assert key == case["order"], (case["mode"], case["n"])


def test_batch_apis_agree_with_find(golden_trees):
    for case in golden_trees[::5]:
        t = build_tree(case)
        q = np.array(case["queries"], dtype=np.int32)
        counts, total = t.count_batch(q[:, 0], q[:, 1])
        offs, hits = t.find_batch(q[:, 0], q[:, 1])
        assert counts.tolist() == [len(h) for h in case["hits"]]
        assert total == sum(len(h) for h in case["hits"]) == offs[-1]
        assert hits.tolist() == [x for h in case["hits"] for x in h]


def test_neighbours_match_reference_vectors(golden_trees):
    n = 0
    for case in golden_trees:
        if not case["neighbours"]:
            continue
        t = build_tree(case)
        for kind, pos, k, md, want in case["neighbours"]:
            got = t.left(pos, n=k, max_dist=md) if kind == "before" else t.right(pos, n=k, max_dist=md)
            assert got == want, (kind, pos, k, md)
            n += 1
    assert n > 300


def test_reference_known_answers_intervaltree():
    # intersection_tests.py:158-184 (IntervalTreeTest)
    t = O.OracleIntervalTree()
    n = 0
    for i in range(1, 1000, 80):
        for a, b in ((i, i + 10), (i + 20, i + 30), (i + 40, i + 50), (i + 60, i + 70)):
            t.insert(a, b)
            n += 1
    assert len(t.find(100, 200)) == 5
    assert len(t.traverse()) == n
    # intersection.pyx:345-361 doctests
    d = O.OracleIntervalTree()
    for a, b in ((0, 10), (3, 7), (3, 40), (13, 50)):
        d.insert(a, b)
    assert d.find(30, 50).tolist() == [2, 3]
    assert d.find(100, 200).tolist() == []
    assert d.left(10) == [1]  # before_interval(Interval(10, 20)) -> [Interval(3, 7)]
    assert d.left(5) == []
    assert d.left(11) == [0]  # upstream_of_interval(Interval(11, 12))
    assert d.right(12) == [3]  # ... strand="-"
    assert d.right(2, n=3) == [1, 2, 3]
    e = O.OracleIntervalTree()
    e.insert(0, 10)
    e.insert(3, 7)
    assert e.find(2, 5).tolist() == [0, 1]
    # empty tree (intersection_tests.py:186-195)
    assert O.OracleIntervalTree().find(100, 300).tolist() == []


def test_reference_known_answers_neighbours():
    # intersection_tests.py:17-54 (NeighborTestCase)
    t = O.OracleIntervalTree()
    t.insert(50, 59)
    for i in range(0, 110, 10):
        if i != 50:
            t.insert(i, i + 9)
    se = lambda idx: [(t.starts[i], t.ends[i]) for i in idx]
    assert se(t.left(60, n=2)) == [(50, 59), (40, 49)]
    for i in range(10, 100, 10):
        assert se(t.left(i, max_dist=10, n=1))[0][1] == i - 1
    assert len(t.left(60, n=200)) == 6
    for i in range(10, 100, 10):
        assert se(t.right(i + 1, n=1))[0][0] == i + 10
    for i in range(0, 100, 10):
        assert se(t.right(i - 1, max_dist=10, n=1))[0][0] == i


def test_reference_known_answers_lotsa():
    # intersection_tests.py:104-141 (LotsaTestCase): 100k zero-length + 600 x (0,1)
    t = O.OracleIntervalTree()
    t.insert(1, 2)
    s = np.arange(0, 1000000, 10, dtype=np.int32)
    t.insert_many(s, s)
    t.insert_many(np.zeros(600, np.int32), np.ones(600, np.int32))
    assert len(t.right(1, n=33)) == 33
    assert len(t.left(1, n=33)) == 1
    assert len(t.right(1, n=9999)) == 250
    assert len(t.right(1, n=9999, max_dist=99999)) == 9999
    assert len(t.right(1, max_dist=0, n=10)) == 0
    for n, d in enumerate(range(10, 1000, 10)):
        assert len(t.right(1, max_dist=d, n=10000)) == n + 1


def _scale_counts(pt):
    (ts, te), _ = synth.cfg2(pt["n_targets"], 1)
    qs, qe = synth.uniform_intervals(pt["n_queries_total"], 202)
    qs, qe = qs[:: pt["stride"]].copy(), qe[:: pt["stride"]].copy()
    t = O.OracleIntervalTree()
    t.insert_many_arrays(ts, te)
    counts, total = t.count_batch(qs, qe)
    return counts, total


def test_scale_1M_hash(golden_scale):
    pt = golden_scale["1M x 200k"]
    counts, total = _scale_counts(pt)
    assert total == pt["total"]
    assert counts[:16].tolist() == pt["first16"]
    assert hashlib.sha256(counts.tobytes()).hexdigest() == pt["counts_sha256"]


@pytest.mark.slow
def test_scale_10M_hash(golden_scale):
    key = "10M x 1M (cfg2 subsample)"
    assert key in golden_scale, "tests/golden/scale.json lost its %r point" % key
    pt = golden_scale[key]
    counts, total = _scale_counts(pt)
    assert total == pt["total"]
    assert hashlib.sha256(counts.tobytes()).hexdigest() == pt["counts_sha256"]


# ------------------------------------------------------------------ bitsets --
def test_binnedbitset_matches_reference_vectors(golden_bitsets):
    for case in golden_bitsets["cases"]:
        replay(case, O.OracleBinnedBitSet)


def test_binnedbitset_genome_scale_vectors(golden_scale_doc):
    """configs[2] at scale: the restatement against the real bx.bitset on the three smallest chromosomes of the cfg 3
    genome (tests/golden/scale.json "cfg3_bitsets": popcounts of A, B, A & B, A | B and the run list of A & B)."""
    g = golden_scale_doc.get("cfg3_bitsets")
    assert g, "tests/golden/scale.json has no cfg3_bitsets point"
    ra, rb = synth.genome_ranges(1_500_000, 301), synth.genome_ranges(1_500_000, 302)
    for chrom in ("chr21", "chr22", "chrY"):
        want, size = g["chroms"][chrom], synth.HG19_SIZES[chrom]
        a, b, a2 = O.OracleBinnedBitSet(size), O.OracleBinnedBitSet(size), O.OracleBinnedBitSet(size)
        a.set_ranges(*ra[chrom]), b.set_ranges(*rb[chrom]), a2.set_ranges(*ra[chrom])
        assert (a.count_range(0, size), b.count_range(0, size)) == (want["pop_a"], want["pop_b"])
        a2.ior(b)
        assert a2.count_range(0, size) == want["pop_or"]
        a.iand(b)
        assert a.count_range(0, size) == want["pop_and"]
        rs, re = a.runs()
        assert len(rs) == want["n_runs"]
        runs = np.concatenate([np.asarray(rs, dtype=np.int64), np.asarray(re, dtype=np.int64)])
        assert hashlib.sha256(runs.tobytes()).hexdigest() == want["runs_sha256"]


def test_binnedbitset_genome_default_max_vectors(golden_scale_doc):
    """The same genome with every chromosome a BinnedBitSet() of the default MAX = 512 Mi bits (what
    lib/bx/bitset_builders.py:31-45 builds when no `lens` is given; a different float32 bin_size, binBits.c:36): the
    restatement against the real bx.bitset on chr21 and chrY (tests/golden/scale.json "cfg3_bitsets_default_max", made by
    oracle/gen_golden.py --only bitsets_genome_default), including count_range on the INVERTED intersection."""
    g = golden_scale_doc.get("cfg3_bitsets_default_max")
    assert g, "tests/golden/scale.json has no cfg3_bitsets_default_max point"
    ra, rb = synth.genome_ranges(1_500_000, 301), synth.genome_ranges(1_500_000, 302)
    for chrom in ("chr21", "chrY"):
        want = g["chroms"][chrom]
        size = want["size"]
        assert size == 512 * 1024 * 1024 and want["chrom_len"] == synth.HG19_SIZES[chrom]
        a, b, a2 = O.OracleBinnedBitSet(size), O.OracleBinnedBitSet(size), O.OracleBinnedBitSet(size)
        assert a.bin_size == want["bin_size"]
        a.set_ranges(*ra[chrom]), b.set_ranges(*rb[chrom]), a2.set_ranges(*ra[chrom])
        assert (a.count_range(0, size), b.count_range(0, size)) == (want["pop_a"], want["pop_b"])
        a2.ior(b)
        assert a2.count_range(0, size) == want["pop_or"]
        a.iand(b)
        assert a.count_range(0, size) == want["pop_and"]
        rs, re = a.runs()
        assert len(rs) == want["n_runs"]
        runs = np.concatenate([np.asarray(rs, dtype=np.int64), np.asarray(re, dtype=np.int64)])
        assert hashlib.sha256(runs.tobytes()).hexdigest() == want["runs_sha256"]
        a.invert()
        assert a.count_range(0, size) == want["inverted_and_count_all"]
        assert a.count_range(0, want["chrom_len"]) == want["inverted_and_count_chrom"]
        for s, n, c in want["inverted_and_windows"]:
            assert a.count_range(s, n) == c, (chrom, s, n)


def test_binnedbitset_big_sizes(golden_bitsets):
    for case in golden_bitsets["big"]:
        replay(case, O.OracleBinnedBitSet, check_final=False)


def test_binnedbitset_ctor_limits(golden_bitsets):
    assert O.MAX == golden_bitsets["MAX"]
    for size, want in golden_bitsets["ctor"]:
        try:
            got = ["ok", O.OracleBinnedBitSet(size).size]
        except ValueError as ex:
            got = ["ValueError", str(ex)]
        assert got == want


def test_reference_known_answers_bitset():
    # lib/bx/bitset_tests.py:51-108 with the same (size, granularity = size % 11) as :117-119
    def new():
        return O.OracleBinnedBitSet(100, 100 % 11)

    b = new()
    for s, e in ((11, 14), (20, 75), (90, 100)):
        b.set_range(s, e - s)
    assert [b.count_range(0, 0), b.count_range(0, 20), b.count_range(25, 25), b.count_range(80, 20), b.count_range(0, 100)] == [0, 3, 25, 10, 68]
    assert [b.next_set(0), b.next_set(13), b.next_set(15)] == [11, 13, 20]
    assert [b.next_clear(0), b.next_clear(11), b.next_clear(20), b.next_clear(92)] == [0, 14, 75, 100]
    with pytest.raises(IndexError):
        b.set(-5)
    with pytest.raises(IndexError):
        b.set(110)
    with pytest.raises(ValueError):
        O.OracleBinnedBitSet(4000000000, 4000000000 % 11)
    x, y = new(), new()
    x.set_range(20, 40)
    y.set_range(50, 25)
    x.iand(y)
    assert x.unpack().tolist() == [1 if 50 <= i < 60 else 0 for i in range(100)]
    x, y = new(), new()
    x.set_range(20, 40)
    y.set_range(50, 25)
    x.ior(y)
    assert x.unpack().tolist() == [1 if 20 <= i < 75 else 0 for i in range(100)]
    z = new()
    z.set_range(20, 40)
    z.invert()
    assert z.unpack().tolist() == [0 if 20 <= i < 60 else 1 for i in range(100)]


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_restatement_equals_compiled_reference_c():
    """Random differential run: our binbits.c vs the reference's C files compiled in place."""
    rng = np.random.default_rng(77)
    for size, gran in ((100, 1), (997, 7), (5000, 64), (70000, 1024), (1 << 20, 1024), (12345, 10)):
        mine = [O.OracleBinnedBitSet(size, gran) for _ in range(2)]
        ref = [O.RefBinnedBitSet(size, gran) for _ in range(2)]
        assert (mine[0].bin_size, mine[0].nbins) == (ref[0].bin_size, ref[0].nbins)
        for step in range(400):
            w = int(rng.integers(0, 2))
            op = int(rng.integers(0, 10))
            s = int(rng.integers(0, size))
            n = int(rng.integers(0, min(size - s, max(2, size // 4)) + 1))
            if op <= 2:
                mine[w].set_range(s, n), ref[w].set_range(s, n)
            elif op == 3:
                assert mine[w].count_range(s, n) == ref[w].count_range(s, n), (size, gran, step)
            elif op == 4:
                assert mine[w].next_set(s) == ref[w].next_set(s), (size, gran, step)
                assert mine[w].next_clear(s) == ref[w].next_clear(s), (size, gran, step)
            elif op == 5:
                mine[w].set(s), ref[w].set(s)
                mine[1 - w].clear(s), ref[1 - w].clear(s)
            elif op == 6 and rng.random() < 0.4:
                mine[w].invert(), ref[w].invert()
            elif op == 7:
                mine[w].iand(mine[1 - w]), ref[w].iand(ref[1 - w])
            elif op == 8:
                mine[w].ior(mine[1 - w]), ref[w].ior(ref[1 - w])
            else:
                assert mine[w][s] == ref[w][s]
        for w in range(2):
            assert mine[w].count_range(0, size) == ref[w].count_range(0, size)
            assert all(mine[w][p] == ref[w][p] for p in range(0, size, max(1, size // 997)))


# ------------------------------------------------------------------ ClusterTree --
def _cluster_golden():
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "operations.json")) as f:
        return json.load(f)["clusters"]


def test_cluster_restatement_matches_reference_vectors():
    """oracle/cluster.c against getregions()/getlines() of the reference's extension (30 trees: the inputs of the
    reference's own cluster_tests.py plus seeded random ones; oracle/gen_golden_ops.py)."""
    cases = _cluster_golden()
    assert len(cases) >= 30
    for c in cases:
        t = c["triples"]
        got = O.cluster_regions([x[0] for x in t], [x[1] for x in t], [x[2] for x in t], c["max_dist"], c["min_intervals"])
        assert [[a, b, ids] for a, b, ids in got] == c["regions"], (c["max_dist"], c["min_intervals"], len(t))
        assert [i for _, _, ids in got for i in ids] == c["lines"]


def test_reference_known_answers_cluster():
    """cluster_tests.py:15-61 as literals."""
    def regions(pairs, md=0, mn=0):
        return O.cluster_regions([p[0] for p in pairs], [p[1] for p in pairs], list(range(len(pairs))), md, mn)

    assert regions([(3, 4), (6, 7), (9, 10), (1, 2), (3, 8)]) == [(1, 2, [3]), (3, 8, [0, 1, 4]), (9, 10, [2])]
    assert regions([(1, 4), (4, 5)]) == [(1, 5, [0, 1])]
    assert regions([(1, 2), (4, 5), (2, 4)]) == [(1, 5, [0, 1, 2])]
    assert regions([(1, 2), (8, 9), (3, 4), (5, 6), (7, 8), (1, 10)]) == [(1, 10, [0, 1, 2, 3, 4, 5])]
    assert regions([(1, 1), (1, 2), (3, 4), (3, 4), (1, 4)]) == [(1, 4, [0, 1, 2, 3, 4])]
    assert regions([(3, 4), (6, 7), (9, 10), (1, 2), (3, 8)], mn=2) == [(3, 8, [0, 1, 4])]
    assert regions([(3, 4), (6, 7), (9, 10), (1, 2), (3, 8)], md=1) == [(1, 10, [0, 1, 2, 3, 4])]
    upto = 100000
    pairs = [(2 * i + 1, 2 * i + 2) for i in range(upto)] + [(0, upto * 3)]
    assert regions(pairs) == [(0, upto * 3, list(range(upto + 1)))]


@pytest.mark.skipif(not O.have_ref_cluster(), reason="oracle/_ref/libcluster_ref.so is built only where /root/reference exists")
def test_cluster_restatement_equals_compiled_reference_c():
    """oracle/cluster.c vs the reference's own src/cluster.c compiled in place, inserted in random orders."""
    rng = np.random.default_rng(9)
    for trial in range(400):
        n = int(rng.integers(1, 80))
        md, mn = int(rng.integers(0, 25)), int(rng.integers(0, 4))
        span = int(rng.choice([40, 300, 10**6]))
        s = rng.integers(-span, span, size=n)
        e = s + rng.integers(0, 30, size=n)
        ids = rng.integers(-50, 5000, size=n)
        order = rng.permutation(n)
        want = O.ref_cluster_regions([(int(s[i]), int(e[i]), int(ids[i])) for i in order], md, mn)
        got = O.cluster_regions(s, e, ids, md, mn)
        assert got == want, (trial, md, mn, n)


def test_restatement_is_clean_under_asan_and_ubsan():
    """The checker itself is checked: the golden-vector tests of this file run once more in a child python whose
    liboracle is built with -fsanitize=address,undefined (oracle/Makefile `san`).  Any out-of-bounds access, use
    after free, signed overflow or misaligned load in oracle/*.c aborts the child."""
    import os
    import subprocess
    import sys

    if os.environ.get("ORACLE_SANITIZED") == "1":
        pytest.skip("already inside the sanitized child")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    env = dict(os.environ, ORACLE_SANITIZED="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    picks = "vectors or known_answers or ctor_limits or big_sizes or order_key or batch_apis"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider", "-k", picks],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_cluster_distance_minus_one_vectors():
    """max_dist = -1 on intervals of positive length is the one negative distance the reference answers reproducibly
    (oracle/cluster_negative_distance.py): the restatement against regions of the reference's own C (tests/golden/
    cluster_minus_one.json), and its refusals."""
    import json

    from oracle import oracle as O

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = json.load(open(os.path.join(root, "tests", "golden", "cluster_minus_one.json")))
    assert len(doc["cases"]) >= 30
    for c in doc["cases"]:
        s, e, ids = (np.array(x, dtype=np.int32) for x in zip(*c["triples"]))
        got = sorted([a, b, sorted(m)] for a, b, m in O.cluster_regions(s, e, ids, -1, c["min_intervals"]))
        assert got == c["regions"], (c["min_intervals"], len(c["triples"]))
    with pytest.raises(ValueError):
        O.cluster_regions([0, 5], [3, 5], None, -1)  # a zero-length interval
    with pytest.raises(ValueError):
        O.cluster_regions([0, 5], [3, 9], None, -2)
    if O.have_ref_cluster():  # the restatement against the reference's C, random insertion orders
        rng = np.random.default_rng(77)
        for _ in range(40):
            n = int(rng.integers(2, 150))
            s = rng.integers(0, 400, size=n)
            e = s + rng.integers(1, 40, size=n)
            tri = [(int(a), int(b), i) for i, (a, b) in enumerate(zip(s, e))]
            ref = sorted([a, b, sorted(m)] for a, b, m in O.ref_cluster_regions([tri[i] for i in rng.permutation(n)], -1, 0))
            assert sorted([a, b, sorted(m)] for a, b, m in O.cluster_regions(s, e, None, -1, 0)) == ref


def test_negative_cluster_distance_has_no_answer_to_reproduce():
    """ClusterTree(max_dist < -1), and -1 with zero-length intervals, are refused by the engine (BXMI_EINVAL).  The committed
    experiment on the reference's own src/cluster.c (oracle/cluster_negative_distance.py, its table in tests/golden/
    cluster_negative_distance.txt: -1 ... -8 and -20, 150 interval sets of three kinds) shows why: there the regions depend on
    the insertion order AND on the rand() priorities of the treap, while every non-negative distance -- and -1 on intervals
    of positive length -- gives one answer, the sweep's."""
    import subprocess
    import sys

    from oracle import oracle as O

    if not O.have_ref_cluster():
        pytest.skip("oracle/_ref/libcluster_ref.so not built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "oracle", "cluster_negative_distance.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "negative distances depend on order / priorities: True" in r.stdout
    row = [l for l in r.stdout.splitlines() if l.split()[:1] == ["-1"]][0]
    assert "| 0 of 50 / 0 of 50 / 50 of 50" in row, row  # positive lengths: one answer; with zero-length intervals: none
    assert all("0 of 50 / 0 of 50 / 0 of 50" not in l for l in r.stdout.splitlines() if l.split()[:1] in (["-2"], ["-3"], ["-4"]))
