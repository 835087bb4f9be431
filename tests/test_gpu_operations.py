"""
GPU parity tests of the operations layer, `-m gpu`.

bxmi.operations + bxmi.genomic against tests/golden/operations.json, which oracle/gen_golden_ops.py
captured from the reference's bx.intervals.operations running on the reference's own readers:
every yielded object (type, fields, chrom/start/end/strand), what escapes as an exception, the warnings,
and the skip bookkeeping left on the primary reader -- all must be identical.
"""
import json
import os
import warnings

import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "operations.json")


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def mods():
    from bxmi import genomic, operations

    return genomic, operations


def _tell(genomic, item):
    if isinstance(item, genomic.Header):
        return ["header", str(item)]
    if isinstance(item, genomic.Comment):
        return ["comment", str(item)]
    if isinstance(item, list):
        return ["list", list(item)]
    return ["interval", [str(f) for f in item.fields], item.chrom, int(item.start), int(item.end), item.strand]


def _run(genomic, operations, case, inputs):
    make = {"nice": genomic.NiceReaderWrapper, "plain": genomic.GenomicIntervalReader}
    readers = [make[k](list(inputs[key])) for k, key in zip(case["readers"], case["inputs"])]
    op, params = case["op"], dict(case["params"])
    if op == "base_coverage":
        return readers, operations.base_coverage(readers[0]), None, []
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        try:
            if op in ("intersect", "subtract", "coverage"):
                it = getattr(operations, op)(readers, **params)
            elif op == "merge":
                it = operations.merge(readers[0])
            else:
                it = operations.complement(readers[0], params["lens"])
            out, err = [_tell(genomic, x) for x in it], None
        except Exception as e:
            out, err = None, [type(e).__name__, str(e)]
    return readers, out, err, [str(x.message) for x in w]


def test_operations_match_the_reference(golden, mods):
    genomic, operations = mods
    inputs = golden["inputs"]
    assert len(golden["cases"]) >= 40
    for case in golden["cases"]:
        readers, out, err, warns = _run(genomic, operations, case, inputs)
        name = case["name"]
        if case["op"] == "base_coverage":
            assert out == case["value"], name
            continue
        assert err == case["error"], (name, err, case["error"])
        if out is not None:
            assert len(out) == len(case["output"]), (name, len(out), len(case["output"]))
            for k, (got, want) in enumerate(zip(out, case["output"])):
                assert got == want, (name, k, got, want)
        assert warns == case["warnings"], (name, warns, case["warnings"])
        if case["primary"] is not None:
            p = readers[0]
            assert p.skipped == case["primary"]["skipped"], (name, p.skipped, case["primary"]["skipped"])
            assert [list(t) for t in p.skipped_lines] == case["primary"]["skipped_lines"], (name, p.skipped_lines)


def test_range_generators_match_the_batched_pieces(mods):
    """bits_set_in_range / bits_clear_in_range (per-call, on the drop-in BinnedBitSet) give the pieces the batched
    path computes from the run list -- including whether the scan runs off the end of the bitset."""
    import numpy as np

    from bx.bitset import BinnedBitSet

    _, operations = mods
    rng = np.random.default_rng(5)
    for size, nr in ((1000, 12), (4096, 40), (777, 0), (500, 3)):
        bits = BinnedBitSet(size)
        for _ in range(nr):
            s = int(rng.integers(0, size))
            bits.set_range(s, int(rng.integers(0, min(60, size - s) + 1)))
        if nr == 3:
            bits.set_range(size - 7, 7)  # a run reaching the very end
        table = operations._RunTable(bits)
        a = rng.integers(0, size, size=300)
        b = np.minimum(a + rng.integers(0, 200, size=300), size)
        a[:3], b[:3] = [0, size - 1, size // 2], [size, size, size]
        for fn, gen in ((table.set_pieces, operations.bits_set_in_range), (table.clear_pieces, operations.bits_clear_in_range)):
            off, ps, pe, off_end = fn(a.astype(np.int64), b.astype(np.int64))
            for i in range(len(a)):
                want, raised = [], False
                try:
                    for piece in gen(bits, int(a[i]), int(b[i])):
                        want.append(piece)
                except IndexError:
                    raised = True
                got = list(zip(ps[off[i]:off[i + 1]].tolist(), pe[off[i]:off[i + 1]].tolist()))
                assert got == want and bool(off_end[i]) == raised, (size, nr, fn.__name__, int(a[i]), int(b[i]), got, want, raised)


# ------------------------------------------------------------------ ClusterTree / find_clusters (rank 4) --
def test_clustertree_matches_reference_vectors(golden):
    """The drop-in bx.intervals.cluster.ClusterTree (device clustering through bxmi_ivl_clusters) against
    getregions()/getlines() of the reference's extension on 30 trees, and against the CPU oracle on a big one."""
    import numpy as np

    from bx.intervals.cluster import ClusterTree
    from oracle import oracle as O

    for c in golden["clusters"]:
        t = ClusterTree(c["max_dist"], c["min_intervals"])
        for a, b, i in c["triples"]:
            t.insert(a, b, i)
        assert [[a, b, ids] for a, b, ids in t.getregions()] == c["regions"], (c["max_dist"], c["min_intervals"], len(c["triples"]))
        assert t.getlines() == c["lines"]
        t.insert(10**8, 10**8 + 1, 7)  # inserting after a query invalidates the cached answer
        if c["min_intervals"] <= 1:
            assert t.getlines().count(7) == c["lines"].count(7) + 1
    t = ClusterTree(0, 0)
    with pytest.raises(ValueError, match="Interval start must be before end"):
        t.insert(4, 2, 0)
    with pytest.raises(OverflowError):
        t.insert(0, 2**31, 0)
    assert t.getregions() == [] and t.getlines() == []
    # 2M intervals, three distances: device result == CPU restatement
    rng = np.random.default_rng(4)
    n = 2_000_000
    s = rng.integers(0, 2_000_000_000, size=n).astype(np.int32)
    e = (s.astype(np.int64) + rng.integers(0, 400, size=n)).clip(max=2**31 - 1).astype(np.int32)
    ids = rng.integers(-(2**31), 2**31 - 1, size=n).astype(np.int32)
    from bxmi.intervals import IntervalIndex

    ix = IntervalIndex()
    ix.append(s, e)
    for md in (0, 300, 5000):
        cs, ce, off, mem = ix.clusters(md, ids)
        want = O.cluster_regions(s, e, ids, md, 0)
        assert len(cs) == len(want)
        assert cs.tolist() == [w[0] for w in want] and ce.tolist() == [w[1] for w in want]
        assert mem.tolist() == [i for w in want for i in w[2]]
        assert off.tolist() == np.concatenate([[0], np.cumsum([len(w[2]) for w in want])]).tolist()


def test_clustertree_distance_minus_one():
    """max_dist = -1 ("overlap by one base or more") on intervals of positive length: the engine against regions of the
    reference's own C (tests/golden/cluster_minus_one.json); refused with zero-length intervals and below -1."""
    import json
    import os

    from bx.intervals.cluster import ClusterTree

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = json.load(open(os.path.join(root, "tests", "golden", "cluster_minus_one.json")))
    for c in doc["cases"]:
        t = ClusterTree(-1, c["min_intervals"])
        for a, b, i in c["triples"]:
            t.insert(a, b, i)
        assert sorted([a, b, sorted(ids)] for a, b, ids in t.getregions()) == c["regions"], (c["min_intervals"], len(c["triples"]))
    t = ClusterTree(-1, 0)
    t.insert(0, 4, 0), t.insert(4, 4, 1)
    with pytest.raises(ValueError):
        t.getregions()
    t = ClusterTree(-2, 0)
    t.insert(0, 4, 0), t.insert(3, 9, 1)
    with pytest.raises(ValueError):
        t.getregions()
    from bxmi import _ffi
    from bxmi.intervals import IntervalIndex

    ix = IntervalIndex()
    ix.append([0, 4], [4, 4])
    with pytest.raises(_ffi.BxmiError):  # the C ABI says so itself
        ix.clusters(-1)
    with pytest.raises(_ffi.BxmiError):
        ix.clusters(-2)
    ix.close()


def test_find_clusters_matches_the_reference(golden, mods):
    genomic, operations = mods
    make = {"nice": genomic.NiceReaderWrapper, "plain": genomic.GenomicIntervalReader}
    for c in golden["find_clusters"]:
        reader = make[c["reader"]](list(golden["find_clusters_inputs"][c["input"]]))
        chroms, extra = operations.find_clusters(reader, **c["params"])
        assert list(chroms) == c["chrom_order"], c["name"]
        for chrom, want in c["chroms"].items():
            assert [[a, b, ids] for a, b, ids in chroms[chrom].getregions()] == want["regions"], (c["name"], chrom)
            assert chroms[chrom].getlines() == want["lines"], (c["name"], chrom)
        assert {str(k): _tell(genomic, v) for k, v in extra.items()} == c["extra"], c["name"]
        if c["primary"] is not None:
            assert reader.skipped == c["primary"]["skipped"]
            assert [list(t) for t in reader.skipped_lines] == c["primary"]["skipped_lines"]


def test_join_matches_the_reference(golden, mods):
    """operations.join against the reference's (captured with a seeded RNG).  The reference lists the matches of one
    left row in the pre-order of a randomly balanced tree, so rows sharing a left row are compared as a sorted group;
    headers, fill rows and the trailing never-matched right rows must be in the reference's exact order."""
    genomic, operations = mods
    g = golden["join"]

    def canon(items, leftlen=6):
        out, run, prev = [], [], None
        for it in items:
            key = tuple(it[1][:leftlen]) if it[0] == "list" else None
            if key is None or key != prev:
                out.extend(sorted(run))
                run = []
            if key is None:
                out.append(it)
            else:
                run.append(it)
            prev = key
        return out + sorted(run)

    assert len(g["cases"]) >= 6
    for c in g["cases"]:
        left = genomic.NiceReaderWrapper(list(g["inputs"][c["left"]]))
        right = genomic.GenomicIntervalReader(list(g["inputs"][c["right"]]))
        got = [_tell(genomic, x) for x in operations.join(left, right, **c["params"])]
        assert len(got) == len(c["output"]), (c["name"], len(got), len(c["output"]))
        assert canon(got) == canon(c["output"]), c["name"]
