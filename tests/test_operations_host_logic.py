"""
CPU tests (-m "not gpu") of the HOST logic of bxmi.operations / bxmi.genomic: the device engine is replaced by the
CPU oracle (test infrastructure, oracle/) behind the two seams the operations use -- `bx.bitset.BinnedBitSet` and
`bxmi.operations.IntervalIndex` -- and everything is compared with the vectors captured from the reference
(tests/golden/operations.json).  This pins the readers, the batching arithmetic (piece clipping, gap building,
run-off-the-end rule, skip replay) and join/find_clusters without a GPU; tests/test_gpu_operations.py runs the same
comparison on the real engine.
"""
import json
import os
import warnings

import numpy as np
import pytest

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "operations.json")


class _OracleBits(O.OracleBinnedBitSet):
    """OracleBinnedBitSet with the extra surface of the drop-in class that bxmi.operations touches."""

    @property
    def _d(self):
        return self

    def check_range_count(self, start, count):
        self._check_range_count(start, count)

    def runs(self, start=0):
        """The engine's contract (bxmi.bitset.DeviceBitSet.runs): the runs that end after `start`, the first one clipped to it;
        start == size gives none, anything else out of range raises like an index."""
        if start == self.size:
            return np.empty(0, np.int32), np.empty(0, np.int32)
        self._check_index(start)
        rs, re = O.OracleBinnedBitSet.runs(self)
        keep = re > start
        rs, re = rs[keep].copy(), re[keep].copy()
        if len(rs) and rs[0] < start:
            rs[0] = start
        return rs, re


class _OracleIndex:
    """IntervalIndex look-alike on the oracle treap (find in tree order -> insertion indices)."""

    def __init__(self):
        self._s, self._e = [], []

    def append(self, starts, ends):
        self._s.extend(np.asarray(starts).tolist())
        self._e.extend(np.asarray(ends).tolist())

    def seal(self):
        pass

    def close(self):
        pass

    def _tree(self):
        t = O.OracleIntervalTree()
        t.insert_many_arrays(np.array(self._s, dtype=np.int32), np.array(self._e, dtype=np.int32))
        return t

    def find(self, qs, qe):
        return self._tree().find_batch(np.asarray(qs, dtype=np.int32), np.asarray(qe, dtype=np.int32))

    def clusters(self, max_dist, ids=None):
        regs = O.cluster_regions(self._s, self._e, ids, max_dist, 0)
        off = np.concatenate([[0], np.cumsum([len(r[2]) for r in regs])]).astype(np.int64)
        return (np.array([r[0] for r in regs], np.int32), np.array([r[1] for r in regs], np.int32), off,
                np.array([i for r in regs for i in r[2]], np.int32))


@pytest.fixture()
def mods(monkeypatch):
    import bx.bitset
    import bx.intervals.cluster as cluster_mod
    from bxmi import genomic, operations

    monkeypatch.setattr(bx.bitset, "BinnedBitSet", _OracleBits)
    monkeypatch.setattr(operations, "IntervalIndex", _OracleIndex)
    monkeypatch.setattr(cluster_mod, "IntervalIndex", _OracleIndex)
    return genomic, operations


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


def _tell(genomic, item):
    if isinstance(item, genomic.Header):
        return ["header", str(item)]
    if isinstance(item, genomic.Comment):
        return ["comment", str(item)]
    if isinstance(item, list):
        return ["list", list(item)]
    return ["interval", [str(f) for f in item.fields], item.chrom, int(item.start), int(item.end), item.strand]


def test_operations_host_logic_matches_the_reference(golden, mods):
    genomic, operations = mods
    make = {"nice": genomic.NiceReaderWrapper, "plain": genomic.GenomicIntervalReader}
    for case in golden["cases"]:
        readers = [make[k](list(golden["inputs"][key])) for k, key in zip(case["readers"], case["inputs"])]
        op, params, name = case["op"], dict(case["params"]), case["name"]
        if op == "base_coverage":
            assert operations.base_coverage(readers[0]) == case["value"], name
            continue
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
        assert err == case["error"], (name, err)
        assert out == case["output"], name
        assert [str(x.message) for x in w] == case["warnings"], name
        if case["primary"] is not None:
            assert readers[0].skipped == case["primary"]["skipped"], name
            assert [list(t) for t in readers[0].skipped_lines] == case["primary"]["skipped_lines"], name


def test_join_and_find_clusters_host_logic(golden, mods):
    genomic, operations = mods
    g = golden["join"]

    def canon(items, leftlen=6):
        out, run, prev = [], [], None
        for it in items:
            key = tuple(it[1][:leftlen]) if it[0] == "list" else None
            if key is None or key != prev:
                out.extend(sorted(run))
                run = []
            (out if key is None else run).append(it)
            prev = key
        return out + sorted(run)

    for c in g["cases"]:
        left = genomic.NiceReaderWrapper(list(g["inputs"][c["left"]]))
        right = genomic.GenomicIntervalReader(list(g["inputs"][c["right"]]))
        got = [_tell(genomic, x) for x in operations.join(left, right, **c["params"])]
        assert canon(got) == canon(c["output"]), c["name"]
    make = {"nice": genomic.NiceReaderWrapper, "plain": genomic.GenomicIntervalReader}
    for c in golden["find_clusters"]:
        reader = make[c["reader"]](list(golden["find_clusters_inputs"][c["input"]]))
        chroms, extra = operations.find_clusters(reader, **c["params"])
        assert list(chroms) == c["chrom_order"]
        for chrom, want in c["chroms"].items():
            assert [[a, b, ids] for a, b, ids in chroms[chrom].getregions()] == want["regions"], (c["name"], chrom)
            assert chroms[chrom].getlines() == want["lines"]
        assert {str(k): _tell(genomic, v) for k, v in extra.items()} == c["extra"]


def test_readers_iterate_like_the_reference(golden):
    """bxmi.genomic readers against plain iteration of the reference's (items, escaping ParseError, skip log, final
    line number, header) over every input of the suite plus space-separated, CRLF and header-less files, in six
    reader configurations.  No engine involved."""
    from bxmi import genomic

    make = {"nice": genomic.NiceReaderWrapper, "plain": genomic.GenomicIntervalReader}
    g = golden["readers"]
    inputs = dict(golden["inputs"])
    inputs.update(g["inputs"])
    assert len(g["cases"]) >= 30
    for c in g["cases"]:
        r = make[c["reader"]](list(inputs[c["input"]]), **c["kwargs"])
        items, err = [], None
        try:
            for x in r:
                items.append(_tell(genomic, x))
        except Exception as e:
            err = [type(e).__name__, str(e)]
        tag = (c["input"], c["reader"], c["kwargs"])
        assert err == c["error"], (tag, err, c["error"])
        assert items == c["items"], tag
        assert r.linenum == c["linenum"], tag
        assert (str(r.header) if r.header is not None else None) == c["header"], tag
        if c["skips"] is not None:
            assert r.skipped == c["skips"]["skipped"], tag
            assert [list(t) for t in r.skipped_lines] == c["skips"]["skipped_lines"], tag


def test_quicksect_tree_host_logic(monkeypatch):
    """bx.intervals.operations.quicksect.IntervalTree on the oracle index: the hit sets, the in-order traverse and the
    per-chromosome objects of the three trees captured from the reference (tests/golden/builders_quicksect.json)."""
    import bx.intervals.intersection as inter
    from bx.intervals.operations.quicksect import IntervalTree

    class _Index(_OracleIndex):
        def find_one_list(self, qs, qe):
            off, hits = self.find([qs], [qe])
            return hits.tolist()

        def order(self):
            return np.array(sorted(range(len(self._s)), key=lambda j: (self._s[j], -j)), dtype=np.int32)

    monkeypatch.setattr(inter, "IntervalIndex", _Index)
    with open(os.path.join(os.path.dirname(GOLDEN), "builders_quicksect.json")) as f:
        doc = json.load(f)

    class Row:
        def __init__(self, chrom, start, end):
            self.chrom, self.start, self.end = chrom, start, end

    for case in doc["quicksect"]:
        tree = IntervalTree()
        for i, (c, s, e) in enumerate(case["rows"]):
            tree.insert(Row(c, s, e), linenum=i, other="row%d" % i)
        assert list(tree.chroms) == case["chrom_order"]
        for (c, s, e), want in zip(case["queries"], case["found"]):
            got = []
            tree.intersect(Row(c, s, e), lambda node: got.append([node.linenum, node.start, node.end, node.other]))
            assert sorted(got) == want, (case["name"], c, s, e)
        order = []
        tree.traverse(lambda node: order.append(node.linenum))
        assert order == case["traverse"], case["name"]


def test_small_builders_host_logic(monkeypatch):
    """bxmi.builders' list / proximity / by-chromosome builders on the oracle bitset: all 27 reference cases of
    tests/golden/builders_quicksect.json (resulting runs per chromosome, dict order, or the exception)."""
    import bxmi.builders as builders

    monkeypatch.setattr(builders, "BinnedBitSet", _OracleBits)
    with open(os.path.join(os.path.dirname(GOLDEN), "builders_quicksect.json")) as f:
        doc = json.load(f)

    def runs(bits):
        s, e = bits.runs()
        return [[int(a), int(b)] for a, b in zip(s, e)]

    def observe(fn):
        try:
            got = fn()
        except Exception as e:  # noqa: BLE001 -- the reference's own failure is the expectation
            return dict(error=[type(e).__name__, str(e)])
        if isinstance(got, dict):
            return dict(order=list(got), runs={c: runs(b) for c, b in got.items()})
        return dict(runs=runs(got))

    assert len(doc["builders"]) >= 25
    for case in doc["builders"]:
        a = case["args"]
        if case["fn"] == "from_list":
            got = observe(lambda: builders.binned_bitsets_from_list(a["rows"]))
        elif case["fn"] == "proximity":
            got = observe(lambda: builders.binned_bitsets_proximity(iter(a["lines"]), **a["kw"]))
        else:
            got = observe(lambda: builders.binned_bitsets_by_chrom(iter(a["lines"]), a["chrom"], **a["kw"]))
        assert got == case["want"], case["name"]


def test_concat_matches_reference_vectors():
    """bx.intervals.operations.concat (lib/bx/intervals/operations/concat.py:20-61) on this repo's readers: two inputs with
    different column orders, comments, headers, over-long rows, a row that cannot be parsed -- every yielded row as text and the
    escaping exception, as captured from the reference (oracle/gen_golden_extra.py --utils).  No engine involved."""
    import io

    from bx.intervals.io import GenomicIntervalReader
    from bx.intervals.operations.concat import concat

    with open(os.path.join(os.path.dirname(GOLDEN), "bitset_utils_concat.json")) as f:
        cases = json.load(f)["concat"]
    assert len(cases) == 12
    for c in cases:
        r1 = GenomicIntervalReader(io.StringIO(c["f1"]), chrom_col=0, start_col=1, end_col=2, strand_col=5)
        r2 = GenomicIntervalReader(io.StringIO(c["f2"]), chrom_col=1, start_col=2, end_col=3, strand_col=0)
        out, err = [], None
        try:
            for x in concat([r1, r2], comments=c["comments"], header=c["header"], sameformat=c["sameformat"]):
                out.append(str(x))
        except Exception as e:
            err = [type(e).__name__, str(e)]
        assert out == c["result"] and err == c["error"], (c["sameformat"], c["comments"], c["header"], out, c["result"], err, c["error"])


def test_bitset_utils_host_logic(monkeypatch):
    """bx.bitset_utils' walks (complement inside the list's range, runs reported whole by bitset_interval_intersect, the run
    that reaches the end of the set) on the oracle bitset behind the drop-in's seam, against the reference's vectors; the same
    comparison runs on the engine in tests/test_gpu_builders_quicksect.py."""
    import bx.bitset_utils as bu

    monkeypatch.setattr(bu, "BinnedBitSet", _OracleBits)
    with open(os.path.join(os.path.dirname(GOLDEN), "bitset_utils_concat.json")) as f:
        cases = json.load(f)["utils"]

    def call(fn, *a):
        try:
            return dict(result=[list(x) for x in fn(*a)])
        except Exception as e:
            return dict(error=[type(e).__name__, str(e)])

    assert len(cases) == 25
    for k, c in enumerate(cases):
        a, b = [tuple(x) for x in c["a"]], [tuple(x) for x in c["b"]]
        assert call(bu.bitset_intersect, a, b) == c["intersect"], (k, "intersect")
        assert call(bu.bitset_subtract, a, b) == c["subtract"], (k, "subtract")
        assert call(bu.bitset_union, a) == c["union"], (k, "union")
        assert call(bu.bitset_complement, b) == c["complement"], (k, "complement")
        lo, hi = c["window"]
        assert call(lambda: bu.bitset_interval_intersect(bu.list2bits(b), lo, hi)) == c["interval_intersect"], (k, "interval_intersect")


def test_bitset_interval_intersect_windows_past_the_end_and_duck_types():
    """bitset_interval_intersect on SMALL sets (the reference vectors are all MAX-sized): a window whose end lies beyond the set
    raises what the reference's walk raises (next_set answers `size` < iend, next_clear(size) is an IndexError,
    lib/bx/bitset_utils.py:73-85), negative / out-of-range starts raise, and an object that only has next_set / next_clear
    (the flat BitSet) is walked the reference's way.  The run-list walk and the reference's loop agree on every case."""
    import bx.bitset_utils as bu

    class Plain:  # the reference's duck type: no runs()
        def __init__(self, b):
            self._b, self.size = b, b.size
            self.next_set, self.next_clear = b.next_set, b.next_clear

    def reference_walk(bits, istart, iend):  # lib/bx/bitset_utils.py:73-85, as written there
        rval, end = [], istart
        while True:
            start = bits.next_set(end)
            if start >= iend:
                break
            end = bits.next_clear(start)
            if start != end:
                rval.append((start, end))
            if end >= iend:
                break
        return rval

    def call(fn, *a):
        try:
            return ("ok", fn(*a))
        except IndexError as e:
            return ("IndexError", str(e))

    rng = np.random.default_rng(77)
    seen = set()
    for _ in range(300):
        size = int(rng.integers(50, 400))
        b = _OracleBits(size, int(rng.choice([1, 3, 16, 1024])))
        for _ in range(int(rng.integers(0, 6))):
            s = int(rng.integers(0, size))
            b.set_range(s, int(rng.integers(0, size - s + 1)))
        lo = int(rng.integers(-3, size + 3))
        hi = int(rng.integers(lo, size + 40))
        want = call(reference_walk, b, lo, hi)
        assert call(bu.bitset_interval_intersect, b, lo, hi) == want, (size, lo, hi)
        assert call(bu.bitset_interval_intersect, Plain(b), lo, hi) == want, (size, lo, hi, "duck type")
        seen.add((want[0], hi > size, lo < 0))
        assert call(bu.bits2list, Plain(b)) == call(bu.bits2list, b)
    assert ("IndexError", True, False) in seen and ("ok", False, False) in seen and ("IndexError", False, True) in seen
