"""
GPU parity of the smaller callers against the reference's own outputs (tests/golden/builders_quicksect.json,
made by oracle/gen_golden_extra.py from lib/bx/bitset_builders.py:107-169 and operations/quicksect.py:11-126):
same bitsets or the same exception from the three list/proximity/by-chromosome builders, the same hit sets and
the same traverse order from the multi-chromosome interval tree.
"""
import json
import os

import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "builders_quicksect.json")


@pytest.fixture(scope="module")
def doc():
    with open(GOLDEN) as f:
        return json.load(f)


def _runs(bits):
    s, e = bits.runs()
    return [[int(a), int(b)] for a, b in zip(s, e)]


def _observe(fn):
    try:
        got = fn()
    except Exception as e:
        return dict(error=[type(e).__name__, str(e)])
    if isinstance(got, dict):
        return dict(order=list(got), runs={c: _runs(b) for c, b in got.items()})
    return dict(runs=_runs(got))


def test_list_proximity_and_by_chrom_builders(doc):
    import bx.bitset_builders as bb

    assert len(doc["builders"]) >= 25
    for case in doc["builders"]:
        a = case["args"]
        if case["fn"] == "from_list":
            got = _observe(lambda: bb.binned_bitsets_from_list(a["rows"]))
        elif case["fn"] == "proximity":
            got = _observe(lambda: bb.binned_bitsets_proximity(iter(a["lines"]), **a["kw"]))
        else:
            got = _observe(lambda: bb.binned_bitsets_by_chrom(iter(a["lines"]), a["chrom"], **a["kw"]))
        assert got == case["want"], case["name"]


class _Row:
    def __init__(self, chrom, start, end):
        self.chrom, self.start, self.end = chrom, start, end


def test_quicksect_tree_reports_the_same_sets_and_traverses_in_the_same_order(doc):
    from bx.intervals.operations.quicksect import IntervalTree

    for case in doc["quicksect"]:
        tree = IntervalTree()
        for i, (c, s, e) in enumerate(case["rows"]):
            tree.insert(_Row(c, s, e), linenum=i, other="row%d" % i)
        assert list(tree.chroms) == case["chrom_order"]
        total = 0
        for (c, s, e), want in zip(case["queries"], case["found"]):
            got = []
            tree.intersect(_Row(c, s, e), lambda node: got.append([node.linenum, node.start, node.end, node.other]))
            assert sorted(got) == want, (case["name"], c, s, e)
            total += len(got)
        assert total > 0
        order = []
        tree.traverse(lambda node: order.append(node.linenum))
        assert order == case["traverse"], case["name"]
        for c, want in case["per_chrom"].items():
            seq = []
            tree.chroms[c].traverse(lambda node: seq.append(node.linenum))
            assert seq == want
        # the additive batch form answers the same questions in one launch per chromosome
        for c in case["chrom_order"]:
            qs = [(s, e) for cc, s, e in case["queries"] if cc == c]
            want = [w for (cc, _, _), w in zip(case["queries"], case["found"]) if cc == c]
            off, nodes = tree.intersect_batch(c, [s for s, _ in qs], [e for _, e in qs])
            for k, w in enumerate(want):
                assert sorted(n.linenum for n in nodes[int(off[k]):int(off[k + 1])]) == [r[0] for r in w]
        off, nodes = tree.intersect_batch("nowhere", [1], [5])
        assert list(off) == [0, 0] and nodes == []


def test_reference_module_names_resolve_to_the_engine():
    import bx.intervals.io as gio
    import bx.tabular.io as tio
    from bx.intervals.operations import MAX_END, bits_clear_in_range, bits_set_in_range  # noqa: F401
    from bx.intervals.operations.base_coverage import base_coverage  # noqa: F401
    from bx.intervals.operations.complement import complement  # noqa: F401
    from bx.intervals.operations.coverage import coverage  # noqa: F401
    from bx.intervals.operations.find_clusters import find_clusters  # noqa: F401
    from bx.intervals.operations.intersect import intersect
    from bx.intervals.operations.join import join  # noqa: F401
    from bx.intervals.operations.merge import merge  # noqa: F401
    from bx.intervals.operations.subtract import subtract  # noqa: F401

    assert MAX_END == 512 * 1024 * 1024
    assert issubclass(gio.GenomicInterval, tio.TableRow) and intersect.__module__ == "bxmi.operations"
    a = ["chr1\t10\t50\n", "chr1\t100\t120\n"]
    b = ["chr1\t40\t110\n"]
    out = [str(x) for x in intersect([gio.GenomicIntervalReader(iter(a)), gio.GenomicIntervalReader(iter(b))])]
    assert out == ["chr1\t40\t50", "chr1\t100\t110"]


def test_bitset_utils_match_reference_vectors():
    """bx.bitset_utils (lib/bx/bitset_utils.py:12-90) on the engine: interval lists through list2bits / bits2list, intersect,
    subtract, union, complement (inside the list's own range only), bitset_interval_intersect (a run that starts before the
    window's end is reported whole) -- results and failures as captured from the reference (oracle/gen_golden_extra.py --utils)."""
    import bx.bitset_utils as bu

    with open(os.path.join(os.path.dirname(GOLDEN), "bitset_utils_concat.json")) as f:
        cases = json.load(f)["utils"]

    def call(fn, *a):
        try:
            return dict(result=[list(x) for x in fn(*a)])
        except Exception as e:
            return dict(error=[type(e).__name__, str(e)])

    for k, c in enumerate(cases):
        a, b = [tuple(x) for x in c["a"]], [tuple(x) for x in c["b"]]
        assert call(bu.bitset_intersect, a, b) == c["intersect"], (k, "intersect")
        assert call(bu.bitset_subtract, a, b) == c["subtract"], (k, "subtract")
        assert call(bu.bitset_union, a) == c["union"], (k, "union")
        assert call(bu.bitset_complement, b) == c["complement"], (k, "complement")
        lo, hi = c["window"]
        assert call(lambda: bu.bitset_interval_intersect(bu.list2bits(b), lo, hi)) == c["interval_intersect"], (k, "interval_intersect")
