#!/usr/bin/env python3
"""
Golden vectors for the smaller callers of the hot path: the REFERENCE's
bx.bitset_builders.{binned_bitsets_from_list, binned_bitsets_proximity, binned_bitsets_by_chrom}
(lib/bx/bitset_builders.py:107-169) and bx.intervals.operations.quicksect.IntervalTree
(quicksect.py:11-126), run here from the out-of-tree build made by oracle/build_pyref.sh.
Writes inputs + everything observable to tests/golden/builders_quicksect.json.

The treap's report order depends on random priorities, so what is recorded per query is the SET of
reported line numbers (sorted); `traverse` is deterministic (in-order) and recorded as is.

Build-container only; test infrastructure -- nothing in the product imports this.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PYREF = os.environ.get("PYREF", "/tmp/bxref")
sys.path.insert(0, os.path.join(PYREF, "lib"))

import bx.bitset_builders as bb  # noqa: E402  (the reference's)
from bx.intervals.operations.quicksect import IntervalTree  # noqa: E402

assert PYREF in bb.__file__


def runs(bits):
    out, end = [], 0
    while end < bits.size:
        start = bits.next_set(end)
        if start == bits.size:
            break
        end = bits.next_clear(start)
        out.append([start, end])
    return out


def observe(fn):
    try:
        got = fn()
    except Exception as e:  # the reference's own failure is the expected behaviour
        return dict(error=[type(e).__name__, str(e)])
    if isinstance(got, dict):
        return dict(order=list(got), runs={c: runs(b) for c, b in got.items()})
    return dict(runs=runs(got))


def bed_lines(seed, n, chroms, span, lmax):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        c = chroms[int(rng.integers(0, len(chroms)))]
        s = int(rng.integers(0, span))
        e = s + int(rng.integers(-3, lmax))
        strand = "+-."[int(rng.integers(0, 3))]
        out.append("%s\t%d\t%d\tn%d\t0\t%s\n" % (c, s, max(e, 0), i, strand) if i % 7 else "%s %d %d\n" % (c, s, max(e, 0)))
    return out


def builder_cases():
    rand = ["# head\n"] + bed_lines(11, 400, ["chr1", "chr2", "chrX"], 100000, 900)
    cases = []
    for name, fn, args in [
        ("list-plain", "from_list", dict(rows=[["chr1", 10, 20], ["chr2", "5", "9"], ["chr1", 15, 40], ["#odd", 1, 2], ["chr1", 7, 7]])),
        ("list-empty", "from_list", dict(rows=[])),
        ("list-reversed", "from_list", dict(rows=[["chr1", 10, 20], ["chr1", 30, 25], ["chr2", 1, 2]])),
        ("list-negative", "from_list", dict(rows=[["chr1", -5, 20]])),
        ("list-past-max", "from_list", dict(rows=[["chr1", 536870900, 536870913]])),
        ("list-at-max", "from_list", dict(rows=[["chr1", 536870900, 536870912]])),
        ("list-start-at-max", "from_list", dict(rows=[["chr1", 536870912, 536870912]])),
        ("list-too-big", "from_list", dict(rows=[["chr1", 1, 5000000000]])),
        ("list-bad-int", "from_list", dict(rows=[["chr1", "x", 3]])),
        ("prox-none", "proximity", dict(lines=rand, kw={})),
        ("prox-up", "proximity", dict(lines=rand, kw=dict(upstream=250))),
        ("prox-down", "proximity", dict(lines=rand, kw=dict(downstream=1000))),
        ("prox-both", "proximity", dict(lines=rand, kw=dict(upstream=70, downstream=3))),
        ("prox-edges", "proximity", dict(lines=["chr1\t5\t9\ta\t0\t-\n", "chr1\t536870000\t536870900\tb\t0\t+\n", "chr2\t3\t3\n", "chr3\t9\t4\n", "chr1\t100\t200\tc\t0\t-\n"],
                                          kw=dict(upstream=2000, downstream=50))),
        ("prox-strandcol", "proximity", dict(lines=["chr1\t50\t60\t-\n", "chr1\t500\t600\t+\n"], kw=dict(strand_col=3, upstream=10))),
        ("prox-blank", "proximity", dict(lines=["chr1\t5\t9\n", "\n", "chr1\t50\t90\n"], kw={})),
        ("prox-short", "proximity", dict(lines=["chr1\t5\n"], kw={})),
        ("prox-bad-int", "proximity", dict(lines=["chr1\t5\t9\n", "chr1\t5\tz\n"], kw={})),
        ("prox-negative", "proximity", dict(lines=["chr1\t-5\t9\n"], kw={})),
        ("prox-negative-grown", "proximity", dict(lines=["chr1\t-5\t9\n"], kw=dict(upstream=1))),
        ("prox-past-max", "proximity", dict(lines=["chr1\t5\t536870913\n"], kw={})),
        ("prox-past-max-grown", "proximity", dict(lines=["chr1\t5\t536870913\n"], kw=dict(downstream=1))),
        ("chrom-rand", "by_chrom", dict(lines=rand, chrom="chr2", kw={})),
        ("chrom-absent", "by_chrom", dict(lines=rand, chrom="chr9", kw={})),
        ("chrom-cols", "by_chrom", dict(lines=["a\tchr1\t5\t9\n", "b\tchr2\t1\t2\n", "c\tchr1\t100\t120\n"], chrom="chr1", kw=dict(chrom_col=1, start_col=2, end_col=3))),
        ("chrom-reversed", "by_chrom", dict(lines=["chr1\t5\t9\n", "chr1\t9\t5\n"], chrom="chr1", kw={})),
        ("chrom-blank", "by_chrom", dict(lines=["chr1\t5\t9\n", "\n"], chrom="chr1", kw={})),
    ]:
        if fn == "from_list":
            want = observe(lambda: bb.binned_bitsets_from_list(args["rows"]))
        elif fn == "proximity":
            want = observe(lambda: bb.binned_bitsets_proximity(iter(args["lines"]), **args["kw"]))
        else:
            want = observe(lambda: bb.binned_bitsets_by_chrom(iter(args["lines"]), args["chrom"], **args["kw"]))
        cases.append(dict(name=name, fn=fn, args=args, want=want))
        print("%-22s %s" % (name, want.get("error") or "runs=%d" % (sum(len(v) for v in want["runs"].values()) if isinstance(want["runs"], dict) else len(want["runs"]))))
    return cases


class Row:
    def __init__(self, chrom, start, end):
        self.chrom, self.start, self.end = chrom, start, end


def quicksect_cases():
    out = []
    for name, seed, n, nq, span, lmax, chroms in [
        ("small", 5, 60, 40, 300, 40, ["chr1", "chr2"]),
        ("ties", 6, 300, 100, 60, 8, ["chr1"]),
        ("wide", 7, 2000, 300, 1000000, 20000, ["chr1", "chr2", "chr3"]),
    ]:
        rng = np.random.default_rng(seed)
        rows = [(chroms[int(rng.integers(0, len(chroms)))], int(s), int(s) + int(ln))
                for s, ln in zip(rng.integers(0, span, n), rng.integers(0, lmax, n))]
        queries = [(chroms[int(rng.integers(0, len(chroms)))], int(s), int(s) + int(ln))
                   for s, ln in zip(rng.integers(0, span, nq), rng.integers(0, lmax, nq))] + [("nowhere", 0, 10)]
        tree = IntervalTree()
        for i, (c, s, e) in enumerate(rows):
            tree.insert(Row(c, s, e), linenum=i, other="row%d" % i)
        found = []
        for c, s, e in queries:
            got = []
            tree.intersect(Row(c, s, e), lambda node: got.append((node.linenum, node.start, node.end, node.other)))
            found.append(sorted(map(list, got)))
        order = []
        tree.traverse(lambda node: order.append(node.linenum))
        per_chrom = {}
        for c in tree.chroms:
            seq = []
            tree.chroms[c].traverse(lambda node: seq.append(node.linenum))
            per_chrom[c] = seq
        out.append(dict(name=name, rows=rows, queries=queries, found=found, traverse=order, chrom_order=list(tree.chroms), per_chrom=per_chrom))
        print("quicksect %-8s rows=%d hits=%d" % (name, n, sum(map(len, found))))
    return out


if __name__ == "__main__" and "--utils" not in sys.argv:
    doc = dict(generator="oracle/gen_golden_extra.py", builders=builder_cases(), quicksect=quicksect_cases())
    path = os.path.join(ROOT, "tests", "golden", "builders_quicksect.json")
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


# --------------------------------------------------------------------------- #
# bx.bitset_utils and bx.intervals.operations.concat (round 5): small callers of the hot-path API
# --------------------------------------------------------------------------- #
def gen_utils():
    import io

    import bx.bitset_utils as bu
    from bx.intervals.io import GenomicIntervalReader
    from bx.intervals.operations.concat import concat

    assert PYREF in bu.__file__

    def call(fn, *a):
        try:
            return dict(result=[list(x) for x in fn(*a)])
        except Exception as e:
            return dict(error=[type(e).__name__, str(e)])

    def exons(rng, n, span, lmax, messy):
        out = []
        for _ in range(n):
            s = int(rng.integers(0, span))
            ln = int(rng.integers(0 if messy else 1, lmax))
            out.append([s, s + ln])
        if messy and n > 3:
            out[2] = [out[2][1] + 5, out[2][1]]  # reversed: set_range gets a negative count
        return out

    cases = []
    for k in range(24):
        rng = np.random.default_rng(9100 + k)
        messy = k % 6 == 5
        a = exons(rng, int(rng.integers(1, 40)), 5000 if k % 2 else 3_000_000, 400, messy)
        b = exons(rng, int(rng.integers(1, 40)), 5000 if k % 2 else 3_000_000, 400, False)
        c = dict(a=a, b=b)
        c["intersect"] = call(bu.bitset_intersect, a, b)
        c["subtract"] = call(bu.bitset_subtract, a, b)
        c["union"] = call(bu.bitset_union, a)
        c["complement"] = call(bu.bitset_complement, b)
        lo, hi = int(rng.integers(0, 3000)), int(rng.integers(3000, 6000))
        c["window"] = [lo, hi]
        try:
            c["interval_intersect"] = dict(result=[list(x) for x in bu.bitset_interval_intersect(bu.list2bits(b), lo, hi)])
        except Exception as e:
            c["interval_intersect"] = dict(error=[type(e).__name__, str(e)])
        cases.append(c)
    cases.append(dict(a=[], b=[], intersect=call(bu.bitset_intersect, [], []), subtract=call(bu.bitset_subtract, [], []), union=call(bu.bitset_union, []),
                      complement=call(bu.bitset_complement, []), window=[0, 10],
                      interval_intersect=dict(result=[list(x) for x in bu.bitset_interval_intersect(bu.list2bits([]), 0, 10)])))
    # concat: two inputs of different column orders, with comments, a header, long and short rows
    f1 = "#header one\nchr1\t10\t20\tn1\t0\t+\nchr1\t30\t40\tn2\t0\t-\textra\n# a comment\nchr2\t5\t9\n"
    f2 = "#header two\n+\tchr3\t100\t200\tx\ty\tz\tw\nchr4bad\n-\tchr3\t300\t400\n"
    f3 = "#header two\n+\tchr3\t100\t200\tx\ty\tz\tw\n#mid\n-\tchr3\t300\t400\n.\tchr5\t1\t2\tq\n"
    ccases = []
    for second in (f2, f3):
        for sameformat in (True, False):
            for comments, header in ((True, True), (False, False), (True, False)):
                r1 = GenomicIntervalReader(io.StringIO(f1), chrom_col=0, start_col=1, end_col=2, strand_col=5)
                r2 = GenomicIntervalReader(io.StringIO(second), chrom_col=1, start_col=2, end_col=3, strand_col=0)
                out, err = [], None
                try:
                    for x in concat([r1, r2], comments=comments, header=header, sameformat=sameformat):
                        out.append(str(x))
                except Exception as e:
                    err = [type(e).__name__, str(e)]
                ccases.append(dict(sameformat=sameformat, comments=comments, header=header, f1=f1, f2=second, result=out, error=err))
    path = os.path.join(ROOT, "tests", "golden", "bitset_utils_concat.json")
    with open(path, "w") as f:
        json.dump(dict(source="bx.bitset_utils / bx.intervals.operations.concat of the reference (0.14.0)", utils=cases, concat=ccases), f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__" and "--utils" in sys.argv:
    gen_utils()
