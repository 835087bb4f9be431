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


if __name__ == "__main__":
    doc = dict(generator="oracle/gen_golden_extra.py", builders=builder_cases(), quicksect=quicksect_cases())
    path = os.path.join(ROOT, "tests", "golden", "builders_quicksect.json")
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")
