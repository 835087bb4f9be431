#!/usr/bin/env python3
"""
Golden vectors for the operations layer: runs the REFERENCE's
bx.intervals.operations.{intersect,subtract,coverage,merge,complement,base_coverage} with the
reference's own readers (built out of tree by oracle/build_pyref.sh, imported from $PYREF/lib) on
seeded synthetic inputs and writes inputs + everything observable to tests/golden/operations.json:
the yielded objects (type + text), and the readers' skip bookkeeping.

Build-container only (needs /root/reference via $PYREF); the GPU tests read just the JSON.
Test infrastructure -- nothing in the product imports this.
"""
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PYREF = os.environ.get("PYREF", "/tmp/bxref")
sys.path.insert(0, os.path.join(PYREF, "lib"))

from bx.intervals.io import GenomicIntervalReader, NiceReaderWrapper  # noqa: E402  (the reference's)
from bx.intervals.operations.base_coverage import base_coverage  # noqa: E402
from bx.intervals.operations.complement import complement  # noqa: E402
from bx.intervals.operations.coverage import coverage  # noqa: E402
from bx.intervals.operations.intersect import intersect  # noqa: E402
from bx.intervals.operations.merge import merge  # noqa: E402
from bx.intervals.operations.subtract import subtract  # noqa: E402
from bx.intervals.cluster import ClusterTree  # noqa: E402  (the reference's extension: cluster.pyx + src/cluster.c)
from bx.intervals.operations.find_clusters import find_clusters  # noqa: E402
from bx.intervals.operations.join import join  # noqa: E402
from bx.tabular.io import Comment, Header  # noqa: E402

assert "bxref" in sys.modules["bx.intervals.operations.intersect"].__file__ or PYREF in sys.modules["bx.intervals.operations.intersect"].__file__


def bed(rows, extra=True):
    return ["%s\t%d\t%d" % r[:3] + ("\tn%d\t0\t%s" % (i, r[3] if len(r) > 3 else "+") if extra else "") + "\n" for i, r in enumerate(rows)]


def random_rows(seed, n, chroms, span, lmax, zero=0.05):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        c = chroms[int(rng.integers(0, len(chroms)))]
        s = int(rng.integers(0, span))
        ln = 0 if rng.random() < zero else int(rng.integers(1, lmax))
        out.append((c, s, s + ln, "+-"[int(rng.integers(0, 2))]))
    return out


PRIMARY_MESSY = (
    ["#chrom\tstart\tend\tname\tscore\tstrand\n", "# a comment\n", "\n", "track name=x\n"]
    + bed([("chr1", 10, 50), ("chr1", 40, 45), ("chr1", 100, 100), ("chr2", 5, 500), ("chrX", 1, 9)])
    + ["chr1\tabc\t20\tbad\t0\t+\n", "chr1\t30\t20\trev\t0\t+\n", "chr1\t7\n", "chr1\t1\t2\tn\t0\t?\n"]
    + bed([("chr1", 180, 260), ("chr1", 250, 400), ("chr1", 990, 1000), ("chr1", 995, 1000), ("chr2", 450, 500), ("chr2", 499, 500),
           (" chr1 ", 60, 75, "."), ("chr1", 900, 1100), ("chr2", 0, 1)])
    + ["chr2\t1\tx\n"] * 9
    + bed([("chr1", 300, 320), ("chr3", 5, 6)])
)
SECOND_A = (
    ["#second\n"]
    + bed([("chr1", 20, 30), ("chr1", 25, 44), ("chr1", 200, 300), ("chr1", 980, 1000), ("chr2", 100, 200), ("chr2", 480, 500),
           ("chr3", 0, 10)], extra=False)
    + ["chr1\t-5\t3\n", "chr1\tzz\t3\n", "chr2\t490\t600\n", "chr1\t310\t310\n"]
)
SECOND_B = bed([("chr1", 0, 28), ("chr1", 210, 990), ("chr2", 150, 490), ("chr4", 1, 2)], extra=False)
LENS = {"chr1": 1000, "chr2": 500}

RAND_P = bed(random_rows(11, 400, ["chr1", "chr2", "chr5"], 20000, 300))
RAND_A = bed(random_rows(12, 300, ["chr1", "chr2", "chr6"], 20000, 500), extra=False)
RAND_B = bed(random_rows(13, 300, ["chr1", "chr2"], 20000, 900), extra=False)


def nice(lines, **kw):
    return NiceReaderWrapper(list(lines), **kw)


def plain(lines, **kw):
    return GenomicIntervalReader(list(lines), **kw)


def tell(item):
    if isinstance(item, Header):
        return ["header", str(item)]
    if isinstance(item, Comment):
        return ["comment", str(item)]
    if isinstance(item, list):
        return ["list", list(item)]
    return ["interval", [str(f) for f in item.fields], item.chrom, int(item.start), int(item.end), item.strand]


def skips(reader):
    if not hasattr(reader, "skipped"):
        return None
    return {"skipped": reader.skipped, "skipped_lines": [list(t) for t in reader.skipped_lines]}


def run(fn):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        try:
            out = [tell(x) for x in fn()]
            err = None
        except Exception as e:  # what escapes is part of the behaviour
            out, err = None, [type(e).__name__, str(e)]
    return out, err, [str(x.message) for x in w]


CASES = []


INPUTS = {}


def case(name, op, inputs, readers, params):
    keys = []
    for lines in inputs:  # every distinct input is stored once
        key = next((k for k, v in INPUTS.items() if v is lines or v == lines), None)
        if key is None:
            key = "in%d" % len(INPUTS)
            INPUTS[key] = lines
        keys.append(key)
    CASES.append(dict(name=name, op=op, inputs=keys, readers=readers, params=params))


for pieces in (True, False):
    for mincols in (1, 5, 0):
        for lens in ({}, LENS):
            tag = "p%d_m%d_%s" % (pieces, mincols, "lens" if lens else "nolens")
            case("intersect_messy_" + tag, "intersect", [PRIMARY_MESSY, SECOND_A], ["nice", "plain"], dict(pieces=pieces, mincols=mincols, lens=lens))
            case("subtract_messy_" + tag, "subtract", [PRIMARY_MESSY, SECOND_A], ["nice", "plain"], dict(pieces=pieces, mincols=mincols, lens=lens))
case("intersect_three", "intersect", [PRIMARY_MESSY, SECOND_A, SECOND_B], ["nice", "plain", "plain"], dict(lens=LENS))
case("subtract_three", "subtract", [PRIMARY_MESSY, SECOND_A, SECOND_B], ["nice", "plain", "plain"], dict(lens=LENS))
case("intersect_nocomments", "intersect", [PRIMARY_MESSY, SECOND_A], ["nice", "plain"], dict(comments=False))
case("subtract_nocomments", "subtract", [PRIMARY_MESSY, SECOND_A], ["nice", "plain"], dict(comments=False))
case("intersect_plain_primary", "intersect", [RAND_P, RAND_A], ["plain", "plain"], dict())
case("intersect_plain_primary_bad", "intersect", [PRIMARY_MESSY, SECOND_A], ["plain", "plain"], dict())
for pieces in (True, False):
    case("intersect_random_p%d" % pieces, "intersect", [RAND_P, RAND_A], ["nice", "plain"], dict(pieces=pieces, mincols=3))
    case("subtract_random_p%d" % pieces, "subtract", [RAND_P, RAND_A], ["nice", "plain"], dict(pieces=pieces, mincols=3))
case("intersect_random_three", "intersect", [RAND_P, RAND_A, RAND_B], ["nice", "plain", "plain"], dict())
case("subtract_random_three", "subtract", [RAND_P, RAND_A, RAND_B], ["nice", "plain", "plain"], dict())
case("coverage_messy", "coverage", [PRIMARY_MESSY, SECOND_A], ["nice", "plain"], dict())
case("coverage_three", "coverage", [PRIMARY_MESSY, SECOND_A, SECOND_B], ["nice", "plain", "plain"], dict(comments=False))
case("coverage_random", "coverage", [RAND_P, RAND_A, RAND_B], ["nice", "plain", "plain"], dict())
case("merge_messy", "merge", [SECOND_A], ["plain"], dict())
case("merge_primary", "merge", [PRIMARY_MESSY], ["plain"], dict())
case("merge_random", "merge", [RAND_A + RAND_B], ["plain"], dict())
case("complement_messy", "complement", [SECOND_A], ["plain"], dict(lens=LENS))
case("complement_nolens", "complement", [SECOND_B], ["plain"], dict(lens={}))
case("complement_random", "complement", [RAND_A], ["plain"], dict(lens={"chr1": 21000, "chr2": 20500, "chr6": 30000}))
case("base_coverage_messy", "base_coverage", [SECOND_A], ["plain"], dict())
case("base_coverage_random", "base_coverage", [RAND_A + RAND_B], ["plain"], dict())

OPS = dict(intersect=intersect, subtract=subtract, coverage=coverage, merge=merge, complement=complement, base_coverage=base_coverage)
MAKE = dict(nice=nice, plain=plain)


def cluster_cases():
    """ClusterTree vectors: the reference's own test inputs (cluster_tests.py:15-130) and seeded random trees."""
    fixed = [
        (0, 0, [(3, 4, 0), (6, 7, 1), (9, 10, 2), (1, 2, 3), (3, 8, 4)]),
        (0, 0, [(1, 4, 0), (4, 5, 1)]),
        (0, 0, [(1, 2, 0), (4, 5, 1), (2, 4, 2)]),
        (0, 0, [(1, 2, 0), (8, 9, 1), (3, 4, 2), (5, 6, 3), (7, 8, 4), (1, 10, 5)]),
        (0, 0, [(1, 1, 0), (1, 2, 1), (3, 4, 2), (3, 4, 3), (1, 4, 4)]),
        (0, 2, [(3, 4, 0), (6, 7, 1), (9, 10, 2), (1, 2, 3), (3, 8, 4)]),
        (1, 0, [(3, 4, 0), (6, 7, 1), (9, 10, 2), (1, 2, 3), (3, 8, 4)]),
        (0, 0, [(6, 7, 1), (1, 2, 3), (9, 10, 2), (3, 4, 0), (3, 8, 4)]),
        (0, 0, [(3, 4, 1), (13, 14, 6), (21, 22, 14), (5, 6, 2), (4, 10, 11), (1, 2, 0), (11, 12, 5), (1, 3, 10), (7, 8, 3),
                (15, 16, 7), (15, 20, 13), (19, 20, 9), (10, 15, 12), (17, 18, 8), (9, 10, 4)]),
        (0, 0, []),
    ]
    rng = np.random.default_rng(77)
    for n, span, lmax in ((50, 400, 20), (300, 4000, 40), (300, 100000, 300), (600, 10**9, 10**6), (500, 30, 5)):
        for md in (0, 1, 7, 1000):
            s = rng.integers(-span, span, size=n)
            e = s + rng.integers(0, lmax, size=n)
            ids = rng.integers(-1000, 100000, size=n)
            fixed.append((md, int(rng.integers(0, 4)), [(int(a), int(b), int(i)) for a, b, i in zip(s, e, ids)]))
    out = []
    for md, mn, triples in fixed:
        t = ClusterTree(md, mn)
        for a, b, i in triples:
            t.insert(a, b, i)
        out.append(dict(max_dist=md, min_intervals=mn, triples=triples, regions=[[a, b, ids] for a, b, ids in t.getregions()],
                        lines=t.getlines()))
    return out


def find_clusters_cases():
    out = []
    for name, lines, kind, params in (("messy", PRIMARY_MESSY, "nice", dict(mincols=1, minregions=2)),
                                      ("messy_wide", PRIMARY_MESSY, "nice", dict(mincols=200, minregions=1)),
                                      ("random", RAND_P, "plain", dict(mincols=10, minregions=2))):
        reader = MAKE[kind](lines)
        chroms, extra = find_clusters(reader, **params)
        out.append(dict(name=name, input=name, reader=kind, params=params,
                        chroms={c: dict(regions=[[a, b, ids] for a, b, ids in t.getregions()], lines=t.getlines()) for c, t in chroms.items()},
                        chrom_order=list(chroms), extra={str(k): tell(v) for k, v in extra.items()}, primary=skips(reader)))
    return out


def join_cases():
    """join: the match order inside one left row is random in the reference (treap priorities); the test canonicalises it."""
    import random

    out = []
    small_l = bed([("chr1", 10, 50), ("chr1", 40, 45), ("chr1", 100, 100), ("chr2", 5, 500), ("chrX", 1, 9), ("chr1", 30, 31), ("chr1", 10, 50)])
    small_r = ["#r\n"] + bed([("chr1", 20, 30), ("chr1", 25, 44), ("chr1", 44, 60), ("chr1", 10, 50), ("chr1", 50, 70), ("chr1", 20, 30),
                              ("chr2", 100, 200), ("chr3", 0, 10), ("chr1", 0, 10)], extra=False)
    inputs = dict(small_l=["#l\n", "# c\n"] + small_l, small_r=small_r, rand_l=RAND_P, rand_r=RAND_A)
    for name, lkey, rkey, params in (("small", "small_l", "small_r", dict()), ("small_m5", "small_l", "small_r", dict(mincols=5)),
                                     ("small_nofill", "small_l", "small_r", dict(leftfill=False, rightfill=False)),
                                     ("small_m0", "small_l", "small_r", dict(mincols=0, rightfill=False)),
                                     ("random", "rand_l", "rand_r", dict(mincols=3)), ("random_m40", "rand_l", "rand_r", dict(mincols=40, leftfill=False))):
        random.seed(5)
        items = [tell(x) for x in join(nice(inputs[lkey]), plain(inputs[rkey]), **params)]
        out.append(dict(name=name, left=lkey, right=rkey, params=params, output=items))
    return dict(inputs=inputs, cases=out)


def reader_cases():
    """Plain iteration of the reference's readers over every input: items, escaping error, skip bookkeeping, header."""
    out = []
    extra_inputs = dict(
        spaces=["chr1 10 20 a 0 +\n", "chr1\t30\t40\n", "chr2  5   9\n"],
        crlf=["#h1\th2\th3\r\n", "chr1\t1\t2\r\n", "\r\n", "chr1\t3\t4\n"],
        noheader=["chr1\t1\t2\n", "#late comment\n", "track x\n", "chr1\t2\t9\tname\t0\t-\n", "chr1\t2\t9\tname\t0\t.\n"],
    )
    everything = dict(INPUTS)
    everything.update(extra_inputs)
    variants = [("nice", {}), ("plain", {}), ("nice", dict(return_header=False, return_comments=False)),
                ("nice", dict(allow_spaces=True)), ("nice", dict(fix_strand=True, default_strand="-")),
                ("nice", dict(chrom_col=0, start_col=1, end_col=2, strand_col=3))]
    for key, lines in everything.items():
        if len(lines) > 60:  # the random files add bulk, not cases
            continue
        for kind, kw in variants:
            r = MAKE[kind](lines, **kw)
            items, err = [], None
            try:
                for x in r:
                    items.append(tell(x))
            except Exception as e:
                err = [type(e).__name__, str(e)]
            out.append(dict(input=key, reader=kind, kwargs=kw, items=items, error=err, skips=skips(r), linenum=r.linenum,
                            header=str(r.header) if r.header is not None else None))
    return dict(inputs=extra_inputs, cases=out)


def main():
    out = []
    for c in CASES:
        readers = [MAKE[k](INPUTS[key]) for k, key in zip(c["readers"], c["inputs"])]
        op, params = c["op"], dict(c["params"])
        if op in ("intersect", "subtract", "coverage"):
            fn = lambda: OPS[op](readers, **params)  # noqa: E731
        elif op == "merge":
            fn = lambda: ([list(x)] if isinstance(x, list) else [x] for x in OPS[op](readers[0]))  # noqa: E731  (the row list is reused: copy it)
            fn = (lambda inner: (lambda: (y for chunk in inner() for y in chunk)))(fn)
        elif op == "complement":
            fn = lambda: OPS[op](readers[0], params["lens"])  # noqa: E731
        else:
            fn = None
        if op == "base_coverage":
            try:
                value, err = OPS[op](readers[0]), None
            except Exception as e:
                value, err = None, [type(e).__name__, str(e)]
            rec = dict(c, value=value, error=err, warnings=[], output=None, primary=None)
        else:
            items, err, warns = run(fn)
            rec = dict(c, output=items, error=err, warnings=warns, primary=skips(readers[0]))
        out.append(rec)
        print("%-40s %s" % (c["name"], "items=%s err=%s" % (len(rec["output"]) if rec["output"] is not None else rec.get("value"), rec["error"])))
    path = os.path.join(ROOT, "tests", "golden", "operations.json")
    with open(path, "w") as f:
        json.dump(dict(generator="oracle/gen_golden_ops.py", inputs=INPUTS, cases=out, clusters=cluster_cases(),
                       join=join_cases(), readers=reader_cases(), find_clusters=find_clusters_cases(), find_clusters_inputs=dict(messy=PRIMARY_MESSY, messy_wide=PRIMARY_MESSY, random=RAND_P)),
                  f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
