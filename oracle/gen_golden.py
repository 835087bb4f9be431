#!/usr/bin/env python3
"""
oracle/gen_golden.py -- produce tests/golden/* by running the REAL reference.

Build-container only.  Imports bx.bitset / bx.intervals.intersection from the
out-of-tree build made by oracle/build_pyref.sh (PYTHONPATH=/tmp/bxref/lib) and
runs the reference's four CLI scripts straight from /root/reference/scripts.
Only inputs + expected outputs are written (data, never reference source).

    ./oracle/build_pyref.sh && python oracle/gen_golden.py [--scale]

--scale additionally runs the 10M-target / 1M-query-subsample point of cfg 2
through the reference treap (about 5 minutes, ~2 GB) and records its hash.
--only genome | join | calibration | bitsets_genome | bitsets_genome_default add, to tests/golden/scale.json, the per-chromosome
hashes of configs[3] (synth.cfg4), the hit-list hash of configs[4] at 50M targets
(~10 GB, ~30 min), the reference-vs-port timing BASELINE.md 4 asks for, and the per-chromosome popcounts /
run-list hashes of configs[2] (two hg19-sized dicts of BinnedBitSets) from the real bx.bitset.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PYREF = os.environ.get("PYREF", "/tmp/bxref")
REFERENCE = os.environ.get("REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")

sys.path.insert(0, os.path.join(PYREF, "lib"))
sys.path.insert(1, os.path.join(ROOT, "bx-python_amd"))

import bx.bitset as rb  # noqa: E402  (the real reference)
import bx.intervals.intersection as ri  # noqa: E402
from bxmi import synth  # noqa: E402

assert rb.__file__.startswith(PYREF) and ri.__file__.startswith(PYREF), "must import the reference build"


def dump(name, obj):
    path = os.path.join(GOLD, name)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote %s (%d bytes)" % (name, os.path.getsize(path)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# --------------------------------------------------------------------------- #
# 1. IntervalTree.find / traverse / before / after on random trees
# --------------------------------------------------------------------------- #
def gen_trees():
    cases = []
    modes = ["proper", "zero", "reversed", "negative", "dups", "proper_obj"]
    sizes = [1, 2, 3, 5, 17, 64, 200, 300]
    spans = [10, 50, 1000, 1_000_000]
    c = 0
    for mode in modes:
        for n in sizes:
            span = spans[c % len(spans)]
            rng = np.random.default_rng(1000 + c)
            c += 1
            lo = -span if mode == "negative" else 0
            s = rng.integers(lo, span, size=n)
            ln = rng.integers(0 if mode in ("zero", "dups") else 1, max(2, span // 4), size=n)
            if mode == "zero":
                ln[rng.random(n) < 0.5] = 0
            if mode == "dups":
                s = rng.integers(lo, max(lo + 1, lo + 4), size=n)
            e = s + ln
            if mode == "reversed":
                flip = rng.random(n) < 0.3
                s, e = np.where(flip, e, s), np.where(flip, s, e)
            tree = ri.IntervalTree()
            use_obj = mode == "proper_obj"
            for i in range(n):
                if use_obj:
                    tree.insert_interval(ri.Interval(int(s[i]), int(e[i]), value=i))
                else:
                    tree.insert(int(s[i]), int(e[i]), i)
            order = []
            tree.traverse(lambda node: order.append(node.interval.value if use_obj else node.interval))
            queries, hits = [], []
            for _ in range(40):
                qs = int(rng.integers(lo - 5, span + 5))
                kind = rng.integers(0, 5)
                if kind == 0:
                    qe = qs  # zero-length query
                elif kind == 1:
                    qe = qs - int(rng.integers(1, 10))  # reversed query
                else:
                    qe = qs + int(rng.integers(1, max(2, span // 3)))
                r = tree.find(qs, qe)
                queries.append([qs, qe])
                hits.append([x.value for x in r] if use_obj else list(r))
            # exact-boundary queries on the first few intervals
            for i in range(min(n, 5)):
                for qs, qe in ((int(s[i]), int(e[i])), (int(e[i]), int(e[i]) + 1), (int(s[i]) - 1, int(s[i]))):
                    r = tree.find(qs, qe)
                    queries.append([qs, qe])
                    hits.append([x.value for x in r] if use_obj else list(r))
            neigh = []
            if use_obj:
                for _ in range(30):
                    pos = int(rng.integers(lo - 5, span + 5))
                    k = int(rng.integers(1, 5))
                    md = int(rng.choice([0, 1, 5, 50, 2500, span]))
                    neigh.append(["before", pos, k, md, [x.value for x in tree.before(pos, num_intervals=k, max_dist=md)]])
                    neigh.append(["after", pos, k, md, [x.value for x in tree.after(pos, num_intervals=k, max_dist=md)]])
            cases.append(
                dict(mode=mode, n=n, span=span, starts=s.tolist(), ends=e.tolist(), order=order,
                     queries=queries, hits=hits, neighbours=neigh)
            )
    dump("ivtree_random.json", dict(source="bx.intervals.intersection.IntervalTree (reference 0.14.0)", cases=cases))


# --------------------------------------------------------------------------- #
# 2. BinnedBitSet op sequences
# --------------------------------------------------------------------------- #
def call(fn, *a):
    try:
        return ["ok", fn(*a)]
    except (IndexError, ValueError, OverflowError, TypeError) as ex:
        return [type(ex).__name__, str(ex)]


def bits_hex(b):
    v = np.array([b[i] for i in range(b.size)], dtype=np.uint8)
    return np.packbits(v, bitorder="little").tobytes().hex()


def gen_bitsets():
    cases = []
    c = 0
    for size in (95, 100, 997, 1000, 4096, 5000, 65539):
        for gran in (1, 3, 7, 10, 16, 64, 1024):
            rng = np.random.default_rng(2000 + c)
            c += 1
            A, B = rb.BinnedBitSet(size, gran), rb.BinnedBitSet(size, gran)
            sets = {"A": A, "B": B}
            ops = []
            nops = 70 if size <= 5000 else 40
            for _ in range(nops):
                which = "A" if rng.random() < 0.6 else "B"
                t = sets[which]
                k = rng.choice(
                    ["set_range", "set_range", "set_range", "count_range", "count_range", "next_set", "next_clear",
                     "get", "set", "clear", "invert", "iand", "ior", "bad"],
                )
                if k == "set_range" or k == "count_range":
                    s = int(rng.integers(0, size))
                    n = int(rng.integers(0, min(size - s, max(1, size // 3)) + 1))
                    ops.append([which, k, s, n, call(getattr(t, k), s, n)])
                elif k in ("next_set", "next_clear", "set", "clear"):
                    s = int(rng.integers(0, size))
                    ops.append([which, k, s, 0, call(getattr(t, k), s)])
                elif k == "get":
                    s = int(rng.integers(0, size))
                    ops.append([which, k, s, 0, call(t.__getitem__, s)])
                elif k == "invert":
                    if rng.random() < 0.5:
                        ops.append([which, k, 0, 0, call(t.invert)])
                elif k in ("iand", "ior"):
                    other = "B" if which == "A" else "A"
                    ops.append([which, k, other, 0, call(getattr(t, k), sets[other])])
                else:  # argument errors (bitset.pyx:177-192)
                    s, n = [(-3, 5), (size, 0), (size + 7, 1), (0, -2), (size - 1, 5), (5, size)][int(rng.integers(0, 6))]
                    m = ["set_range", "count_range"][int(rng.integers(0, 2))]
                    ops.append([which, m, s, n, call(getattr(t, m), s, n)])
                    ops.append([which, "next_set", s, 0, call(t.next_set, s)])
            final = {w: bits_hex(sets[w]) for w in ("A", "B")}
            full = {w: call(sets[w].count_range, 0, size) for w in ("A", "B")}
            cases.append(dict(size=size, granularity=gran, bin_size=A.bin_size, ops=ops, final=final, full_count=full))
    # big sizes: results only
    big = []
    for size, gran in ((16777217, 1024), (rb.MAX, 1024), (248956422, 1024), (2147483647, 1024), (rb.MAX, 100000)):
        rng = np.random.default_rng(2500 + len(big))
        A, B = rb.BinnedBitSet(size, gran), rb.BinnedBitSet(size, gran)
        limit = min(size, 16777216) if size == 16777217 else size  # float32 bin_size leaves pos 2^24 without a bin
        ops = []
        for j in range(60):
            which = "A" if j % 3 else "B"
            t = A if which == "A" else B
            s = int(rng.integers(0, limit - 3000))
            n = int(rng.integers(0, 3000)) if j % 7 else int(rng.integers(0, min(limit - s, 3 * A.bin_size)))
            ops.append([which, "set_range", s, n, call(t.set_range, s, n)])
        for j in range(30):
            which = "A" if j % 2 else "B"
            t = A if which == "A" else B
            s = int(rng.integers(0, limit - 5000))
            n = int(rng.integers(0, 5000))
            ops.append([which, "count_range", s, n, call(t.count_range, s, n)])
            ops.append([which, "next_set", s, 0, call(t.next_set, s)])
            ops.append([which, "next_clear", s, 0, call(t.next_clear, s)])
        ops.append(["A", "count_range", 0, limit, call(A.count_range, 0, limit)])
        ops.append(["A", "ior", "B", 0, call(A.ior, B)])
        ops.append(["A", "count_range", 0, limit, call(A.count_range, 0, limit)])
        ops.append(["B", "invert", 0, 0, call(B.invert)])
        ops.append(["A", "iand", "B", 0, call(A.iand, B)])
        ops.append(["A", "count_range", 0, limit, call(A.count_range, 0, limit)])
        ops.append(["B", "count_range", 0, limit, call(B.count_range, 0, limit)])  # ALL_ONE arithmetic
        ops.append(["B", "count_range", 7, limit - 7, call(B.count_range, 7, limit - 7)])
        ops.append(["B", "next_clear", 0, 0, call(B.next_clear, 0)])
        big.append(dict(size=size, granularity=gran, bin_size=A.bin_size, ops=ops))
    ctor = [[s, call(lambda s=s: rb.BinnedBitSet(s).size)] for s in (4000000000, 2147483648, 2147483647)]
    dump("binnedbitset_ops.json", dict(source="bx.bitset.BinnedBitSet (reference 0.14.0)", cases=cases, big=big, ctor=ctor, MAX=rb.MAX))


# --------------------------------------------------------------------------- #
# 3. CLI goldens: the four scripts run from the reference tree
# --------------------------------------------------------------------------- #
def run_script(name, args, stdin=None):
    env = dict(os.environ, PYTHONPATH=os.path.join(PYREF, "lib"), PYTHONWARNINGS="ignore")
    p = subprocess.run([sys.executable, os.path.join(REFERENCE, "scripts", name)] + args, input=stdin,
                       capture_output=True, text=True, env=env)
    return dict(stdout=p.stdout, returncode=p.returncode, stderr_tail=p.stderr.strip().splitlines()[-1:] if p.returncode else [])


def gen_cli():
    d = os.path.join(GOLD, "cli")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(3001)
    # small hand-shaped two-chromosome inputs (12 + 12 lines; comments, blank line, touching and nested ranges)
    a_lines = ["# comment line\n", "chr1\t10\t20\ta0\t0\t+\n", "chr1\t15\t40\ta1\t0\t-\n", "chr1\t100\t100\ta2\t0\t+\n",
               "chr2\t5\t9\ta3\t0\t+\n", "\n", "chr1\t300\t450\ta4\t0\t+\n", "chr3\t1\t1000\ta5\t0\t+\n",
               "chr2\t50\t75\ta6\t0\t-\n", "chr1\t40\t41\ta7\t0\t+\n", "chr2\t0\t5\ta8\t0\t+\n", "chr1\t449\t460\ta9\t0\t+\n",
               "chr2\t74\t200\ta10\t0\t+\n", "chr1\t19\t21\ta11\t0\t+\n"]
    b_lines = ["chr1\t0\t12\tb0\t0\t+\n", "chr1\t18\t30\tb1\t0\t+\n", "chr2\t8\t60\tb2\t0\t-\n", "chr1\t30\t35\tb3\t0\t+\n",
               "chr1\t99\t101\tb4\t0\t+\n", "chr2\t60\t61\tb5\t0\t+\n", "chr4\t0\t10\tb6\t0\t+\n", "chr1\t400\t449\tb7\t0\t+\n",
               "chr1\t440\t455\tb8\t0\t-\n", "chr2\t199\t300\tb9\t0\t+\n", "chr1\t41\t42\tb10\t0\t+\n", "chr1\t20\t20\tb11\t0\t+\n"]
    small_a, small_b = os.path.join(d, "small_a.bed"), os.path.join(d, "small_b.bed")
    open(small_a, "w").writelines(a_lines)
    open(small_b, "w").writelines(b_lines)
    # interval_join needs header-free, tab-separated files without blank/comment lines
    ja, jb = os.path.join(d, "join_a.bed"), os.path.join(d, "join_b.bed")
    open(ja, "w").writelines([l for l in a_lines if not l.startswith("#") and l.strip()])
    open(jb, "w").writelines(b_lines)
    # medium random inputs: 3 chroms, 400 x 400
    def rand_bed(path, n, seed, tag):
        r = np.random.default_rng(seed)
        ch = r.choice(["chr1", "chr2", "chrX"], size=n, p=[0.5, 0.3, 0.2])
        s = r.integers(0, 100000, size=n)
        e = s + r.integers(0, 600, size=n)
        with open(path, "w") as f:
            for i in range(n):
                f.write("%s\t%d\t%d\t%s%d\t0\t%s\n" % (ch[i], s[i], e[i], tag, i, "+-"[i % 2]))
    med_a, med_b = os.path.join(d, "med_a.bed"), os.path.join(d, "med_b.bed")
    rand_bed(med_a, 400, 3002, "p")
    rand_bed(med_b, 400, 3003, "q")

    exp = {}
    for tag, (fa, fb) in dict(small=(small_a, small_b), med=(med_a, med_b)).items():
        for flags in ([], ["-b"], ["-v"], ["-b", "-v"], ["-m", "5"], ["--mincols=50"], ["-m", "5", "-v"]):
            exp["bed_intersect %s %s" % (tag, " ".join(flags))] = run_script("bed_intersect.py", flags + [fa, fb])
        exp["bed_intersect_basewise %s" % tag] = run_script("bed_intersect_basewise.py", [fa, fb])
        exp["bed_coverage %s a" % tag] = run_script("bed_coverage.py", [fa])
        exp["bed_coverage %s ab" % tag] = run_script("bed_coverage.py", [fa, fb])
    exp["bed_coverage small stdin"] = run_script("bed_coverage.py", [], stdin="".join(b_lines))
    exp["interval_join small"] = run_script("interval_join.py", [ja, jb])
    exp["interval_join med"] = run_script("interval_join.py", [med_a, med_b])
    # error behaviour pinned too
    bad = os.path.join(d, "bad_reversed.bed")
    open(bad, "w").write("chr1\t50\t40\tx\n")
    exp["bed_coverage bad_reversed"] = run_script("bed_coverage.py", [bad])
    big = os.path.join(d, "bad_toolarge.bed")
    open(big, "w").write("chr1\t536870911\t536870913\tx\n")
    exp["bed_coverage bad_toolarge"] = run_script("bed_coverage.py", [big])
    exp["bed_intersect bad_toolarge_query"] = run_script("bed_intersect.py", [big, small_b])

    # cfg 1: chr1 10k x 10k from seeds 101/102 -- inputs are regenerated by the test, outputs hashed
    import tempfile
    (ts, te), (qs, qe) = synth.cfg1()
    with tempfile.TemporaryDirectory() as td:
        fa, fb = os.path.join(td, "q.bed"), os.path.join(td, "t.bed")
        open(fa, "w").writelines(synth.bed_lines("chr1", qs, qe, "q"))
        open(fb, "w").writelines(synth.bed_lines("chr1", ts, te, "t"))
        cfg1 = {}
        for name, args in (("bed_intersect", [fa, fb]), ("bed_intersect -b", ["-b", fa, fb]), ("bed_intersect -m 500", ["-m", "500", fa, fb]),
                           ("bed_intersect_basewise", [fa, fb]), ("bed_coverage", [fb]), ("interval_join", [fa, fb])):
            r = run_script(name.split()[0] + ".py", args)
            cfg1[name] = dict(sha256=hashlib.sha256(r["stdout"].encode()).hexdigest(), nbytes=len(r["stdout"]),
                              head=r["stdout"][:200], returncode=r["returncode"])
    dump("cli/expected.json", dict(source="reference scripts/*.py run with the reference build", cases=exp, cfg1=cfg1))


def gen_cli_crlf():
    """Carriage returns: the reference reads its inputs in text mode, so '\\r\\n' reaches it as '\\n' (and a lone '\\r' ends a
    line).  CRLF copies of the small pair, plus a file with a form feed and a vertical tab inside a field (no line break
    for file iteration, one for str.splitlines)."""
    d = os.path.join(GOLD, "cli")
    exp = {}
    for name in ("small_a", "small_b"):
        text = open(os.path.join(d, name + ".bed")).read()
        open(os.path.join(d, name + "_crlf.bed"), "w", newline="").write(text.replace("\n", "\r\n"))
    odd = os.path.join(d, "odd_separators.bed")
    open(odd, "w", newline="").write("chr1\t10\t20\tx\x0cy\t0\t+\nchr1\t15\t40\tv\x0bw\t0\t-\rchr2\t5\t9\tz\t0\t+\n")
    a, b = os.path.join(d, "small_a_crlf.bed"), os.path.join(d, "small_b_crlf.bed")
    la, lb = os.path.join(d, "small_a.bed"), os.path.join(d, "small_b.bed")
    for flags in ([], ["-b"], ["-v"], ["-m", "5"]):
        exp["bed_intersect crlf_query %s" % " ".join(flags)] = run_script("bed_intersect.py", flags + [a, lb])
        exp["bed_intersect crlf_both %s" % " ".join(flags)] = run_script("bed_intersect.py", flags + [a, b])
    exp["bed_intersect odd_query"] = run_script("bed_intersect.py", [odd, lb])
    exp["bed_intersect_basewise crlf"] = run_script("bed_intersect_basewise.py", [a, b])
    exp["bed_coverage crlf"] = run_script("bed_coverage.py", [a, b])
    exp["bed_coverage odd"] = run_script("bed_coverage.py", [odd])
    dump("cli/expected_crlf.json", dict(source="reference scripts/*.py run with the reference build", cases=exp))


def gen_cli_siblings():
    """the sibling scripts that use only the hot-path API."""
    d = os.path.join(GOLD, "cli")
    small_a, small_b = os.path.join(d, "small_a.bed"), os.path.join(d, "small_b.bed")
    med_a, med_b = os.path.join(d, "med_a.bed"), os.path.join(d, "med_b.bed")
    lens = os.path.join(d, "chrom.len")
    open(lens, "w").write("chr1\t1000\nchr2\t400\nchr5\t77\nchrX\t100500\nchr3\t500\n")
    track = os.path.join(d, "track_offset.bed")
    open(track, "w").write("browser position chr1:1-100\ntrack name=x offset=100\nchr1\t10\t20\nchr1\t15\t30\n# c\n"
                           "track name=y offset=1000\nchr2\t0\t5\nchr1\t5\t9\nchr2\t3\t8\n")
    plain = os.path.join(d, "plain_a.bed")  # no comment/blank lines: some siblings do not skip them
    open(plain, "w").writelines([l for l in open(small_a) if not l.startswith("#") and l.strip()])
    exp = {}
    for tag, (fa, fb) in dict(small=(small_a, small_b), med=(med_a, med_b)).items():
        exp["bed_subtract_basewise %s" % tag] = run_script("bed_subtract_basewise.py", [fa, fb])
        exp["bed_subtract_basewise %s rev" % tag] = run_script("bed_subtract_basewise.py", [fb, fa])
        exp["bed_complement %s" % tag] = run_script("bed_complement.py", [fa, lens])
        exp["bed_merge_overlapping %s" % tag] = run_script("bed_merge_overlapping.py", [fa, fb])
        exp["bed_diff_basewise_summary %s" % tag] = run_script("bed_diff_basewise_summary.py", [fa, fb])
    exp["bed_merge_overlapping track"] = run_script("bed_merge_overlapping.py", [track])
    exp["bed_merge_overlapping stdin"] = run_script("bed_merge_overlapping.py", [], stdin=open(small_b).read())
    for tag, fa, fb in (("small", plain, small_b), ("med", med_a, med_b)):
        exp["bed_coverage_by_interval %s" % tag] = run_script("bed_coverage_by_interval.py", [fa, fb])
        exp["bed_coverage_by_interval %s mask" % tag] = run_script("bed_coverage_by_interval.py", [fa, fb, fa if tag == "med" else small_b])
        exp["bed_count_overlapping %s" % tag] = run_script("bed_count_overlapping.py", [fa, fb])
        exp["bed_count_by_interval %s" % tag] = run_script("bed_count_by_interval.py", [fa, fb])
        exp["interval_count_intersections %s" % tag] = run_script("interval_count_intersections.py", [fb, fa])
    dump("cli/expected_siblings.json", dict(source="reference scripts/*.py run with the reference build", cases=exp))


# --------------------------------------------------------------------------- #
# 4. Scale points of cfg 2 through the real treap (hashes only)
# --------------------------------------------------------------------------- #
def scale_point(n_targets, n_queries_total, stride):
    (ts, te), _ = synth.cfg2(n_targets, 1)
    rngq = synth.uniform_intervals(n_queries_total, 202)
    qs, qe = rngq[0][::stride], rngq[1][::stride]
    tree = ri.IntervalTree()
    ins = tree.insert
    for s, e in zip(ts.tolist(), te.tolist()):
        ins(s, e, None)
    find = tree.find
    counts = np.fromiter((len(find(a, b)) for a, b in zip(qs.tolist(), qe.tolist())), dtype=np.int32, count=len(qs))
    return dict(n_targets=n_targets, n_queries_total=n_queries_total, stride=stride, n_queries=len(qs),
                counts_sha256=sha(counts), total=int(counts.sum(dtype=np.int64)), first16=counts[:16].tolist())


def _ref_tree(ts, te):
    tree = ri.IntervalTree()
    ins = tree.insert
    for i, (s, e) in enumerate(zip(ts.tolist(), te.tolist())):
        ins(s, e, i)
    return tree


def genome_point(stride=100):
    """BASELINE configs[3] (synth.cfg4): per chromosome the reference treap over that chromosome's targets, queried with
    every stride-th of its queries."""
    out = {}
    for chrom in synth.HG19_SIZES:
        (ts, te), (qs, qe) = synth.cfg4_chrom(chrom)
        qs, qe = qs[::stride], qe[::stride]
        find = _ref_tree(ts, te).find
        counts = np.fromiter((len(find(a, b)) for a, b in zip(qs.tolist(), qe.tolist())), dtype=np.int32, count=len(qs))
        out[chrom] = dict(n_targets=len(ts), n_queries_total=len(synth.cfg4_chrom(chrom)[1][0]), n_queries=len(qs), counts_sha256=sha(counts),
                          total=int(counts.sum(dtype=np.int64)))
        print(chrom, out[chrom], flush=True)
    return dict(stride=stride, workload="synth.cfg4_chrom(chrom): targets seed (401,i), queries seed (402,i); queries[::stride]", chroms=out)


def join_point(n_targets=50_000_000, n_queries=50_000_000, stride=500):
    """BASELINE configs[4] (synth.cfg5) at full target count: the hit LISTS (payload = insertion index, the reference's
    order) of every stride-th query against the 50M-target reference treap (about 10 GB of Python objects)."""
    import time

    (ts, te), (qs, qe) = synth.cfg5(n_targets, n_queries)
    qs, qe = qs[::stride], qe[::stride]
    t0 = time.perf_counter()
    tree = _ref_tree(ts, te)
    t_ins = time.perf_counter() - t0
    find = tree.find
    t0 = time.perf_counter()
    res = [find(a, b) for a, b in zip(qs.tolist(), qe.tolist())]
    t_find = time.perf_counter() - t0
    counts = np.array([len(r) for r in res], dtype=np.int32)
    hits = np.array([x for r in res for x in r], dtype=np.int32)
    return dict(n_targets=n_targets, n_queries_total=n_queries, stride=stride, n_queries=len(qs), counts_sha256=sha(counts), hits_sha256=sha(hits),
                total=int(counts.sum(dtype=np.int64)), first_hits=hits[:16].tolist(), reference_insert_s=round(t_ins, 1), reference_find_s=round(t_find, 3))


def calibration_point(n_targets=10_000_000, n_queries_total=100_000_000, stride=100):
    """SURVEY 8(d) / BASELINE.md 4 step 1: the reference's IntervalTree.find and the C restatement (oracle/ivtree.c) timed
    on the same machine and the same inputs -- the 10M-target treap, the 1M-query subsample of cfg 2; single thread,
    best of 3 for the queries.  bench.py turns its port timing on the GPU box into "x reference" with this ratio."""
    import platform
    import time

    sys.path.insert(0, ROOT)
    from oracle import oracle as O

    (ts, te), _ = synth.cfg2(n_targets, 1)
    q = synth.uniform_intervals(n_queries_total, 202)
    qs, qe = q[0][::stride].copy(), q[1][::stride].copy()
    t0 = time.perf_counter()
    tree = _ref_tree(ts, te)
    ref_ins = time.perf_counter() - t0
    find = tree.find
    ql, el = qs.tolist(), qe.tolist()
    ref_find, counts = None, None
    for _ in range(3):
        t0 = time.perf_counter()
        counts = np.fromiter((len(find(a, b)) for a, b in zip(ql, el)), dtype=np.int32, count=len(ql))
        dt = time.perf_counter() - t0
        ref_find = dt if ref_find is None or dt < ref_find else ref_find
    del tree
    t0 = time.perf_counter()
    ot = O.OracleIntervalTree()
    ot.insert_many_arrays(ts, te)
    port_ins = time.perf_counter() - t0
    port_find = None
    for _ in range(3):
        t0 = time.perf_counter()
        pc, _ = ot.count_batch(qs, qe)
        dt = time.perf_counter() - t0
        port_find = dt if port_find is None or dt < port_find else port_find
    assert np.array_equal(pc, counts)
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        cpu = platform.processor()
    return dict(n_targets=n_targets, n_queries=len(qs), stride=stride, cpu=cpu, threads=1,
                reference_insert_s=round(ref_ins, 2), reference_find_s=round(ref_find, 3), reference_mq_per_s=round(len(qs) / ref_find / 1e6, 5),
                port_insert_s=round(port_ins, 2), port_find_s=round(port_find, 3), port_mq_per_s=round(len(qs) / port_find / 1e6, 5),
                port_over_reference=round(ref_find / port_find, 3), counts_sha256=sha(counts),
                note="reference = len(IntervalTree.find(qs, qe)) per query through the Cython extension (a Python-level loop, as "
                     "its callers use it); port = oracle/ivtree.c count_batch; same arrays, same machine, one thread")


def bitsets_genome_point():
    """BASELINE configs[2] (synth.genome_ranges(1_500_000, 301 / 302)): two hg19-sized dicts of BinnedBitSets through the
    real bx.bitset -- per chromosome popcount(A), popcount(B), popcount(A & B), popcount(A | B), the number of runs of
    A & B and the sha256 of its run list (int64 starts then int64 ends, from next_set / next_clear as bed_intersect_basewise
    walks them, scripts/bed_intersect_basewise.py:39-51)."""
    ra, rb_ = synth.genome_ranges(1_500_000, 301), synth.genome_ranges(1_500_000, 302)
    out = {}
    for chrom, size in synth.HG19_SIZES.items():
        sets = []
        for r in (ra, rb_, ra):
            bs = rb.BinnedBitSet(size)
            sr = bs.set_range
            for s, n in zip(r[chrom][0].tolist(), r[chrom][1].tolist()):
                sr(s, n)
            sets.append(bs)
        a, b, a2 = sets
        ca, cb = a.count_range(0, size), b.count_range(0, size)
        a2.ior(b)
        c_or = a2.count_range(0, size)
        a.iand(b)
        c_and = a.count_range(0, size)
        starts, ends = [], []
        end = 0
        while True:
            start = a.next_set(end)
            if start == size:
                break
            end = a.next_clear(start)
            starts.append(start), ends.append(end)
        runs = np.concatenate([np.array(starts, dtype=np.int64), np.array(ends, dtype=np.int64)])
        out[chrom] = dict(size=size, pop_a=ca, pop_b=cb, pop_and=c_and, pop_or=c_or, n_runs=len(starts), runs_sha256=sha(runs))
        print(chrom, out[chrom], flush=True)
    return dict(workload="synth.genome_ranges(1_500_000, 301) / (.., 302): one BinnedBitSet(size) per hg19 chromosome and set",
                source="bx.bitset.BinnedBitSet (reference 0.14.0): set_range, ior, iand, count_range, next_set / next_clear", chroms=out)


def bitsets_genome_default_point():
    """BASELINE configs[2] with the size the reference's own builders give every chromosome when no `lens` is passed
    (lib/bx/bitset_builders.py:31-45 -> BinnedBitSet() = MAX = 512 Mi bits, bitset.pyx:196-203): a different bin_size
    (float32, binBits.c:36) and so different ALL_ONE arithmetic after invert().  Per chromosome through the real bx.bitset:
    bin_size, popcounts of A, B, A & B, A | B over the whole set, the number of runs of A & B with the sha256 of its run list,
    and -- on the INVERTED A & B -- count_range over the whole set, over [0, chromosome length) and over eight windows that
    start and end inside bins (the first-bin / last-bin arithmetic of binBits.c:140-170)."""
    ra, rb_ = synth.genome_ranges(1_500_000, 301), synth.genome_ranges(1_500_000, 302)
    out = {}
    MAX = rb.MAX
    for chrom, size in synth.HG19_SIZES.items():
        sets = []
        for r in (ra, rb_, ra):
            bs = rb.BinnedBitSet()
            sr = bs.set_range
            for s, n in zip(r[chrom][0].tolist(), r[chrom][1].tolist()):
                sr(s, n)
            sets.append(bs)
        a, b, a2 = sets
        assert a.size == MAX
        ca, cb = a.count_range(0, MAX), b.count_range(0, MAX)
        a2.ior(b)
        c_or = a2.count_range(0, MAX)
        a.iand(b)
        c_and = a.count_range(0, MAX)
        starts, ends = [], []
        end = 0
        while True:
            start = a.next_set(end)
            if start == MAX:
                break
            end = a.next_clear(start)
            starts.append(start), ends.append(end)
        runs = np.concatenate([np.array(starts, dtype=np.int64), np.array(ends, dtype=np.int64)])
        a.invert()
        rngw = np.random.default_rng(size)
        ws = rngw.integers(0, MAX - 1, size=8)
        wn = np.minimum(rngw.integers(1, 50_000_000, size=8), MAX - ws)
        windows = [[int(s), int(n), int(a.count_range(int(s), int(n)))] for s, n in zip(ws, wn)]
        out[chrom] = dict(size=MAX, chrom_len=size, bin_size=int(a.bin_size), pop_a=ca, pop_b=cb, pop_and=c_and, pop_or=c_or, n_runs=len(starts),
                          runs_sha256=sha(runs), inverted_and_count_all=int(a.count_range(0, MAX)),
                          inverted_and_count_chrom=int(a.count_range(0, size)), inverted_and_windows=windows)
        print(chrom, out[chrom], flush=True)
    return dict(workload="synth.genome_ranges(1_500_000, 301) / (.., 302): one BinnedBitSet() (MAX = 512 Mi bits) per hg19 chromosome and set",
                source="bx.bitset.BinnedBitSet (reference 0.14.0): set_range, ior, iand, invert, count_range, next_set / next_clear", chroms=out)


def gen_extra(which):
    path = os.path.join(GOLD, "scale.json")
    doc = json.load(open(path))
    if which == "genome":
        doc["cfg4_genome"] = genome_point()
    elif which == "join":
        doc["cfg5_join"] = join_point()
    elif which == "calibration":
        doc["calibration"] = calibration_point()
    elif which == "bitsets_genome":
        doc["cfg3_bitsets"] = bitsets_genome_point()
    elif which == "bitsets_genome_default":
        doc["cfg3_bitsets_default_max"] = bitsets_genome_default_point()
    # another generator may have rewritten the file meanwhile: merge on the freshest copy
    fresh = json.load(open(path))
    for k in ("cfg4_genome", "cfg5_join", "calibration", "cfg3_bitsets", "cfg3_bitsets_default_max"):
        if k in doc and (k == {"genome": "cfg4_genome", "join": "cfg5_join", "calibration": "calibration", "bitsets_genome": "cfg3_bitsets",
                               "bitsets_genome_default": "cfg3_bitsets_default_max"}[which]):
            fresh[k] = doc[k]
    dump("scale.json", fresh)


def gen_scale(full):
    path = os.path.join(GOLD, "scale.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    pts = dict(old.get("points", {}))
    pts["1M x 200k"] = scale_point(1_000_000, 2_000_000, 10)
    if full:
        pts["10M x 1M (cfg2 subsample)"] = scale_point(10_000_000, 100_000_000, 100)
    dump("scale.json", dict(source="bx.intervals.intersection.IntervalTree.find (reference 0.14.0), len() of each result",
                            workload="synth.cfg2 targets seed 201; queries = synth.uniform_intervals(n_queries_total, 202)[::stride]",
                            points=pts))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    warnings.simplefilter("ignore")
    os.makedirs(GOLD, exist_ok=True)
    todo = a.only.split(",") if a.only else ["trees", "bitsets", "cli", "scale"]
    if "trees" in todo:
        gen_trees()
    if "bitsets" in todo:
        gen_bitsets()
    if "cli" in todo:
        gen_cli()
        gen_cli_siblings()
    if "siblings" in todo:
        gen_cli_siblings()
    if "crlf" in todo and a.only:
        gen_cli_crlf()
    if "scale" in todo:
        gen_scale(a.scale)
    for which in ("genome", "join", "calibration", "bitsets_genome", "bitsets_genome_default"):  # --only genome | join | calibration | bitsets_genome: long-running extras of scale.json
        if which in todo and a.only:
            gen_extra(which)
