"""
GPU tests of the CLI counterparts (bxmi.cli.*) against stdout captured from the
reference's own scripts (tests/golden/cli/expected.json, made by oracle/gen_golden.py),
plus the per-call drop-in API driven in the call pattern those scripts use.
Byte-identical stdout, same exit status, same final exception line.
"""
import hashlib
import io
import os
import subprocess
import sys

import numpy as np
import pytest

from bxmi import synth
from conftest import GOLDEN, PKG, ROOT

pytestmark = pytest.mark.gpu
CLI = os.path.join(GOLDEN, "cli")


def run_cli(module, args, stdin=None):
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + ROOT, PYTHONWARNINGS="ignore")
    p = subprocess.run([sys.executable, "-m", "bxmi.cli." + module] + args, input=stdin, capture_output=True, text=True, env=env)
    tail = p.stderr.strip().splitlines()[-1:] if p.returncode else []
    return dict(stdout=p.stdout, returncode=p.returncode, stderr_tail=tail)


def files(tag):
    return os.path.join(CLI, tag + "_a.bed"), os.path.join(CLI, tag + "_b.bed")


def check(got, want, key):
    assert got["returncode"] == want["returncode"], (key, got)
    assert got["stdout"] == want["stdout"], key
    assert got["stderr_tail"] == want["stderr_tail"], key


@pytest.mark.parametrize("tag", ["small", "med"])
def test_bed_intersect_flags(golden_cli, tag):
    fa, fb = files(tag)
    for flags in ([], ["-b"], ["-v"], ["-b", "-v"], ["-m", "5"], ["--mincols=50"], ["-m", "5", "-v"]):
        key = "bed_intersect %s %s" % (tag, " ".join(flags))
        check(run_cli("bed_intersect", flags + [fa, fb]), golden_cli["cases"][key], key)


@pytest.mark.parametrize("tag", ["small", "med"])
def test_basewise_and_coverage(golden_cli, tag):
    fa, fb = files(tag)
    check(run_cli("bed_intersect_basewise", [fa, fb]), golden_cli["cases"]["bed_intersect_basewise %s" % tag], tag)
    check(run_cli("bed_coverage", [fa]), golden_cli["cases"]["bed_coverage %s a" % tag], tag)
    check(run_cli("bed_coverage", [fa, fb]), golden_cli["cases"]["bed_coverage %s ab" % tag], tag)


def test_carriage_returns_and_odd_separators():
    """Inputs with '\\r\\n' line ends (query only, and both files), a lone '\\r' as a line end, and a form feed / vertical tab
    inside a field: the reference sees them through text-mode universal newlines, and so must the fast ingest path
    (which used to hand back '\\r\\n' lines and to split fields at \\x0b / \\x0c)."""
    import json

    want = json.load(open(os.path.join(CLI, "expected_crlf.json")))["cases"]
    a, b = os.path.join(CLI, "small_a_crlf.bed"), os.path.join(CLI, "small_b_crlf.bed")
    lb, odd = os.path.join(CLI, "small_b.bed"), os.path.join(CLI, "odd_separators.bed")
    assert b"\r\n" in open(a, "rb").read() and b"\x0c" in open(odd, "rb").read()
    for flags in ([], ["-b"], ["-v"], ["-m", "5"]):
        for tag, fb in (("crlf_query", lb), ("crlf_both", b)):
            key = "bed_intersect %s %s" % (tag, " ".join(flags))
            check(run_cli("bed_intersect", flags + [a, fb]), want[key], key)
    check(run_cli("bed_intersect", [odd, lb]), want["bed_intersect odd_query"], "odd_query")
    check(run_cli("bed_intersect_basewise", [a, b]), want["bed_intersect_basewise crlf"], "basewise crlf")
    check(run_cli("bed_coverage", [a, b]), want["bed_coverage crlf"], "coverage crlf")
    check(run_cli("bed_coverage", [odd]), want["bed_coverage odd"], "coverage odd")


def test_coverage_stdin_and_errors(golden_cli):
    stdin = open(os.path.join(CLI, "small_b.bed")).read()
    check(run_cli("bed_coverage", [], stdin=stdin), golden_cli["cases"]["bed_coverage small stdin"], "stdin")
    for name, key in (("bad_reversed", "bed_coverage bad_reversed"), ("bad_toolarge", "bed_coverage bad_toolarge")):
        check(run_cli("bed_coverage", [os.path.join(CLI, name + ".bed")]), golden_cli["cases"][key], key)
    key = "bed_intersect bad_toolarge_query"
    check(run_cli("bed_intersect", [os.path.join(CLI, "bad_toolarge.bed"), os.path.join(CLI, "small_b.bed")]), golden_cli["cases"][key], key)


def test_interval_join(golden_cli):
    check(run_cli("interval_join", [os.path.join(CLI, "join_a.bed"), os.path.join(CLI, "join_b.bed")]),
          golden_cli["cases"]["interval_join small"], "small")
    fa, fb = files("med")
    check(run_cli("interval_join", [fa, fb]), golden_cli["cases"]["interval_join med"], "med")


def test_cfg1_10k_x_10k(golden_cli, tmp_path):
    """BASELINE configs[0]: chr1 10k x 10k synthetic BED, every script, output hashes from the reference run."""
    (ts, te), (qs, qe) = synth.cfg1()
    fa, fb = str(tmp_path / "q.bed"), str(tmp_path / "t.bed")
    open(fa, "w").writelines(synth.bed_lines("chr1", qs, qe, "q"))
    open(fb, "w").writelines(synth.bed_lines("chr1", ts, te, "t"))
    runs = {
        "bed_intersect": ("bed_intersect", [fa, fb]), "bed_intersect -b": ("bed_intersect", ["-b", fa, fb]),
        "bed_intersect -m 500": ("bed_intersect", ["-m", "500", fa, fb]), "bed_intersect_basewise": ("bed_intersect_basewise", [fa, fb]),
        "bed_coverage": ("bed_coverage", [fb]), "interval_join": ("interval_join", [fa, fb]),
    }
    for key, (mod, args) in runs.items():
        want = golden_cli["cfg1"][key]
        got = run_cli(mod, args)
        assert got["returncode"] == want["returncode"] == 0, key
        assert len(got["stdout"]) == want["nbytes"] and got["stdout"][:200] == want["head"], key
        assert hashlib.sha256(got["stdout"].encode()).hexdigest() == want["sha256"], key


# ---- sibling scripts that use only the hot-path API ----------------------
@pytest.fixture(scope="module")
def golden_siblings():
    from conftest import load_golden

    return load_golden("cli/expected_siblings.json")["cases"]


def check_sibling(golden_siblings, key, module, args, stdin=None):
    want = dict(golden_siblings[key])
    # the goldens were captured with files under /root/repo/tests/golden/cli; two scripts echo the file names
    want["stdout"] = want["stdout"].replace("/root/repo/tests/golden/cli", CLI)
    check(run_cli(module, args, stdin=stdin), want, key)


@pytest.mark.parametrize("tag", ["small", "med"])
def test_sibling_bitset_scripts(golden_siblings, tag):
    fa, fb = files(tag)
    lens = os.path.join(CLI, "chrom.len")
    check_sibling(golden_siblings, "bed_subtract_basewise %s" % tag, "bed_subtract_basewise", [fa, fb])
    check_sibling(golden_siblings, "bed_subtract_basewise %s rev" % tag, "bed_subtract_basewise", [fb, fa])
    check_sibling(golden_siblings, "bed_complement %s" % tag, "bed_complement", [fa, lens])
    check_sibling(golden_siblings, "bed_merge_overlapping %s" % tag, "bed_merge_overlapping", [fa, fb])
    check_sibling(golden_siblings, "bed_diff_basewise_summary %s" % tag, "bed_diff_basewise_summary", [fa, fb])


def test_sibling_merge_track_offsets_and_stdin(golden_siblings):
    check_sibling(golden_siblings, "bed_merge_overlapping track", "bed_merge_overlapping", [os.path.join(CLI, "track_offset.bed")])
    check_sibling(golden_siblings, "bed_merge_overlapping stdin", "bed_merge_overlapping", [], stdin=open(os.path.join(CLI, "small_b.bed")).read())


@pytest.mark.parametrize("tag", ["small", "med"])
def test_sibling_interval_scripts(golden_siblings, tag):
    fa, fb = (os.path.join(CLI, "plain_a.bed"), os.path.join(CLI, "small_b.bed")) if tag == "small" else files("med")
    check_sibling(golden_siblings, "bed_coverage_by_interval %s" % tag, "bed_coverage_by_interval", [fa, fb])
    check_sibling(golden_siblings, "bed_coverage_by_interval %s mask" % tag, "bed_coverage_by_interval",
                  [fa, fb, fa if tag == "med" else os.path.join(CLI, "small_b.bed")])
    check_sibling(golden_siblings, "bed_count_overlapping %s" % tag, "bed_count_overlapping", [fa, fb])
    check_sibling(golden_siblings, "bed_count_by_interval %s" % tag, "bed_count_by_interval", [fa, fb])
    check_sibling(golden_siblings, "interval_count_intersections %s" % tag, "interval_count_intersections", [fb, fa])


# ---- native BED ingest: bulk path == per-line path, byte for byte --------
def test_fast_ingest_equals_per_line_path(tmp_path):
    rng = np.random.default_rng(808)

    def make(path, n, tag, oddities):
        ch = rng.choice(["chr1", "chr2", "chr10", "chrUn_x"], size=n, p=[0.4, 0.3, 0.2, 0.1])
        s = rng.integers(0, 5_000_000, size=n)
        e = s + rng.integers(0, 900, size=n)
        with open(path, "w") as f:
            for i in range(n):
                if oddities and i % 5003 == 0:
                    f.write("# comment\n\n   \n")
                if oddities and i == n // 2:
                    f.write("chr2\t1_000\t2_000\tunderscores_are_valid_python_ints\n")  # parser hands over here
                f.write("%s\t%d\t%d\t%s%d\t0\t+\n" % (ch[i], s[i], e[i], tag, i))

    fa, fb = str(tmp_path / "a.bed"), str(tmp_path / "b.bed")
    make(fa, 60_000, "a", True)
    make(fb, 60_000, "b", True)
    slow_env = {"BXMI_NO_FASTPARSE": "1"}

    def run(module, args, extra=None):
        env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + ROOT, PYTHONWARNINGS="ignore", **(extra or {}))
        p = subprocess.run([sys.executable, "-m", "bxmi.cli." + module] + args, capture_output=True, text=True, env=env)
        return p.returncode, p.stdout, p.stderr.strip().splitlines()[-1:] if p.returncode else []

    for module, args in (("bed_intersect", [fa, fb]), ("bed_intersect", ["-v", fa, fb]), ("bed_intersect", ["-b", "-m", "300", fa, fb]),
                         ("bed_coverage", [fa, fb]), ("bed_intersect_basewise", [fa, fb]), ("bed_subtract_basewise", [fa, fb])):
        fast, slow = run(module, args), run(module, args, slow_env)
        assert fast == slow and fast[0] == 0 and len(fast[1]) > 0, (module, args)
    # an invalid row in the middle: same output before it, same exception
    bad = str(tmp_path / "bad.bed")
    lines = open(fa).read().splitlines(keepends=True)
    lines.insert(40_000, "chr1\t600000000\t600000010\tbeyond_MAX\n")
    open(bad, "w").writelines(lines)
    for module, args in (("bed_intersect", [bad, fb]), ("bed_coverage", [bad])):
        fast, slow = run(module, args), run(module, args, slow_env)
        assert fast == slow and fast[0] == 1 and "IndexError" in fast[2][0], (module, fast[2])


# ---- the per-call drop-in API, used the way the unmodified scripts use it -----------------
def per_line_bitsets(path):
    """One BinnedBitSet per chromosome, one set_range call per BED line (file order)."""
    import bx.bitset

    sets = {}
    for line in open(path):
        if line.startswith("#") or line.isspace():
            continue
        f = line.split()
        if f[0] not in sets:
            sets[f[0]] = bx.bitset.BinnedBitSet(bx.bitset.MAX)
        sets[f[0]].set_range(int(f[1]), int(f[2]) - int(f[1]))
    return sets


@pytest.mark.parametrize("tag", ["small", "med"])
def test_dropin_classes_per_call_pattern(golden_cli, tag):
    fa, fb = files(tag)
    # count_range per query line, echoing hits with the trailing-space print quirk
    sets = per_line_bitsets(fb)
    out = io.StringIO()
    for line in open(fa):
        if line.startswith("#") or line.isspace():
            continue
        f = line.split()
        s, e = int(f[1]), int(f[2])
        if f[0] in sets and sets[f[0]].count_range(s, e - s) >= 1:
            print(line, end=" ", file=out)
    assert out.getvalue() == golden_cli["cases"]["bed_intersect %s " % tag]["stdout"]
    # iand + the next_set / next_clear walk
    a, b = per_line_bitsets(fa), per_line_bitsets(fb)
    out = io.StringIO()
    for chrom in a:
        if chrom not in b:
            continue
        a[chrom].iand(b[chrom])
        bits, end = a[chrom], 0
        while True:
            start = bits.next_set(end)
            if start == bits.size:
                break
            end = bits.next_clear(start)
            print("%s\t%d\t%d" % (chrom, start, end), file=out)
    assert out.getvalue() == golden_cli["cases"]["bed_intersect_basewise %s" % tag]["stdout"]
    # coverage
    total = sum(s.count_range(0, s.size) for s in per_line_bitsets(fa).values())
    assert "%d\n" % total == golden_cli["cases"]["bed_coverage %s a" % tag]["stdout"]


def test_dropin_intersecter_join_pattern(golden_cli):
    """Per-chromosome Intersecter, add_interval per row, find per row -- interval_join's call pattern."""
    import bx.intervals

    class Row:
        def __init__(self, line):
            self.fields = line.rstrip("\r\n").split("\t")
            self.chrom, self.start, self.end = self.fields[0], int(self.fields[1]), int(self.fields[2])

        def __str__(self):
            return "\t".join(self.fields)

    fa, fb = files("med")
    trees = {}
    for line in open(fb):
        r = Row(line)
        if r.chrom not in trees:
            trees[r.chrom] = bx.intervals.Intersecter()
        trees[r.chrom].add_interval(r)
    out = io.StringIO()
    for line in open(fa):
        r = Row(line)
        if r.chrom in trees:
            for other in trees[r.chrom].find(r.start, r.end):
                print("\t".join([str(r), str(other)]), file=out)
    assert out.getvalue() == golden_cli["cases"]["interval_join med"]["stdout"]
