"""CPU-only tests: the C ABI library loads and exports what include/bxmi.h declares, and the
host-side logic (argument coercion, error text, sharding) behaves -- no compute calls."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def header_functions():
    txt = open(os.path.join(ROOT, "include", "bxmi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bxmi_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from bxmi import _ffi

    if not os.path.exists(_ffi.LIB_PATH):
        subprocess.check_call(["bash", os.path.join(ROOT, "bx-python_amd", "csrc", "build.sh")])
    lib = _ffi.load()
    declared = header_functions()
    assert len(declared) >= 45
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_ffi.EXPORTED) == declared, set(_ffi.EXPORTED) ^ set(declared)
    assert lib.bxmi_version() >= 100
    out = subprocess.check_output(["nm", "-D", "--defined-only", _ffi.LIB_PATH], text=True)
    exported = set(re.findall(r" T (bxmi_\w+)", out))
    assert set(declared) <= exported


def test_no_gpu_means_loud_failure():
    from bxmi import _ffi

    if _ffi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_ffi.BxmiError):
        from bxmi.intervals import IntervalIndex

        IntervalIndex()
    with pytest.raises(_ffi.BxmiError):
        import bx.bitset

        bx.bitset.BinnedBitSet(100)
    import bx.intervals

    t = bx.intervals.IntervalTree()
    assert t.find(1, 2) == []  # empty tree needs no device (intersection.pyx:404-405)
    with pytest.raises(_ffi.BxmiError):
        t.insert(1, 2, "x")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bx-python_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".sh")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("the oracle's", ""), os.path.join(dirpath, f)


def test_cint_coercion_matches_cython():
    from bx.bitset import _cint

    assert _cint(5) == 5 and _cint(2.9) == 2 and _cint(-2.9) == -2 and _cint(np.int64(7)) == 7 and _cint(True) == 1
    for bad in (2**31, -(2**31) - 1):
        with pytest.raises(OverflowError, match="value too large to convert to int"):
            _cint(bad)
    for bad in ("a", None, [1]):
        with pytest.raises(TypeError, match="an integer is required"):
            _cint(bad)


def test_range_validation_messages():
    from bxmi.bitset import _first_bad_range

    s = np.array([0, 5, 7], np.int32)
    assert _first_bad_range(100, s, np.array([1, 2, 3], np.int32)) == (-1, None)
    for starts, lens, msg in (
        ([0, -3], [1, 1], "BitSet index (-3) must be non-negative."),
        ([0, 100], [1, 0], "100 is larger than the size of this BitSet (100)."),
        ([0, 5], [1, -2], "Count (-2) must be non-negative."),
        ([0, 99], [1, 2], "End (101) is larger than the size of this BinnedBitSet (100)."),
    ):
        k, err = _first_bad_range(100, np.array(starts, np.int32), np.array(lens, np.int32))
        assert k == 1 and isinstance(err, IndexError) and str(err) == msg
    k, err = _first_bad_range(100, np.array([99], np.int32), np.array([2], np.int32), binned=False)
    assert str(err) == "End 101 is larger than the size of this BitSet (100)."


def test_interval_value_class():
    from bx.intervals.intersection import Interval

    i = Interval(3, 9, value={"a": 1}, chrom="c", strand="-")
    assert repr(i) == "Interval(3, 9, value={'a': 1})" and repr(Interval(3, 9)) == "Interval(3, 9)"
    assert (i.start, i.end, i.chrom, i.strand) == (3, 9, "c", "-")
    assert i < Interval(4, 5) and i == Interval(3, 9) and i != Interval(3, 10) and i >= Interval(3, 9)
    with pytest.raises(AssertionError, match="start must be less than end"):
        Interval(5, 3)


def test_lpt_sharding_hg19():
    from bxmi import shard, synth

    parts = shard.lpt_assign(synth.HG19_SIZES, 8)
    assert sorted(c for p in parts for c in p) == sorted(synth.HG19_SIZES)
    assert shard.balance(synth.HG19_SIZES, parts) < 1.05  # ~96 % balance
    assert shard.lpt_assign(synth.HG19_SIZES, 8) == parts  # deterministic
    assert shard.lpt_assign({"a": 1}, 3) == [["a"], [], []]
    blocks = [shard.query_block(10, r, 4) for r in range(4)]
    assert blocks == [(0, 3), (3, 6), (6, 8), (8, 10)]


WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "bx-python_amd"))
import numpy as np, torch, torch.distributed as dist
from bxmi import shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
rng = np.random.default_rng(3)
sizes = {"chr1": 5000, "chr2": 3000, "chr3": 2500, "chrX": 900, "chrM": 40}
tg, qr = {}, {}
for c, n in sizes.items():
    s = rng.integers(0, 100000, size=n).astype(np.int32); tg[c] = (s, (s + rng.integers(1, 300, size=n)).astype(np.int32))
    q = rng.integers(0, 100000, size=n // 2).astype(np.int32); qr[c] = (q, (q + rng.integers(1, 300, size=n // 2)).astype(np.int32))
def brute(ts, te, qs, qe):   # stand-in for the GPU engine: the exact predicate of intersection.pyx:185
    S, E = np.sort(ts), np.sort(te)
    counts = (np.searchsorted(S, qe, "left") - np.searchsorted(E, qs, "right")).astype(np.int32)
    return counts, int(counts.sum())
totals, mine = shard.count_genome(tg, qr, rank, world, counter=brute)
full = {c: brute(*tg[c], *qr[c])[1] for c in sizes}
assert totals == full, (rank, totals, full)
owned = shard.lpt_assign({c: len(tg[c][0]) + len(qr[c][0]) for c in sizes}, world)[rank]
assert sorted(mine) == sorted(owned)
lo, hi = shard.query_block(len(qr["chr1"][0]), rank, world)
part = brute(*tg["chr1"], qr["chr1"][0][lo:hi], qr["chr1"][1][lo:hi])[0]
whole = shard.gather_concat(part)
assert np.array_equal(whole, brute(*tg["chr1"], *qr["chr1"])[0])
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_count_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", str(script), ROOT]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert p.stdout.count("ok") == 2


@pytest.mark.skipif(not os.path.isdir("/root/reference/lib/bx"), reason="reference tree not mounted (build container only)")
def test_overlay_resolves_next_to_an_installed_bx_python():
    """PYTHONPATH=bx-python_amd:<bx-python>/lib: the hot-path modules and their batch-aware callers are ours (since round 5 also
    the concat operation and bx.bitset_utils), everything else (cookbook, sequence modules, ...) comes from bx-python."""
    code = (
        "import bx, bx.bitset, bx.intervals, bx.intervals.intersection, bx.bitset_builders, bx.intervals.io, bx.cookbook.doc_optparse\n"
        "import bx.intervals.operations.concat, bx.intervals.operations.intersect, bx.tabular.io, bx.cookbook.attribute\n"
        "print(bx.bitset.__file__); print(bx.intervals.intersection.__file__)\n"
        "print(bx.bitset_builders.__file__); print(bx.intervals.io.__file__)\n"
        "print(bx.bitset_builders.BinnedBitSet is bx.bitset.BinnedBitSet, bx.intervals.Intersecter is bx.intervals.intersection.IntervalTree)\n"
        "print(bx.cookbook.doc_optparse.__file__); print(bx.intervals.operations.concat.__file__); print(bx.cookbook.attribute.__file__)\n"
        "print(bx.intervals.operations.intersect.__file__); print(bx.tabular.io.__file__)\n"
        "print(bx.intervals.io.GenomicInterval.__mro__[1] is bx.tabular.io.TableRow)\n"
    )
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "bx-python_amd") + os.pathsep + "/root/reference/lib")
    out = subprocess.check_output([sys.executable, "-c", code], text=True, env=env).splitlines()
    assert out[0].endswith("bx-python_amd/bx/bitset.py") and out[1].endswith("bx-python_amd/bx/intervals/intersection.py")
    assert out[2].endswith("bx-python_amd/bx/bitset_builders.py") and out[3].endswith("bx-python_amd/bx/intervals/io.py")
    assert out[4] == "True True"
    assert out[5].startswith("/root/reference/lib/bx/") and out[7].startswith("/root/reference/lib/bx/"), out[5:8]
    assert out[6].endswith("bx-python_amd/bx/intervals/operations/concat.py"), out[6]
    assert out[8].endswith("bx-python_amd/bx/intervals/operations/intersect.py") and out[9].endswith("bx-python_amd/bx/tabular/io.py")
    assert out[10] == "True"


def _python_parse(text, chrom_col=0, start_col=1, end_col=2):
    """What the reference's per-line code accepts, restated: returns (rows, index of the first line the strict
    native parser must refuse or None)."""
    import re

    rows, stop = [], None
    plain_int = re.compile(r"[+-]?[0-9]{1,18}\Z")
    # the parser's notion of a line: bytes up to and including '\n' (a '\r' anywhere makes it stop: such text is left
    # to the per-line path, which sees it through universal newlines)
    for ln, line in enumerate(re.findall(r"[^\n]*\n|[^\n]+\Z", text)):
        body = line[:-1] if line.endswith("\n") else line
        if any(ord(ch) >= 0x80 for ch in body) or "\r" in body or "\x00" in body:
            stop = ln
            break
        if line.startswith("#") or line.isspace():
            continue
        f = line.split()
        need = max(chrom_col, start_col, end_col) + 1
        if len(f) < need or not plain_int.match(f[start_col]) or not plain_int.match(f[end_col]):
            stop = ln
            break
        rows.append((f[chrom_col], int(f[start_col]), int(f[end_col]), line))
    return rows, stop


def test_native_bed_parser_matches_per_line_semantics():
    """csrc/bedparse.cpp (host-only code in libbxmi.so): consumes exactly the plain lines, in order, and stops where
    the per-line Python path has to take over."""
    from bxmi import bedio

    rng = np.random.default_rng(5)
    body = []
    for i in range(3000):
        ch = "chr%d" % rng.integers(1, 5)
        s = int(rng.integers(0, 10**6))
        sep = ["\t", " ", "  \t "][int(rng.integers(0, 3))]
        body.append("%s%s%d%s%d%sname%d\n" % (ch, sep, s, sep, s + int(rng.integers(0, 500)), sep, i))
        if i % 97 == 0:
            body.append("# comment %d\n" % i)
        if i % 131 == 0:
            body.append(["\n", "   \n", "\t\n", "\x1c\x1d \n"][i % 4])
    cases = {
        "clean": "".join(body),
        "no trailing newline": "".join(body)[:-1],
        "signs and zeros": "chr1\t+5\t-3\nchr1\t0007\t09\n  chr2   1   2   \n",
        "leading space then hash is data": " #x\t1\t2\n",
        "underscore int": "chr1\t1\t2\nchr1\t1_0\t20\nchr1\t3\t4\n",
        "float": "chr1\t1\t2\nchr1\t1.5\t2\n",
        "too few columns": "chr1\t1\t2\nchr1\t5\n",
        "crlf": "chr1\t1\t2\r\nchr1\t3\t4\r\n",
        "form feed in a field, lone cr": "chr1\t1\t2\nchr1\t3\t4\tx\x0cy\nchr1\t5\t6\rchr1\t7\t8\n",
        "non ascii": "chr1\t1\t2\nchré\t3\t4\n",
        "huge literal": "chr1\t1\t2\nchr1\t1234567890123456789012\t5\n",
        "empty": "",
        "only comments": "# a\n#b\n\n",
    }
    for name, text in cases.items():
        data = text.encode("utf-8")
        want, stop = _python_parse(text)
        bed = bedio.ParsedBed(data)
        got = [(bed.names[c], int(s), int(e), data[int(o):int(o) + int(n)].decode("utf-8"))
               for c, s, e, o, n in zip(bed.chrom.tolist(), bed.start.tolist(), bed.end.tolist(), bed.line_off.tolist(), bed.line_len.tolist())]
        assert got == want, name
        # what is handed back is what iterating a text-mode file would yield from there on: universal newlines,
        # breaks at '\n' / '\r' / '\r\n' only
        import io

        raw = re.findall(r"[^\n]*\n|[^\n]+\Z", text)
        rest = list(io.StringIO("".join(raw[stop:]), newline=None)) if stop is not None else []
        assert bed.rest_lines() == rest, name
        # chromosome ids are handed out in first-appearance order
        seen = []
        for ch, _, _, _ in want:
            if ch not in seen:
                seen.append(ch)
        assert bed.names == seen, name
        bed.close()


def test_interval_readers_follow_the_reference_docstring_examples():
    """The known-answer examples of lib/bx/intervals/io.py:111-137,220-229 (reader protocol, field write-back,
    NiceReaderWrapper bookkeeping), on bxmi.genomic -- host-only text handling, no GPU involved."""
    from bxmi.genomic import Comment, GenomicInterval, GenomicIntervalReader, Header, NiceReaderWrapper

    lines = ["#chrom\tname\tstart\tend\textra", "chr1\tfoo\t1\t100\txxx", "chr2\tbar\t20\t300\txxx", "#I am a comment",
             "chr2\tbar\t20\t300\txxx"]
    r = GenomicIntervalReader(lines, start_col=2, end_col=3)
    elements = list(r)
    assert isinstance(elements[0], Header) and str(elements[0]) == "#chrom\tname\tstart\tend\textra"
    assert isinstance(elements[1], GenomicInterval) and (elements[1].start, elements[1].end) == (1, 100)
    assert str(elements[1]) == "chr1\tfoo\t1\t100\txxx"
    elements[1].start = 30
    assert (elements[1].start, elements[1].end) == (30, 100) and str(elements[1]) == "chr1\tfoo\t30\t100\txxx"
    assert isinstance(elements[2], GenomicInterval) and isinstance(elements[3], Comment) and isinstance(elements[4], GenomicInterval)
    assert elements[1]["name"] == "foo" and elements[1][0] == "chr1"

    n = NiceReaderWrapper(lines + ["chr1\tbaz\tx\t5", "chr1\tq\t9\t3"], start_col=2, end_col=3)
    assert isinstance(next(n), Header) and n.current_line == lines[0]
    assert len(list(n)) == 4
    assert n.skipped == 2 and [t[0] for t in n.skipped_lines] == [6, 7]
    assert n.skipped_lines[0][2] == "Could not parse start_col: invalid literal for int() with base 10: 'x' on line 6, integer expected"
    assert n.skipped_lines[1][2] == "Start is greater than End. Interval length is < 1. on line 7"
    # construction normalises the fields it parsed (whitespace, sign, strand '.')
    row = GenomicInterval(None, [" chr7 ", "+5", "9", "x", "0", "."], 0, 1, 2, 5, "-")
    assert row.fields == ["chr7", "5", "9", "x", "0", "-"] and row.copy().fields == row.fields


def test_native_table_parse_equals_per_line_readers(monkeypatch):
    """bxmi.tabio (csrc/bedparse.cpp table mode) under the three readers: items, line numbers, raw lines, skip bookkeeping
    and escaping errors are those of the per-line code, on files where the parser gets far (clean rows), nowhere (a header
    it must leave alone, CRLF) and part of the way (odd rows in the middle: signs, spaces, leading zeros, '.', bad strands,
    too few fields, start > end, non-ASCII)."""
    from bxmi import genomic, tabio

    rng = np.random.default_rng(5)
    clean = ["chr%d\t%d\t%d\tn%d\t0\t%s\n" % (rng.integers(1, 4), a, a + rng.integers(0, 500), i, "+-"[i % 2])
             for i, a in enumerate(rng.integers(0, 10**6, size=300).tolist())]
    odd = ["chr1\t+5\t9\tx\t0\t+\n", "chr1\t 5\t9\n", "chr1\t007\t9\n", "chr1\t5\t9\tx\t0\t.\n", "chr1\t5\t9\tx\t0\t*\n", "chr1\t5\n",
           "chr1\t9\t5\n", " chr1 \t5\t9\n", "chr\u00e9\t5\t9\n", "chr1\t-0\t9\n", "chr1\t5\t9\tx\t0\t+\r\n", "\n", "# note\n", "track name=x\n",
           "chr1\t1_0\t20\n", "chr2\t-7\t-3\n", "chr1\t5\t9\tx\t0\t-\n", "chr\x001\t5\t9\n", "chr1\t5\t9\tna\x00me\t0\t+\n"]
    files = {
        "clean": clean,
        "header then clean": ["#chrom\tstart\tend\n"] + clean[:50],
        "comment later": clean[:20] + ["# c\n", "\n", "track t\n"] + clean[20:40],
        "odd rows": [x for pair in zip(clean[:len(odd)], odd) for x in pair] + clean[100:120],
        "crlf everywhere": [x[:-1] + "\r\n" for x in clean[:30]],
        "no line ends": [x.rstrip("\n") for x in clean[:10]],
        "empty": [],
    }

    def run(cls, lines, **kw):
        r = cls(list(lines), **kw) if cls is not genomic.BitsetSafeReaderWrapper else cls(genomic.GenomicIntervalReader(list(lines)), lens={"chr1": 600000})
        out, err = [], None
        try:
            for x in r:
                out.append((type(x).__name__, str(x), r.linenum, getattr(r, "current_line", None), getattr(x, "strand", None),
                            getattr(x, "start", None), getattr(x, "end", None), getattr(x, "chrom", None)))
        except Exception as e:  # noqa: BLE001 -- whatever escapes must be the same thing
            err = (type(e).__name__, str(e))
        return out, err, getattr(r, "skipped", None), list(getattr(r, "skipped_lines", [])), list(getattr(r, "skip_log", [])), getattr(r, "delivered", None)

    used = 0
    for name, lines in files.items():
        for cls, kw in ((genomic.GenomicIntervalReader, {}), (genomic.GenomicIntervalReader, {"fix_strand": True, "strand_col": 5}),
                        (genomic.NiceReaderWrapper, {}), (genomic.NiceReaderWrapper, {"return_comments": False}), (genomic.BitsetSafeReaderWrapper, {})):
            monkeypatch.setenv("BXMI_NO_FASTPARSE", "1")
            want = run(cls, lines, **kw)
            monkeypatch.delenv("BXMI_NO_FASTPARSE")
            got = run(cls, lines, **kw)
            assert got == want, (name, cls.__name__, kw)
        t = tabio.parse_input(list(lines), 0, 1, 2, 5, ["#", "track "])
        used += 0 if t is None else t.n
    assert used > 300  # the parser did take most of the clean lines: the comparison above was not per-line against per-line


def test_small_builders_raise_what_the_reference_raises_before_touching_the_device():
    """bitset_builders.py:107-169: the failing cases of tests/golden/builders_quicksect.json are decided on the
    host (the reference's own exceptions, made by oracle/gen_golden_extra.py), so they are checked here too."""
    import json

    import bx.bitset_builders as bb

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "builders_quicksect.json")) as f:
        doc = json.load(f)
    failing = [c for c in doc["builders"] if "error" in c["want"]]
    assert len(failing) >= 10
    for case in failing:
        a = case["args"]
        with pytest.raises(Exception) as ei:
            if case["fn"] == "from_list":
                bb.binned_bitsets_from_list(a["rows"])
            elif case["fn"] == "proximity":
                bb.binned_bitsets_proximity(iter(a["lines"]), **a["kw"])
            else:
                bb.binned_bitsets_by_chrom(iter(a["lines"]), a["chrom"], **a["kw"])
        assert [type(ei.value).__name__, str(ei.value)] == case["want"]["error"], case["name"]


def test_bxmi_opts_environment_applies_the_tuning_knobs():
    """BXMI_OPTS="key=value,..." is read when the library is loaded (bxmi/_ffi.py): known keys are applied through
    bxmi_set_option (no GPU call involved), an unknown key is an error at load time, not a silently ignored typo."""
    code = "from bxmi import _ffi; _ffi.load(); print('loaded')"
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "bx-python_amd"), BXMI_OPTS="ivl.bm_variant=2, core.poll=0,ivl.sl_run_cap=128")
    assert subprocess.check_output([sys.executable, "-c", code], text=True, env=env).strip() == "loaded"
    env["BXMI_OPTS"] = "ivl.no_such_knob=1"
    r = subprocess.run([sys.executable, "-c", code], text=True, env=env, capture_output=True)
    assert r.returncode != 0 and "unknown key" in r.stderr


def test_count_range_from_the_run_list_host_logic(monkeypatch):
    """bx.bitset's read-only phase (count_range answered from the set's run list after two device calls) against the
    oracle BinnedBitSet on a device look-alike: random op sequences with mutations in between (each drops the list),
    inverted sets (ALL_ONE bins: the first bin's offset is subtracted, binBits.c:155,161), empty and full ranges."""
    import bx.bitset as B
    from oracle import oracle as O

    calls = {"device": 0}

    class FakeDev:
        def __init__(self, size, granularity=1024, flat=False):
            self.o = O.OracleBinnedBitSet(size, granularity)
            self.size, self.bin_size, self.flat = size, self.o.bin_size, flat

        def check_index(self, i):
            self.o._check_index(i)

        def check_range_count(self, s, c):
            self.o._check_range_count(s, c)

        def check_same_size(self, other):
            self.o._check_same(other.o)

        def set_ranges(self, s, c):
            self.o.set_ranges(np.asarray(s, np.int32), np.asarray(c, np.int32))

        def count_range_checked(self, s, c):
            calls["device"] += 1
            return self.o.count_range(s, c)

        def runs(self, start=0):
            rs, re = self.o.runs()
            keep = re > start
            return np.maximum(rs[keep], start), re[keep]

        def bin_states(self):
            return np.asarray(self.o.states(), dtype=np.uint8)

        def next(self, start, val):
            return self.o.next_set(start) if val else self.o.next_clear(start)

        def invert(self):
            self.o.invert()

        def iand(self, other):
            self.o.iand(other.o)

        def ior(self, other):
            self.o.ior(other.o)

        def clear(self, i):
            self.o.clear(i)

        def get(self, i):
            return self.o[i]

    monkeypatch.setattr(B, "DeviceBitSet", FakeDev)
    rng = np.random.default_rng(5)
    size = 300_000
    a, ref = B.BinnedBitSet(size, 1000), O.OracleBinnedBitSet(size, 1000)
    other, oref = B.BinnedBitSet(size, 1000), O.OracleBinnedBitSet(size, 1000)
    for s, c in zip(rng.integers(0, size - 3000, 300).tolist(), rng.integers(1, 3000, 300).tolist()):
        other.set_range(s, c)
        oref.set_range(s, c)
    answered_on_host = 0
    for rnd in range(60):
        op = rng.integers(0, 6)
        if op == 0:
            for s, c in zip(rng.integers(0, size - 5000, 20).tolist(), rng.integers(0, 5000, 20).tolist()):
                a.set_range(s, c)
                ref.set_range(s, c)
        elif op == 1:
            a.invert()
            ref.invert()
        elif op == 2:
            a.iand(other)
            ref.iand(oref)
        elif op == 3:
            a.ior(other)
            ref.ior(oref)
        elif op == 4:
            i = int(rng.integers(0, size))
            a.clear(i)
            ref.clear(i)
        before = calls["device"]
        qs = rng.integers(0, size, 40).tolist() + [0, 0, size - 1, 999, 1000, 1001]
        for s in qs:
            c = int(rng.integers(0, size - s + 1)) if rng.random() < 0.5 else int(rng.integers(0, min(size - s, 3000) + 1))
            assert a.count_range(s, c) == ref.count_range(s, c), (rnd, op, s, c)
            assert a.next_set(s) == ref.next_set(s) and a.next_clear(s) == ref.next_clear(s)
        assert a.count_range(0, size) == ref.count_range(0, size)
        assert calls["device"] - before <= 2  # after a mutation: two device calls, then the run list
        answered_on_host += len(qs) + 1 - (calls["device"] - before)
    assert answered_on_host > 2000


def test_ring_kernels_keep_registers_of_loads_in_flight_untouched():
    """The flat walk's record loads are issued by inline asm (count_dense.hpp: bd_issue_load / bd_wait), so the compiler
    does not know which registers are still being written.  tools/check_ring_isa.py compiles intervals.hip to gfx950
    assembly (no GPU needed) and follows every such kernel's basic blocks: no instruction may name a register while a
    hand-issued load is on its way to it, and every wait must name the destination of a load in flight."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_ring_isa.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    m = re.match(r"(\d+) kernels checked, 0 with problems", last)
    assert m and int(m.group(1)) >= 6, last  # (one shape per stage and layout since round 4)
    assert " ring " in r.stdout and "two sets" in r.stdout and "persistent walk W8 1 DEPTH 3" in r.stdout
    assert "persistent walk W8 1 DEPTH 3 offset cells" in r.stdout and "persistent walk W8 0 DEPTH 3 offset cells" in r.stdout


def test_offset_cells_encoding_and_unit_model(tmp_path):
    """bx-python_amd/csrc/offset_cells.hpp is plain C++ as well as device code: tests/cpp/offset_cells_test.cpp checks the
    cell encoding (pack / rank at every position), the record format and the density rule against brute force, and a
    scalar model of one unit's two images -- built the way bo_image_kernel builds them, asked the way the search kernel asks
    -- against the definition of an overlap (repeated coordinates, cells with more than five keys, units of 16..128 cells)."""
    exe = str(tmp_path / "offset_cells_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "bx-python_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "offset_cells_test.cpp"), "-o", exe])
    out = subprocess.check_output([exe], text=True, timeout=300)
    assert out.strip().endswith("offset cells ok"), out


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus N` launches its own N ranks (the driver's command shape carries no launcher); with fewer than N
    devices visible it must exit non-zero with a message, never run one rank and print n_gpus 1."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_DRY_MULTI"):
        env.pop(k, None)
    import torch

    n = (torch.cuda.device_count() if torch.cuda.is_available() else 0) + 7
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0
    assert "refusing to run fewer ranks than asked" in p.stderr, p.stderr[-2000:]
    assert not p.stdout.strip(), p.stdout[-500:]   # no bench line for a job that did not run


def test_bench_refuses_a_launcher_of_another_size():
    """WORLD_SIZE from a launcher that disagrees with --gpus: no line for a job of another size."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr, p.stderr[-2000:]
