"""Child process of tests/test_parser_sanitized.py: hammers the two text parsers of csrc/bedparse.cpp (built with
AddressSanitizer + UBSan, loaded by path) with hostile input and walks everything they hand back.  Any out-of-bounds
access, overflow or leak of a dangling view aborts the process; the plain-BED results are also compared with the
per-line model of tests/test_host_logic.py.  argv: library path, seed, rounds."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_host_logic import _python_parse  # noqa: E402

L = C.CDLL(sys.argv[1])
rng = np.random.default_rng(int(sys.argv[2]))
rounds = int(sys.argv[3])
vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
L.bxmi_bed_parse.argtypes = [C.c_char_p, i64, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
L.bxmi_bed_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i32), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
L.bxmi_bed_columns.argtypes = [vp] + [C.POINTER(vp)] * 5
L.bxmi_bed_chrom_name.argtypes = [vp, i32]
L.bxmi_bed_chrom_name.restype = C.c_char_p
L.bxmi_bed_destroy.argtypes = [vp]
L.bxmi_bed_emit_lines.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
L.bxmi_tab_parse.argtypes = [C.c_char_p, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.POINTER(vp)]
L.bxmi_tab_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i32), C.POINTER(i64)]
L.bxmi_tab_columns.argtypes = [vp] + [C.POINTER(vp)] * 7
L.bxmi_tab_chrom_name.argtypes = [vp, i32]
L.bxmi_tab_chrom_name.restype = C.c_char_p
L.bxmi_tab_destroy.argtypes = [vp]

PIECES = [b"chr1", b"chrX_random", b"", b" ", b"\t", b"\t\t", b"  ", b"\n", b"\r\n", b"\r", b"#", b"# c", b"track name=x", b"0", b"-0", b"+5", b"17",
          b"2147483647", b"2147483648", b"-2147483649", b"9223372036854775807", b"9223372036854775808", b"99999999999999999999999", b"1_000", b"1e3",
          b"0x10", b" 12", b"12 ", b"+", b"-", b".", b"++", b"\x00", b"\xff\xfe", "é".encode(), b"a" * 300, b"\t" * 40, b"1\t2\t3", b"chr2\t5\t9\tn\t0\t-"]


def hostile(n_lines):
    out = []
    for _ in range(n_lines):
        k = int(rng.integers(0, 4))
        if k == 0:  # a plain row
            s = int(rng.integers(0, 10**6))
            out.append(b"chr%d\t%d\t%d\tn\t0\t%s\n" % (int(rng.integers(1, 4)), s, s + int(rng.integers(0, 500)), b"+-"[int(rng.integers(0, 2)):][:1]))
        elif k == 1:  # pieces glued with random separators
            parts = [PIECES[int(i)] for i in rng.integers(0, len(PIECES), size=int(rng.integers(1, 7)))]
            sep = [b"\t", b" ", b"", b"\t "][int(rng.integers(0, 4))]
            out.append(sep.join(parts) + [b"\n", b"", b"\r\n"][int(rng.integers(0, 3))])
        elif k == 2:  # random bytes
            out.append(bytes(rng.integers(0, 256, size=int(rng.integers(0, 60)), dtype=np.uint8).tolist()) + b"\n")
        else:
            out.append([b"\n", b"#x\n", b"   \n", b"chr1\t1\n", b"chr1\t5\t3\n", b"chr1\t1\t2"][int(rng.integers(0, 6))])
    return b"".join(out)


def arr(ptr, ctype, n):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).copy() if n else np.empty(0)


checked = 0
for r in range(rounds):
    data = hostile(int(rng.integers(0, 40))) if r % 7 else bytes(rng.integers(0, 256, size=int(rng.integers(0, 400)), dtype=np.uint8).tolist())
    buf = C.create_string_buffer(data, len(data))  # exactly len(data) bytes, no terminator to lean on
    cols = [int(x) for x in rng.integers(0, 5, size=3)] if r % 3 == 0 else [0, 1, 2]
    # ---- BED mode
    h = vp()
    rc = L.bxmi_bed_parse(C.cast(buf, C.c_char_p), len(data), cols[0], cols[1], cols[2], C.byref(h))
    if rc == 0:
        n, nc, sl, so, seen = i64(), i32(), i64(), i64(), i64()
        assert L.bxmi_bed_info(h, C.byref(n), C.byref(nc), C.byref(sl), C.byref(so), C.byref(seen)) == 0
        p = [vp() for _ in range(5)]
        assert L.bxmi_bed_columns(h, *[C.byref(x) for x in p]) == 0
        chrom, start, end = arr(p[0], C.c_int32, n.value), arr(p[1], C.c_int64, n.value), arr(p[2], C.c_int64, n.value)
        off, ln = arr(p[3], C.c_int64, n.value), arr(p[4], C.c_int32, n.value)
        names = [L.bxmi_bed_chrom_name(h, i) for i in range(nc.value)]
        assert L.bxmi_bed_chrom_name(h, nc.value) is None and L.bxmi_bed_chrom_name(h, -1) is None
        assert all(0 <= c < nc.value for c in chrom.tolist()) and all(0 <= o and o + k <= len(data) for o, k in zip(off.tolist(), ln.tolist()))
        assert -1 <= so.value <= len(data)
        try:
            text = data.decode("utf-8")
        except UnicodeDecodeError:
            text = None
        if text is not None:
            rows, stop = _python_parse(text, *cols)
            assert n.value <= len(rows), (n.value, len(rows))  # the strict parser may stop EARLIER than the per-line code, never later
            for i in range(n.value):
                assert (names[chrom[i]].decode(), int(start[i]), int(end[i])) == rows[i][:3], (i, rows[i])
            if stop is not None:
                assert sl.value >= 0
        if n.value:  # emit a random selection to /dev/null through the C writer
            mask = (rng.random(n.value) < 0.5).astype(np.uint8)
            fd = os.open(os.devnull, os.O_WRONLY)
            assert L.bxmi_bed_emit_lines(h, C.cast(buf, C.c_char_p), mask.tobytes(), b" ", fd) == 0
            os.close(fd)
        assert L.bxmi_bed_destroy(h) == 0
    # ---- table mode
    prefixes = [b"#", b"track", b"browser"][: int(rng.integers(0, 4))]
    pa = (C.c_char_p * max(1, len(prefixes)))(*prefixes) if prefixes else None
    t = vp()
    strand_col = int(rng.integers(-1, 7))
    rc = L.bxmi_tab_parse(C.cast(buf, C.c_char_p), len(data), cols[0], cols[1], cols[2], strand_col, pa, len(prefixes), C.byref(t))
    if rc == 0:
        nl, nc, so = i64(), i32(), i64()
        assert L.bxmi_tab_info(t, C.byref(nl), C.byref(nc), C.byref(so)) == 0
        p = [vp() for _ in range(7)]
        assert L.bxmi_tab_columns(t, *[C.byref(x) for x in p]) == 0
        kind, off, ln = arr(p[0], C.c_uint8, nl.value), arr(p[1], C.c_int64, nl.value), arr(p[2], C.c_int32, nl.value)
        chrom, start, end, strand = (arr(p[3], C.c_int32, nl.value), arr(p[4], C.c_int64, nl.value), arr(p[5], C.c_int64, nl.value),
                                     arr(p[6], C.c_uint8, nl.value))
        assert set(kind.tolist()) <= {0, 1, 2, 3} and -1 <= so.value <= len(data)
        for i in np.nonzero(kind == 0)[0].tolist():
            assert 0 <= chrom[i] < nc.value and start[i] <= end[i] and strand[i] in (0, ord("+"), ord("-"))
            assert 0 <= off[i] and off[i] + ln[i] <= len(data)
            fields = data[off[i]:off[i] + ln[i]].split(b"\t")
            got = (L.bxmi_tab_chrom_name(t, int(chrom[i])), int(start[i]), int(end[i]))
            assert (fields[cols[0]], int(fields[cols[1]]), int(fields[cols[2]])) == got, (data[off[i]:off[i] + ln[i]], cols, got)
        assert L.bxmi_tab_chrom_name(t, nc.value) is None
        assert L.bxmi_tab_destroy(t) == 0
    checked += 1
print("parser fuzz: %d inputs" % checked)
