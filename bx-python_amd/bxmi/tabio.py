"""
Native pre-parse of delimited interval text for the readers of `bxmi.genomic` (the reader side):
csrc/bedparse.cpp's table mode classifies and parses, in one C++ pass, every line whose outcome under
lib/bx/intervals/io.py:106-216 / lib/bx/tabular/io.py:86-156 is certain -- blank lines, comment and header lines, rows
whose fields are already in the form the reader writes back -- and stops at the first line that is not; the reader hands
out the parsed prefix and then goes on per line from there, so items, exceptions and bookkeeping are unchanged.
Host code only (libbxmi.so, no GPU needed).  Set BXMI_NO_FASTPARSE=1 to force the per-line path everywhere.
"""
import ctypes as C

import numpy as np

from . import _ffi, bedio
from ._ffi import call


def enabled():
    return bedio.enabled()


class ParsedTable:
    """The consumed prefix of an input: per-LINE arrays (Python lists: they are read one element at a time)."""

    def __init__(self, data, chrom_col, start_col, end_col, strand_col, prefixes, source_lines=None, source_file=None):
        self.data = data
        self.source_lines, self.source_file = source_lines, source_file
        arr = (C.c_char_p * len(prefixes))(*[p.encode("ascii") for p in prefixes])
        h = C.c_void_p()
        call("bxmi_tab_parse", data, len(data), chrom_col, start_col, end_col, strand_col, arr, len(prefixes), C.byref(h))
        try:
            n, nc, so = C.c_int64(), C.c_int32(), C.c_int64()
            call("bxmi_tab_info", h, C.byref(n), C.byref(nc), C.byref(so))
            self.n, self.stop_off = n.value, so.value
            p = [C.c_void_p() for _ in range(7)]
            call("bxmi_tab_columns", h, *[C.byref(x) for x in p])

            def col(ptr, ctype):
                if self.n == 0:
                    return np.empty(0, dtype=ctype)
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(self.n,)).copy()

            self.kind_a, self.off_a, self.len_a = col(p[0], C.c_uint8), col(p[1], C.c_int64), col(p[2], C.c_int32)
            self.chrom_a, self.start_a, self.end_a, self.strand_a = col(p[3], C.c_int32), col(p[4], C.c_int64), col(p[5], C.c_int64), col(p[6], C.c_uint8)
            lib = _ffi.load()
            self.names = [lib.bxmi_tab_chrom_name(h, i).decode("ascii") for i in range(nc.value)]
        finally:
            _ffi.load().bxmi_tab_destroy(h)
        self.kind, self.off, self.len = self.kind_a.tolist(), self.off_a.tolist(), self.len_a.tolist()
        self.chrom, self.start, self.end, self.strand = self.chrom_a.tolist(), self.start_a.tolist(), self.end_a.tolist(), self.strand_a.tolist()

    def line(self, i):
        """Line i without its line end (plain ASCII by construction)."""
        o = self.off[i]
        return self.data[o:o + self.len[i]].decode("ascii")

    def raw_line(self, i):
        """Line i as the input delivered it."""
        if self.source_lines is not None:
            return self.source_lines[i]
        o = self.off[i]
        e = o + self.len[i]
        return self.data[o:e + 1].decode("ascii") if e < len(self.data) else self.data[o:e].decode("ascii")

    def rest_lines(self, source):
        """The input after the parsed prefix, as the per-line code expects it."""
        if self.source_lines is not None:
            return self.source_lines[self.n:]
        if self.stop_off < 0:
            return []
        return bedio.text_lines(self.source_file, self.data[self.stop_off:])


def parse_input(source, chrom_col, start_col, end_col, strand_col, prefixes):
    """ParsedTable for a list / tuple of '\\n'-terminated lines or a text file at its beginning; None for anything else
    (generators, pipes, lines without line ends, non-ASCII prefixes ...) or when nothing could be consumed."""
    is_file = False
    try:
        prefixes = [str(p) for p in prefixes]
        if any(not p or not p.isascii() or "\0" in p for p in prefixes):
            return None
        # the native parser takes column NUMBERS; the per-line code also accepts Python's negative indices
        if min(int(chrom_col), int(start_col), int(end_col)) < 0 or (strand_col is not None and int(strand_col) < -1):
            return None
        if isinstance(source, (list, tuple)):
            if not source or not all(type(x) is str for x in source):
                return None
            text = "".join(source)
            # one line per element: every element ends with its only '\n' (a '\r' anywhere stops the parser by itself)
            if text.count("\n") != len(source) or not all(x.endswith("\n") for x in source):
                return None
            try:
                data = text.encode("ascii")
            except UnicodeEncodeError:
                data = text.encode("utf-8")  # (the parser stops at the first non-ASCII byte)
            table = ParsedTable(data, chrom_col, start_col, end_col, strand_col, prefixes, source_lines=source)
        else:
            data = bedio.file_bytes(source)
            if data is None:
                return None
            is_file = True  # (drained: whatever happens from here on, the per-line code must find the file at its beginning)
            table = ParsedTable(data, chrom_col, start_col, end_col, strand_col, prefixes, source_file=source)
            if table.n == 0:
                source.seek(0)  # nothing taken: the per-line code reads the file itself
                return None
        return table if table.n else None
    except (_ffi.BxmiError, OSError, ValueError, TypeError):  # (TypeError: a column given as None or text -- the per-line code says what is wrong with it)
        if is_file:
            try:
                source.seek(0)
            except (OSError, ValueError):
                pass
        return None
