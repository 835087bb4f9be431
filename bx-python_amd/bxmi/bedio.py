"""
Native BED ingest: text -> SoA columns in one C++ pass (csrc/bedparse.cpp).

The parser is strict: it consumes plain ASCII BED lines and stops at the first line it is not sure
about; callers process the consumed prefix in bulk and hand the rest of the text to the per-line
code that mirrors the reference, so behaviour (including exceptions) is unchanged.
Set BXMI_NO_FASTPARSE=1 to force the per-line path everywhere.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from ._ffi import call


def enabled():
    return not os.environ.get("BXMI_NO_FASTPARSE")


def file_bytes(f):
    """Remaining bytes of a text-mode file object opened on a real file, or None if `f` is something else
    (stdin pipe wrappers, fileinput, lists of lines...) or holds a carriage return anywhere: universal-newline
    translation turns '\\r\\n' and lone '\\r' into '\\n' in what `for line in f` yields, which the byte-level fast path
    could only imitate, so such files take the per-line path from the first line on (the file is rewound for it)."""
    buf = getattr(f, "buffer", None)
    if buf is None or not hasattr(buf, "read"):
        return None
    try:
        if f.tell() != 0:  # somebody already consumed text: the decoder may hold read-ahead
            return None
        data = buf.read()
        if b"\r" in data:
            f.seek(0)
            return None
    except (OSError, ValueError):
        return None
    return data


def text_lines(f, data):
    """What `for line in f` would yield for the bytes `data` of file object `f`: same codec, same error policy,
    universal newlines (str.splitlines would also break at \\x0b, \\x0c, \\x1c-\\x1e, \\x85, \\u2028, \\u2029)."""
    import io

    return list(io.TextIOWrapper(io.BytesIO(data), encoding=getattr(f, "encoding", None) or "utf-8", errors=getattr(f, "errors", None) or "strict",
                                 newline=None))


class ParsedBed:
    """Columns of the consumed prefix of a BED text (numpy views into the C++ object)."""

    def __init__(self, data, chrom_col=0, start_col=1, end_col=2):
        self.data = data
        h = C.c_void_p()
        call("bxmi_bed_parse", data, len(data), chrom_col, start_col, end_col, C.byref(h))
        self._h = h
        n, nc, sl, so, seen = C.c_int64(), C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64()
        call("bxmi_bed_info", h, C.byref(n), C.byref(nc), C.byref(sl), C.byref(so), C.byref(seen))
        self.n, self.stop_off = n.value, so.value
        p = [C.c_void_p() for _ in range(5)]
        call("bxmi_bed_columns", h, *[C.byref(x) for x in p])

        def view(ptr, ctype, dtype):
            if self.n == 0:
                return np.empty(0, dtype=dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(self.n,))

        self.chrom = view(p[0], C.c_int32, np.int32)
        self.start = view(p[1], C.c_int64, np.int64)
        self.end = view(p[2], C.c_int64, np.int64)
        self.line_off = view(p[3], C.c_int64, np.int64)
        self.line_len = view(p[4], C.c_int32, np.int32)
        lib = _ffi.load()
        self.names = [lib.bxmi_bed_chrom_name(h, i).decode("ascii") for i in range(nc.value)]

    def rest_lines(self, first_row=None, f=None):
        """Text lines (with line ends) from row `first_row` of the consumed prefix, or -- if None -- from
        where the parser stopped; [] when everything was consumed.  `f` = the file object the bytes came from
        (its codec and error policy decode the rest, as iterating it would have)."""
        off = int(self.line_off[first_row]) if first_row is not None else self.stop_off
        if off < 0:
            return []
        return text_lines(f, self.data[off:])

    def emit(self, mask, suffix, fd):
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        call("bxmi_bed_emit_lines", self._h, self.data, m.ctypes.data_as(C.c_void_p), suffix, fd)

    def close(self):
        if getattr(self, "_h", None):
            self.chrom = self.start = self.end = self.line_off = self.line_len = None
            _ffi.load().bxmi_bed_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
