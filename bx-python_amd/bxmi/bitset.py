"""
Dense device bitset -- the engine under ``bx.bitset.BinnedBitSet`` / ``BitSet``
(reference: lib/bx/bitset.pyx over src/binBits.c and src/kent/bits.c).

``DeviceBitSet`` is the batch API: whole arrays of ranges per kernel launch.
Argument errors raise the same exception types and messages as bitset.pyx.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import as_i32, call, ptr

MAX_INT = 2147483647  # bitset.pyx:105
MAX = 512 * 1024 * 1024  # bitset.pyx:196


def _first_bad_range(size, starts, lens, binned=True):
    """Index and exception of the first range bitset.pyx's bb_check_range_count (:184-189) rejects."""
    s64 = starts.astype(np.int64)
    l64 = lens.astype(np.int64)
    bad = (s64 < 0) | (s64 >= size) | (l64 < 0) | (s64 + l64 > size)
    if not bad.any():
        return -1, None
    k = int(np.argmax(bad))
    s, n = int(s64[k]), int(l64[k])
    if s < 0:
        return k, IndexError("BitSet index (%d) must be non-negative." % s)
    if s >= size:
        return k, IndexError("%d is larger than the size of this BitSet (%d)." % (s, size))
    if n < 0:
        return k, IndexError("Count (%d) must be non-negative." % n)
    if binned:
        return k, IndexError("End (%d) is larger than the size of this BinnedBitSet (%d)." % (s + n, size))
    return k, IndexError("End %d is larger than the size of this BitSet (%d)." % (s + n, size))


class DeviceBitSet:
    """One chromosome's bits as dense uint64 words in HBM (+ the reference's per-bin tri-state)."""

    def __init__(self, size=MAX, granularity=1024, flat=False):
        if size > MAX_INT:  # bitset.pyx:113-114 / :201-202
            kind = "BitSet" if flat else "BinnedBitSet"
            raise ValueError("%d is larger than the maximum %s size of %d." % (size, kind, MAX_INT))
        size = int(size)
        granularity = 0 if flat else int(granularity)
        if size < 1 or (not flat and granularity < 1):
            raise ValueError("size and granularity must be >= 1 (the reference's behaviour is undefined otherwise)")
        _ffi.require_gpu()
        h = C.c_void_p()
        call("bxmi_bits_create", size, granularity, C.byref(h))
        self._h = h
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        call("bxmi_bits_info", h, C.byref(a), C.byref(b), C.byref(c))
        self.size, self.bin_size, self.nbins = a.value, b.value, c.value
        self.flat = flat

    def close(self):
        if getattr(self, "_h", None):
            _ffi.load().bxmi_bits_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- checks (bitset.pyx:78-100,177-192) ---------------------------------
    def check_index(self, index):
        if index < 0:
            raise IndexError("BitSet index (%d) must be non-negative." % index)
        if index >= self.size:
            raise IndexError("%d is larger than the size of this BitSet (%d)." % (index, self.size))

    def check_range_count(self, start, count):
        self.check_index(start)
        if count < 0:
            raise IndexError("Count (%d) must be non-negative." % count)
        if start + count > self.size:
            if self.flat:
                raise IndexError("End %d is larger than the size of this BitSet (%d)." % (start + count, self.size))
            raise IndexError("End (%d) is larger than the size of this BinnedBitSet (%d)." % (start + count, self.size))

    def check_same_size(self, other):
        if self.size != other.size:
            raise ValueError("BitSets must have the same size")

    # ---- batch ops -----------------------------------------------------------
    def set_ranges(self, starts, lens):
        """set_range(starts[i], lens[i]) for i in order; the first invalid range raises after the
        valid prefix has been applied (what a per-line loop over the reference would have done)."""
        s, n = as_i32(starts), as_i32(lens)
        k, err = _first_bad_range(self.size, s, n, not self.flat)
        m = len(s) if k < 0 else k
        if m:
            call("bxmi_bits_set_ranges", self._h, ptr(s), ptr(n), m)
        if err is not None:
            raise err

    def count_ranges(self, starts, lens):
        s, n = as_i32(starts), as_i32(lens)
        k, err = _first_bad_range(self.size, s, n, not self.flat)
        if err is not None:
            raise err
        out = np.empty(len(s), dtype=np.int32)
        if len(s):
            call("bxmi_bits_count_ranges", self._h, ptr(s), ptr(n), len(s), ptr(out))
        return out

    def count_range(self, start, count):
        self.check_range_count(start, count)
        return self.count_range_checked(int(start), int(count))

    def count_range_checked(self, start, count):
        """count_range for arguments the caller has validated already (the drop-in class has): the per-call path an
        unmodified script pays once per line, so nothing is looked up or allocated here that can be kept."""
        fast = self.__dict__.get("_cr")
        if fast is None:
            out = C.c_int32(0)
            fast = self._cr = (_ffi.load().bxmi_bits_count_range, out, C.byref(out))
        fn, out, ref = fast
        rc = fn(self._h, start, count, ref)
        if rc:
            _ffi.check(rc)
        return out.value

    def get(self, index):
        self.check_index(index)
        bit = C.c_int(0)
        call("bxmi_bits_get", self._h, int(index), C.byref(bit))
        return bit.value

    def set(self, index):
        self.check_index(index)
        call("bxmi_bits_set", self._h, int(index))

    def clear(self, index):
        self.check_index(index)
        call("bxmi_bits_clear", self._h, int(index))

    def next(self, start, val):
        self.check_index(start)
        out = C.c_int32(0)
        call("bxmi_bits_next", self._h, int(start), int(val), C.byref(out))
        return out.value

    def iand(self, other):
        self.check_same_size(other)
        self._same_bins(other)
        call("bxmi_bits_and", self._h, other._h)

    def ior(self, other):
        self.check_same_size(other)
        self._same_bins(other)
        call("bxmi_bits_or", self._h, other._h)

    def ixor(self, other):
        self.check_same_size(other)
        call("bxmi_bits_xor", self._h, other._h)

    def and_count(self, other):
        """Fused ``self &= other`` + popcount of the result over [0, size)."""
        self.check_same_size(other)
        self._same_bins(other)
        c = C.c_int64(0)
        call("bxmi_bits_and_count", self._h, other._h, C.byref(c))
        return c.value

    def invert(self):
        call("bxmi_bits_not", self._h)

    def _same_bins(self, other):
        if self.bin_size != other.bin_size or self.nbins != other.nbins:
            # binBits.c:233 asserts this, but the assert is compiled out: undefined in the reference
            raise ValueError("BitSets must have the same granularity")

    def runs(self, start=0, cap_hint=1 << 20):
        """Maximal runs of set bits in [start, size) as (run_start int32[], run_end int32[])."""
        if start == self.size:
            return np.empty(0, np.int32), np.empty(0, np.int32)
        self.check_index(start)
        n = C.c_int64(0)
        cap = int(cap_hint)
        for _ in range(2):
            rs, re = np.empty(cap, np.int32), np.empty(cap, np.int32)
            rc = call("bxmi_bits_runs", self._h, int(start), ptr(rs), ptr(re), cap, C.byref(n), allow=(_ffi.ERANGE,))
            if rc == _ffi.OK:
                return rs[: n.value].copy(), re[: n.value].copy()
            cap = n.value
        raise _ffi.BxmiError(_ffi.ERANGE, "runs: buffer still too small")

    def bin_states(self):
        out = np.empty(self.nbins, dtype=np.uint8)
        call("bxmi_bits_bin_states", self._h, ptr(out))
        return out

    def words_dev(self):
        p, n = C.c_void_p(), C.c_int64(0)
        call("bxmi_bits_words_dev", self._h, C.byref(p), C.byref(n))
        return p.value, n.value

    def to_bits(self):
        """Logical bit values [0, size) as a uint8 array (test helper; one D2H copy)."""
        p, n = self.words_dev()
        w = np.empty(n, dtype=np.uint64)
        call("bxmi_memcpy_d2h", ptr(w), p, n * 8)
        return np.unpackbits(w.view(np.uint8), bitorder="little")[: self.size]


class BitSetGroup:
    """Several bitsets (one per chromosome) operated on in ONE kernel launch.

    The batch form of the per-chromosome loops in the reference's scripts:
    ``for key in bits1: bits1[key].iand(bits2[key])`` (bed_intersect_basewise.py:25-28) and
    ``total += bitsets[chrom].count_range(0, size)`` (bed_coverage.py:27-29)."""

    def __init__(self, members):
        self.members = list(members)
        if not self.members:
            raise ValueError("a group needs at least one bitset")
        n = len(self.members)
        arr = (C.c_void_p * n)(*[m._h for m in self.members])
        g = C.c_void_p()
        call("bxmi_bits_group_create", arr, n, C.byref(g))
        self._g = g
        self._counts = _ffi.DeviceArray(8 * n)

    def close(self):
        if getattr(self, "_g", None):
            _ffi.load().bxmi_bits_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _pair(self, other):
        if len(other.members) != len(self.members):
            raise ValueError("groups must have the same number of members")
        for a, b in zip(self.members, other.members):
            a.check_same_size(b)
            a._same_bins(b)

    def _fetch(self):
        call("bxmi_synchronize", None)
        return self._counts.to_numpy(np.int64, len(self.members))

    def iand(self, other, want_counts=False):
        """member[i] &= other.member[i] for all i; optionally the popcount of every result."""
        self._pair(other)
        if want_counts:
            self._counts.zero()
        call("bxmi_bits_group_and_dev", self._g, other._g, self._counts.ptr if want_counts else None, None)
        return self._fetch() if want_counts else call("bxmi_synchronize", None) and None

    def ior(self, other):
        self._pair(other)
        call("bxmi_bits_group_or_dev", self._g, other._g, None)
        call("bxmi_synchronize", None)

    def popcounts(self):
        """count_range(0, size) of every member (true popcounts: callers on inverted sets beware the
        reference's ALL_ONE arithmetic only bites when start % bin_size != 0, never here)."""
        self._counts.zero()
        call("bxmi_bits_group_popcount_dev", self._g, self._counts.ptr, None)
        return self._fetch()
