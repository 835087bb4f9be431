"""
bx.bitset -- drop-in for the reference's Cython module lib/bx/bitset.pyx,
served by the MI355X engine (bxmi.bitset.DeviceBitSet over libbxmi.so).

Same classes, signatures, exception types and messages as the reference:
  BitSet        bitset.pyx:107-173
  BinnedBitSet  bitset.pyx:198-241
  MAX           bitset.pyx:196

How the per-call API meets a batch engine: ``set_range``/``set`` calls are
validated immediately (so errors surface where the reference raises them) and
queued; the queue is flushed as ONE ``set_ranges`` kernel launch before any
call that reads or combines bits.  A loop of ``next_set``/``next_clear`` calls
(bed_intersect_basewise.py:32-38) is served from the run list the device
extracts in one pass.  Nothing is computed on the CPU: without libbxmi.so and
a GPU these classes raise.
"""
import bisect
import operator

import numpy as np

from bxmi.bitset import MAX, MAX_INT, DeviceBitSet

__all__ = ["BitSet", "BinnedBitSet", "MAX"]

_FLUSH_AT = 1 << 20


def _cint(x):
    """Cython's coercion of a Python object to C ``int`` (OverflowError / TypeError as in the reference)."""
    if type(x) is not int:
        if isinstance(x, float):
            x = int(x)
        else:
            try:
                x = operator.index(x)
            except TypeError:
                try:
                    x = int(x) if hasattr(x, "__int__") and not isinstance(x, str) else None
                except (TypeError, ValueError):
                    x = None
                if x is None:
                    raise TypeError("an integer is required") from None
    if not -2147483648 <= x <= 2147483647:
        raise OverflowError("value too large to convert to int")
    return x


_MIRROR_MAX_RUNS = 2_000_000  # count_range answers from a host copy of the device's run list only while it is this short


class _Queued:
    """Shared machinery: queued set_range + run-list cache on top of a DeviceBitSet."""

    def _init(self, dev):
        self._d = dev
        self._ps, self._pc = [], []
        self._runs = None  # (from_pos, run_starts list, run_ends list)
        self._scans = 0  # next_* calls since the last mutation
        self._mirror = None  # (run_starts, run_ends, bits before each run, per-bin states or None): see _count
        self._reads = 0  # count_range calls since the last mutation
        self._mirror_ok = True  # False: the run list was found too long to mirror (until the next mutation)

    def _touch(self):
        self._mirror_ok = True
        self._runs = None
        self._scans = 0
        self._mirror = None
        self._reads = 0

    def _flush(self):
        if self._ps:
            s, c = self._ps, self._pc
            self._ps, self._pc = [], []
            self._d.set_ranges(np.array(s, dtype=np.int32), np.array(c, dtype=np.int32))

    def _queue(self, start, count):
        self._ps.append(start)
        self._pc.append(count)
        self._runs = None
        self._scans = 0
        self._mirror = None
        self._reads = 0
        self._mirror_ok = True
        if len(self._ps) >= _FLUSH_AT:
            self._flush()

    def _next(self, start, val):
        """binBitsFindSet/FindClear semantics via the device run list (or one find kernel)."""
        self._flush()
        size = self._d.size
        self._scans += 1
        if self._runs is None or start < self._runs[0]:
            if self._scans < 2:
                return self._d.next(start, val)
            rs, re = self._d.runs(start)
            self._runs = (start, rs.tolist(), re.tolist())
        _, rs, re = self._runs
        i = bisect.bisect_right(re, start)  # first run ending after `start`
        if val:
            if i == len(rs):
                return size
            return start if rs[i] <= start else rs[i]
        if i < len(rs) and rs[i] <= start:
            return re[i]  # inside a run: first clear bit is its end (== size when it runs to the end)
        return start

    def _count(self, start, count, binned):
        """count_range behind a read-only phase (scripts/bed_intersect.py:46-60 asks once per line and never writes
        again): the first calls after a mutation go to the device one by one; from the third on the set's RUN LIST --
        extracted on the device once, what next_set / next_clear already walk -- answers them: two bisections and a
        subtraction, nothing crosses PCIe.  A run list describes the logical bits; what the reference makes observable
        on top of them is the first bin's offset when that bin is ALL_ONE (binBits.c:155,161), taken from the per-bin
        states.  Any mutation drops the list (_touch / _queue)."""
        m = self._mirror
        if m is None:
            self._reads += 1
            if self._reads < 3:
                return self._d.count_range_checked(start, count)
            if self._reads > 3 and not self._mirror_ok:
                return self._d.count_range_checked(start, count)  # (a set with millions of runs: asked on the device every time)
            rs, re = self._d.runs(0)
            if len(rs) > _MIRROR_MAX_RUNS:  # tens of millions of runs would cost gigabytes of host lists to save microseconds
                self._mirror_ok = False
                return self._d.count_range_checked(start, count)
            before = np.concatenate(([0], np.cumsum(re.astype(np.int64) - rs.astype(np.int64)))).tolist()
            rs, re = rs.tolist(), re.tolist()
            m = self._mirror = (rs, re, before, bytes(self._d.bin_states()) if binned else None)
            self._runs = (0, rs, re)
        if count <= 0:
            return 0
        rs, re, before, states = m
        end = start + count
        i0 = bisect.bisect_right(re, start)  # first run that ends after `start`
        i1 = bisect.bisect_left(rs, end)     # first run that starts at or after `end`
        n = 0
        if i1 > i0:
            n = before[i1] - before[i0]
            if rs[i0] < start:
                n -= start - rs[i0]
            if re[i1 - 1] > end:
                n -= re[i1 - 1] - end
        if states is not None:
            bs = self._d.bin_size
            b = start // bs
            if states[b] == 1:  # ALL_ONE: the reference counts `piece - offset` for the first bin
                n -= start - b * bs
        return n

    @property
    def size(self):
        return self._d.size


class BinnedBitSet(_Queued):
    """bitset.pyx:198-241"""

    def __init__(self, size=MAX, granularity=1024):
        self._init(DeviceBitSet(size, granularity))

    def __getitem__(self, index):
        self._d.check_index(index)
        self._flush()
        return self._d.get(_cint(index))

    def set(self, index):
        self._d.check_index(index)
        self._queue(_cint(index), 1)  # binBitsSetOne == a one-bit binBitsSetRange (binBits.c:67-80 vs :98-128)

    def clear(self, index):
        self._d.check_index(index)
        self._flush()
        self._touch()
        self._d.clear(_cint(index))

    def set_range(self, start, count):
        start = _cint(start)  # `int start` in the signature (bitset.pyx:216)
        self._d.check_range_count(start, count)
        count = _cint(count)
        if count:
            self._queue(start, count)

    def count_range(self, start, count):
        self._d.check_range_count(start, count)
        if self._ps:
            self._flush()
        return self._count(_cint(start), _cint(count), True)

    def next_set(self, start):
        self._d.check_index(start)
        return self._next(_cint(start), 1)

    def next_clear(self, start):
        self._d.check_index(start)
        return self._next(_cint(start), 0)

    @property
    def bin_size(self):
        return self._d.bin_size

    def _other(self, other):
        if not isinstance(other, BinnedBitSet):
            raise TypeError(
                "Argument 'other' has incorrect type (expected bx.bitset.BinnedBitSet, got %s)" % type(other).__name__
            )
        other._flush()
        return other._d

    def iand(self, other):
        o = self._other(other)
        self._d.check_same_size(o)
        self._flush()
        self._touch()
        self._d.iand(o)

    def ior(self, other):
        o = self._other(other)
        self._d.check_same_size(o)
        self._flush()
        self._touch()
        self._d.ior(o)

    def invert(self):
        self._flush()
        self._touch()
        self._d.invert()

    # ---- additive batch API (not in the reference) ---------------------------
    def set_ranges(self, starts, counts):
        self._flush()
        self._touch()
        self._d.set_ranges(starts, counts)

    def count_ranges(self, starts, counts):
        self._flush()
        return self._d.count_ranges(starts, counts)

    def runs(self, start=0):
        self._flush()
        return self._d.runs(start)

    def and_count(self, other):
        o = self._other(other)
        self._d.check_same_size(o)
        self._flush()
        self._touch()
        return self._d.and_count(o)


class BitSet(_Queued):
    """bitset.pyx:107-173 (flat, unbinned)"""

    def __init__(self, bitCount):
        self._init(DeviceBitSet(bitCount, flat=True))

    def _check_range(self, start, end):  # bitset.pyx:84-89
        self._d.check_index(start)
        if end < start:
            raise IndexError("Range end (%d) must be greater than range start(%d)." % (end, start))
        if end > self._d.size:
            raise IndexError("End %d is larger than the size of this BitSet (%d)." % (end, self._d.size))

    def set(self, index):
        self._d.check_index(index)
        self._queue(_cint(index), 1)

    def clear(self, index):
        self._d.check_index(index)
        self._flush()
        self._touch()
        self._d.clear(_cint(index))

    def clone(self):
        other = BitSet(self._d.size)
        other.ior(self)
        return other

    def set_range(self, start, count):
        self._d.check_range_count(start, count)
        start, count = _cint(start), _cint(count)
        if count:
            self._queue(start, count)

    def get(self, index):
        self._d.check_index(index)
        self._flush()
        return self._d.get(_cint(index))

    def count_range(self, start=0, count=None):
        if count is None:
            count = self._d.size - start
        self._d.check_range_count(start, count)
        if self._ps:
            self._flush()
        return self._count(_cint(start), _cint(count), False)

    def next_set(self, start, end=None):
        if end is None:
            end = self._d.size
        self._check_range(start, end)
        return min(self._next(_cint(start), 1), _cint(end))  # bitFind(..., bitCount=end), bits.c:143-176

    def next_clear(self, start, end=None):
        if end is None:
            end = self._d.size
        self._check_range(start, end)
        return min(self._next(_cint(start), 0), _cint(end))

    def _other(self, other):
        if not isinstance(other, BitSet):
            raise TypeError("Argument 'other' has incorrect type (expected bx.bitset.BitSet, got %s)" % type(other).__name__)
        other._flush()
        return other._d

    def iand(self, other):
        o = self._other(other)
        self._d.check_same_size(o)
        self._flush()
        self._touch()
        self._d.iand(o)

    def ior(self, other):
        o = self._other(other)
        self._d.check_same_size(o)
        self._flush()
        self._touch()
        self._d.ior(o)

    def ixor(self, other):
        o = self._other(other)
        self._d.check_same_size(o)
        self._flush()
        self._touch()
        self._d.ixor(o)

    def invert(self):
        self._flush()
        self._touch()
        self._d.invert()

    def __getitem__(self, index):
        return self.get(index)

    def __iand__(self, other):
        self.iand(other)
        return self

    def __ior__(self, other):
        self.ior(other)
        return self

    def __invert__(self):
        self.invert()
        return self
