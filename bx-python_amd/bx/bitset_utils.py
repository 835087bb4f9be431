"""
bx.bitset_utils -- lists of (start, end) treated as bitsets (the reference's lib/bx/bitset_utils.py:12-90), on the device
engine: a list becomes ONE queued batch of set_range calls (flushed by one launch), a bitset comes back as ONE run extraction
instead of a next_set / next_clear round trip per interval.  Same names, results, result order and failures -- including what
a walk does when a run reaches the end of the set (the reference then asks next_set(size) and gets IndexError).
"""
from bx.bitset import MAX, BinnedBitSet


def list2bits(ex):
    """lib/bx/bitset_utils.py:28-32: every (start, end) is set_range(start, end - start) on a MAX-sized set."""
    bits = BinnedBitSet(MAX)
    put = bits.set_range
    for start, end in ex:
        put(start, end - start)
    return bits


def _runs_from(bits, begin):
    """The maximal runs of set bits that end after `begin`, in order, as two lists; the first one is clipped to `begin`
    exactly as next_set(begin) would report it."""
    if begin >= bits.size:
        return [], []
    runs = getattr(bits, "runs", None)
    if runs is None:
        # any next_set / next_clear duck type (the flat bx.bitset.BitSet): the reference's own walk, lib/bx/bitset_utils.py:35-44
        rs, re, pos = [], [], begin
        while True:
            s = bits.next_set(pos)
            if s >= bits.size:
                break
            e = bits.next_clear(s)
            rs.append(int(s))
            re.append(int(e))
            if e >= bits.size:
                break
            pos = e
        return rs, re
    rs, re = runs(begin)
    return [int(x) for x in rs], [int(x) for x in re]


def bits2list(bits):
    """lib/bx/bitset_utils.py:35-44.  A run that reaches bits.size makes the reference's loop call next_set(size): IndexError."""
    rs, re = _runs_from(bits, 0)
    if re and re[-1] == bits.size:
        bits.next_set(bits.size)  # (raises what the reference raises)
    return list(zip(rs, re))


def bitset_intersect(ex1, ex2):
    """lib/bx/bitset_utils.py:12-16"""
    bits1, bits2 = list2bits(ex1), list2bits(ex2)
    bits1.iand(bits2)
    return bits2list(bits1)


def bitset_subtract(ex1, ex2):
    """lib/bx/bitset_utils.py:19-25"""
    bits1, bits2 = list2bits(ex1), list2bits(ex2)
    bits2.invert()
    bits1.iand(bits2)
    return bits2list(bits1)


def bitset_union(exons):
    """lib/bx/bitset_utils.py:88-90"""
    return bits2list(list2bits(exons))


def bitset_complement(exons):
    """lib/bx/bitset_utils.py:47-70: the gaps between the intervals, inside [smallest start, largest end) only."""
    bits = list2bits(exons)
    bits.invert()
    lo = min(a[0] for a in exons)  # (an empty list fails here, as the reference's does)
    hi = max(a[1] for a in exons)
    introns = []
    rs, re = _runs_from(bits, lo)
    pos = lo
    for a, b in zip(rs, re):
        if b <= pos:
            continue
        start = a if a > pos else pos
        end = b if b < hi else hi
        if start != end:
            introns.append((start, end))
        if end == hi:
            return introns
        pos = end
    # (no further set bit: the reference's next_set answers `size` and the walk ends -- or raises, standing at the very end)
    if pos >= bits.size:
        bits.next_set(pos)
    return introns


def bitset_interval_intersect(bits, istart, iend):
    """lib/bx/bitset_utils.py:73-85: the runs of `bits` met walking from istart while they START before iend -- a run that begins
    before iend is reported whole (only its start is clipped to istart), as the reference reports it."""
    rval = []
    pos = istart
    rs, re = _runs_from(bits, istart)
    for a, b in zip(rs, re):
        if b <= pos:
            continue
        start = a if a > pos else pos
        if start >= iend:
            return rval
        rval.append((start, b))  # (a run is never empty)
        if b >= iend:
            return rval
        pos = b
    # no further set bit: the reference's next_set returns size (>= iend ends the loop) -- unless the walk stands at the very end
    # of the set, where next_set raises
    if pos >= bits.size:
        bits.next_set(pos)
    if iend > bits.size:
        # next_set answered `size`, which is still below iend: the reference goes on to next_clear(size) -- IndexError
        bits.next_clear(bits.size)
    return rval
