"""
BED text -> {chrom: BinnedBitSet}, the batch counterpart of the reference's
lib/bx/bitset_builders.py:17-54 (``binned_bitsets_from_file``).

Same observable behaviour -- skip rules, first-appearance chromosome order,
``lens`` lookup, padding arithmetic, the start>end warning, and the exception a
bad line raises -- but the per-line ``set_range`` calls are collected per
chromosome and issued as ONE ``set_ranges`` kernel launch each.
"""
import re
from warnings import warn

import numpy as np

from bx.bitset import MAX, BinnedBitSet


def _check_range(size, start, count):
    """bitset.pyx:184-189 (bb_check_range_count), evaluated at parse time so errors keep file order."""
    if start < 0:
        return IndexError("BitSet index (%d) must be non-negative." % start)
    if start >= size:
        return IndexError("%d is larger than the size of this BitSet (%d)." % (start, size))
    if count < 0:
        return IndexError("Count (%d) must be non-negative." % count)
    if start + count > size:
        return IndexError("End (%d) is larger than the size of this BinnedBitSet (%d)." % (start + count, size))
    return None


def _bulk_prefix(f, chrom_col, start_col, end_col, lens, sizes, parts):
    """Fast path: parse a real file in C++ and queue every row up to the first one the reference would
    reject; returns the text lines that still have to go through the per-line loop."""
    from . import bedio

    data = bedio.file_bytes(f) if bedio.enabled() else None
    if data is None:
        return f
    bed = bedio.ParsedBed(data, chrom_col, start_col, end_col)
    try:
        size_of = np.array([lens[c] if c in lens else MAX for c in bed.names] or [MAX], dtype=np.int64)
        sz = size_of[bed.chrom] if bed.n else np.empty(0, np.int64)
        s, e = bed.start, bed.end
        bad = (sz > 2147483647) | (s < 0) | (s >= sz) | (e < s) | (e > sz)  # includes start beyond int32
        k = int(np.argmax(bad)) if bad.any() else bed.n
        ids = bed.chrom[:k]
        order = np.argsort(ids, kind="stable")
        bounds = np.searchsorted(ids[order], np.arange(len(bed.names) + 1))
        first_row = np.full(len(bed.names), bed.n, dtype=np.int64)
        if k:
            np.minimum.at(first_row, ids, np.arange(k))
        for c in np.argsort(first_row, kind="stable").tolist():  # chromosomes in first-appearance order
            if first_row[c] >= k:
                break
            rows = order[bounds[c]:bounds[c + 1]]
            name = bed.names[c]
            sizes[name] = int(size_of[c])
            keep = rows[e[rows] > s[rows]]
            parts[name] = [(s[keep].astype(np.int32), (e[keep] - s[keep]).astype(np.int32))]
        return bed.rest_lines(k if k < bed.n else None)
    finally:
        bed.close()


def binned_bitsets_from_file(f, chrom_col=0, start_col=1, end_col=2, strand_col=5, upstream_pad=0, downstream_pad=0, lens={},
                             _bed_track_lines=False):
    """
    Read a file into a dictionary of bitsets (same arguments as the reference):
    - 'f' should be a file like object (or any iterable containing strings)
    - 'chrom_col', 'start_col', and 'end_col' must exist in each line.
    - if 'lens' is provided bitset sizes will be looked up from it, otherwise
      chromosomes will be assumed to be the maximum size
    """
    sizes = {}  # chrom -> size, in first-appearance order (drives dict order of the result)
    parts = {}  # chrom -> [(starts, counts) arrays] queued by the bulk prefix
    starts, counts = {}, {}
    error = None
    size = None
    last_chrom = None
    offset = 0
    if not (upstream_pad or downstream_pad or _bed_track_lines):
        f = _bulk_prefix(f, chrom_col, start_col, end_col, lens, sizes, parts)
        for chrom in sizes:
            starts[chrom], counts[chrom] = [], []
        if sizes:
            size = sizes[next(reversed(sizes))]
    for line in f:
        if line.startswith("#") or line.isspace():  # bitset_builders.py:33-34
            continue
        if _bed_track_lines:  # bitset_builders.py:77-85: browser lines ignored, track lines may carry offset=N
            if line.startswith("browser"):
                continue
            if line.startswith("track"):
                m = re.search(r"offset=(\d+)", line)
                if m and m.group(1):
                    offset = int(m.group(1))
                continue
        try:
            fields = line.split()
            chrom = fields[chrom_col]
            if chrom != last_chrom:
                if chrom not in sizes:
                    size = lens[chrom] if chrom in lens else MAX
                    if size > 2147483647:
                        raise ValueError("%d is larger than the maximum BinnedBitSet size of %d." % (size, 2147483647))
                    sizes[chrom] = size
                    starts[chrom], counts[chrom] = [], []
                last_chrom = chrom
            start, end = int(fields[start_col]) + offset, int(fields[end_col]) + offset
            if upstream_pad:
                start = max(0, start - upstream_pad)
            if downstream_pad:
                end = min(size, end + downstream_pad)  # `size` of the most recently created bitset, as in :48-49
            if start > end:
                warn("Interval start after end!")
            if not -2147483648 <= start <= 2147483647:
                raise OverflowError("value too large to convert to int")
            error = _check_range(sizes[chrom], start, end - start)
            if error is not None:
                break
            if end > start:
                starts[chrom].append(start)
                counts[chrom].append(end - start)
        except (ValueError, IndexError, OverflowError) as ex:
            error = ex
            break
    bitsets = {}
    for chrom, sz in sizes.items():
        b = BinnedBitSet(sz)
        for s, c in parts.get(chrom, ()):
            if len(s):
                b.set_ranges(s, c)
        if starts[chrom]:
            b.set_ranges(np.array(starts[chrom], dtype=np.int32), np.array(counts[chrom], dtype=np.int32))
        bitsets[chrom] = b
    if error is not None:
        raise error
    return bitsets


def binned_bitsets_from_bed_file(f, chrom_col=0, start_col=1, end_col=2, strand_col=5, upstream_pad=0, downstream_pad=0, lens={}):
    """bitset_builders.py:57-104: like binned_bitsets_from_file, but `browser` lines are skipped and
    `track ... offset=N` lines shift every following interval by N."""
    return binned_bitsets_from_file(f, chrom_col, start_col, end_col, strand_col, upstream_pad, downstream_pad, lens,
                                    _bed_track_lines=True)


def write_runs(out, chrom, bits, clip=None):
    """The `start = next_set(end); end = next_clear(start)` walk of the basewise scripts as one
    device run extraction.  Raises where the reference's loop would (a run reaching `size` makes
    it call next_set(size), bitset.pyx:180-181).  `clip` = bed_complement.py's chromosome length."""
    starts, ends = bits.runs()
    w = out.write
    for s, e in zip(starts.tolist(), ends.tolist()):
        if clip is not None and e > clip:
            e = clip
        w("%s\t%d\t%d\n" % (chrom, s, e))
        if clip is not None and e == clip:
            return
    if len(ends) and ends[-1] == bits.size:
        out.flush()
        raise IndexError("%d is larger than the size of this BitSet (%d)." % (bits.size, bits.size))


def binned_bitsets_from_list(rows):
    """bitset_builders.py:142-156: rows of (chrom, start, end)."""
    return binned_bitsets_from_file(("%s\t%s\t%s\n" % (r[0], int(r[1]), int(r[2])) for r in rows))
