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

from . import _ffi


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


class BitsetAccumulator:
    """State of one `binned_bitsets_from_file` run, so that several files can be fed in sequence
    (what fileinput does for bed_coverage.py / bed_merge_overlapping.py).  feed() queues ranges per chromosome,
    finish() creates the bitsets in first-appearance order, issues one set_ranges launch per queued block and
    re-raises the first error the reference would have hit."""

    def __init__(self, chrom_col=0, start_col=1, end_col=2, upstream_pad=0, downstream_pad=0, lens={}, bed_track_lines=False):
        self.cols = (chrom_col, start_col, end_col)
        self.upstream_pad, self.downstream_pad, self.lens, self.track = upstream_pad, downstream_pad, lens, bed_track_lines
        self.sizes = {}   # chrom -> size, in first-appearance order (drives dict order of the result)
        self.blocks = {}  # chrom -> [(starts, counts) int32 arrays] in file order
        self.starts, self.counts = {}, {}  # per-line rows not yet turned into a block
        self.error = None
        self.size = None  # `size` of the most recently created bitset (the padding code of the reference uses it)
        self.last_chrom = None
        self.offset = 0

    def _new_chrom(self, chrom, size):
        self.sizes[chrom] = size
        self.blocks[chrom] = []
        self.starts[chrom], self.counts[chrom] = [], []

    def _close_lists(self, chrom):
        if self.starts[chrom]:
            self.blocks[chrom].append((np.array(self.starts[chrom], dtype=np.int32), np.array(self.counts[chrom], dtype=np.int32)))
            self.starts[chrom], self.counts[chrom] = [], []

    def _bulk_prefix(self, f):
        """Fast path: parse a real file in C++ and queue every row up to the first one the reference would
        reject; returns the text lines that still have to go through the per-line loop."""
        from . import bedio

        data = bedio.file_bytes(f) if bedio.enabled() else None
        if data is None:
            return f
        bed = bedio.ParsedBed(data, *self.cols)
        try:
            size_of = np.array([self.sizes.get(c, self.lens[c] if c in self.lens else MAX) for c in bed.names] or [MAX], dtype=np.int64)
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
                if name not in self.sizes:
                    self._new_chrom(name, int(size_of[c]))
                    self.size = int(size_of[c])
                self._close_lists(name)
                keep = rows[e[rows] > s[rows]]
                self.blocks[name].append((s[keep].astype(np.int32), (e[keep] - s[keep]).astype(np.int32)))
            self.last_chrom = None
            return bed.rest_lines(k if k < bed.n else None, f)
        finally:
            bed.close()

    def feed(self, f):
        if self.error is not None:
            return self
        chrom_col, start_col, end_col = self.cols
        if not (self.upstream_pad or self.downstream_pad or self.track):
            f = self._bulk_prefix(f)
        for line in f:
            if line.startswith("#") or line.isspace():  # bitset_builders.py:33-34
                continue
            if self.track:  # bitset_builders.py:77-85: browser lines ignored, track lines may carry offset=N
                if line.startswith("browser"):
                    continue
                if line.startswith("track"):
                    m = re.search(r"offset=(\d+)", line)
                    if m and m.group(1):
                        self.offset = int(m.group(1))
                    continue
            try:
                fields = line.split()
                chrom = fields[chrom_col]
                if chrom != self.last_chrom:
                    if chrom not in self.sizes:
                        size = self.lens[chrom] if chrom in self.lens else MAX
                        if size > 2147483647:
                            raise ValueError("%d is larger than the maximum BinnedBitSet size of %d." % (size, 2147483647))
                        self._new_chrom(chrom, size)
                        self.size = size
                    self.last_chrom = chrom
                start, end = int(fields[start_col]) + self.offset, int(fields[end_col]) + self.offset
                if self.upstream_pad:
                    start = max(0, start - self.upstream_pad)
                if self.downstream_pad:
                    end = min(self.size, end + self.downstream_pad)  # `size` of the newest bitset, as in :48-49
                if start > end:
                    warn("Interval start after end!")
                if not -2147483648 <= start <= 2147483647:
                    raise OverflowError("value too large to convert to int")
                self.error = _check_range(self.sizes[chrom], start, end - start)
                if self.error is not None:
                    break
                if end > start:
                    self.starts[chrom].append(start)
                    self.counts[chrom].append(end - start)
            except (ValueError, IndexError, OverflowError) as ex:
                self.error = ex
                break
        return self

    def finish(self):
        bitsets = {}
        for chrom, sz in self.sizes.items():
            self._close_lists(chrom)
            b = BinnedBitSet(sz)
            for s, c in self.blocks[chrom]:
                if len(s):
                    b.set_ranges(s, c)
            bitsets[chrom] = b
        if self.error is not None:
            raise self.error
        return bitsets


def binned_bitsets_from_file(f, chrom_col=0, start_col=1, end_col=2, strand_col=5, upstream_pad=0, downstream_pad=0, lens={},
                             _bed_track_lines=False):
    """
    Read a file into a dictionary of bitsets (same arguments as the reference):
    - 'f' should be a file like object (or any iterable containing strings)
    - 'chrom_col', 'start_col', and 'end_col' must exist in each line.
    - if 'lens' is provided bitset sizes will be looked up from it, otherwise
      chromosomes will be assumed to be the maximum size
    """
    return BitsetAccumulator(chrom_col, start_col, end_col, upstream_pad, downstream_pad, lens, _bed_track_lines).feed(f).finish()


def binned_bitsets_from_paths(paths, bed_track_lines=False):
    """`binned_bitsets_from_file(fileinput.input(paths))` with every file going through the bulk ingest."""
    acc = BitsetAccumulator(bed_track_lines=bed_track_lines)
    for path in paths:
        with open(path) as f:
            acc.feed(f)
    return acc.finish()


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


def as_group(bitsets, mutated=False):
    """One BitSetGroup over the device sets behind a sequence of drop-in bitsets: their queued ranges are flushed, and with
    `mutated` their host-side caches of device results are dropped (the group is about to change them).  The group entry
    points run ONE kernel launch for all members where the reference's scripts loop over the chromosomes."""
    from .bitset import BitSetGroup

    members = list(bitsets)
    for b in members:
        b._flush()
        if mutated:
            b._touch()
    return BitSetGroup([b._d for b in members])


# A group launch works on whole word arrays: it makes every member allocate its full size (64 MiB for a default-sized set),
# where the per-set calls leave untouched bins unallocated.  Fine for a genome's chromosomes; a scaffold-level assembly
# with thousands of sequence names keeps the per-set loop (tests/test_gpu_bitset.py::test_thousands_of_default_sized_sets_stay_small).
# The gate is BYTES, not members: the word arrays the group would materialise (all operands) must fit a quarter of what
# the device has free right now, and a group that still runs out of memory falls back to the per-set loop.
GROUP_MAX_MEMBERS = 128
GROUP_MAX_FRACTION_OF_FREE = 0.25


def group_bytes(*operands):
    """Bytes of dense words a group launch over these sets would make resident (every member whole)."""
    return sum((b.size + 7) // 8 for members in operands for b in members)


def group_fits(*operands):
    """May these sets go through one group launch?  Few enough members, and their whole word arrays small against the
    device memory that is free now (bxmi_mem_info)."""
    if not operands or not operands[0] or any(len(m) > GROUP_MAX_MEMBERS for m in operands):
        return False
    free = _ffi.i64(0)
    _ffi.call("bxmi_mem_info", _ffi.C.byref(free), None)
    return group_bytes(*operands) <= GROUP_MAX_FRACTION_OF_FREE * free.value


def group_coverage(bitsets):
    """sum of count_range(0, size) over the sets -- bed_coverage.py:27-29 -- as one grid-wide popcount launch."""
    members = list(bitsets)
    if not members:
        return 0
    if group_fits(members):
        try:
            return int(as_group(members).popcounts().sum())
        except _ffi.BxmiError as e:  # the estimate was taken before the allocations: another process may have won the race
            if e.code != _ffi.ENOMEM:
                raise
    return sum(b.count_range(0, b.size) for b in members)


def group_iand(targets, others):
    """targets[i].iand(others[i]) for all i -- bed_intersect_basewise.py:25-28 -- as one launch."""
    targets, others = list(targets), list(others)
    if not targets:
        return
    if group_fits(targets, others):
        try:
            # (both groups exist before the first word changes: running out of memory leaves every set as it was)
            gt, go = as_group(targets, mutated=True), as_group(others)
            gt.iand(go)
            return
        except _ffi.BxmiError as e:
            if e.code != _ffi.ENOMEM:
                raise
    for a, b in zip(targets, others):
        a.iand(b)


def _c_int(v):
    if not -2147483648 <= v <= 2147483647:
        raise OverflowError("value too large to convert to int")
    return v


def _set_all(rows):
    """rows: iterable of (chrom, start, count) in call order, each what the reference hands to set_range on a
    MAX-sized bitset.  The first range the reference would reject raises (nothing is returned then, so the sets made
    before it are not observable); otherwise one set_ranges launch per chromosome."""
    per = {}
    for chrom, start, count in rows:
        g = per.get(chrom)
        if g is None:
            g = per[chrom] = ([], [])
        if start is None:  # chromosome seen, nothing to set
            continue
        err = _check_range(MAX, _c_int(start), count)  # `int start`, then the range check, then count -> C int
        if err is not None:
            raise err
        if _c_int(count):
            g[0].append(start), g[1].append(count)
    out = {}
    for chrom, (s, c) in per.items():
        b = out[chrom] = BinnedBitSet(MAX)
        if s:
            b.set_ranges(np.array(s, dtype=np.int32), np.array(c, dtype=np.int32))
    return out


def binned_bitsets_from_list(list=[]):
    """bitset_builders.py:142-156: rows of (chrom, start, end); every row is set_range(start, end - start)."""
    def rows():
        for r in list:
            chrom = r[0]
            start, end = int(r[1]), int(r[2])
            yield chrom, start, end - start
    return _set_all(rows())


def binned_bitsets_proximity(f, chrom_col=0, start_col=1, end_col=2, strand_col=5, upstream=0, downstream=0):
    """bitset_builders.py:107-139: each interval grown by `upstream` / `downstream` bases on the strand-aware side,
    clamped to [0, MAX]; empty and reversed intervals are dropped without a word."""
    def rows():
        for line in f:
            if line.startswith("#"):
                continue
            fields = line.split()
            minus = len(fields) >= strand_col + 1 and fields[strand_col] == "-"
            chrom = fields[chrom_col]
            start, end = int(fields[start_col]), int(fields[end_col])
            grow_start, grow_end = (downstream, upstream) if minus else (upstream, downstream)
            if grow_start:
                start = max(0, start - grow_start)
            if grow_end:
                end = min(MAX, end + grow_end)
            if end - start > 0:
                yield chrom, start, end - start
            else:
                yield chrom, None, None
    return _set_all(rows())


def binned_bitsets_by_chrom(f, chrom, chrom_col=0, start_col=1, end_col=2):
    """bitset_builders.py:159-169: ONE bitset holding the rows of `chrom`."""
    def rows():
        yield chrom, None, None
        for line in f:
            if line.startswith("#"):
                continue
            fields = line.split()
            if fields[chrom_col] == chrom:
                start, end = int(fields[start_col]), int(fields[end_col])
                yield chrom, start, end - start
    return _set_all(rows())[chrom]
