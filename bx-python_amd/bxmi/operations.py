"""
The Galaxy-level interval operations on the MI355X engine.

Counterparts of lib/bx/intervals/operations/{__init__,intersect,subtract,coverage,merge,complement,
base_coverage}.py: same call signatures, the same sequence of yielded objects (Header / Comment /
GenomicInterval / field lists), the same skip bookkeeping on the readers.  What changes is the shape of
the work.  The reference walks the primary file row by row and asks the bitset three or more questions
per row (`count_range`, then `next_set` / `next_clear` in a loop).  Here the primary is read once, and per
chromosome the engine answers all rows together:

  * covered bases of every row        -> one `bxmi_bits_count_ranges` launch,
  * the set runs of the secondary set -> one `bxmi_bits_runs` scan,
  * which runs each row overlaps      -> one batched `find` on an interval index over the runs,

and the pieces are clipped from those answers.  Consequences a caller can see: the primary reader is
drained on the first `next()` (the reference drains it lazily), and the skips an operation itself records
on the primary appear then -- interleaved with the reader's own skips in file order, as in the reference.

The per-call generators `bits_set_in_range` / `bits_clear_in_range` are kept for callers that use them
directly (operations/__init__.py:10-33).
"""
from warnings import warn

import numpy as np

from bx.bitset import MAX

from .genomic import BitsetSafeReaderWrapper, Comment, GenomicInterval, Header
from .intervals import IntervalIndex

BED_DEFAULT_COLS = 0, 1, 2, 5
MAX_END = 512 * 1024 * 1024


# ------------------------------------------------------------------ per-call generators --
def bits_set_in_range(bits, range_start, range_end):
    """(start, end) of every span of set bits inside [range_start, range_end) (operations/__init__.py:10-20).
    Like the reference it raises IndexError when the scan runs off the end of the bitset."""
    pos = range_start
    while True:
        first = bits.next_set(pos)
        pos = min(bits.next_clear(first), range_end)
        if first >= pos:
            return
        yield first, pos


def bits_clear_in_range(bits, range_start, range_end):
    """(start, end) of every span of clear bits inside [range_start, range_end) (operations/__init__.py:23-33)."""
    pos = range_start
    while True:
        first = bits.next_clear(pos)
        if first >= range_end:
            return
        pos = min(bits.next_set(first), range_end)
        yield first, pos


# ------------------------------------------------------------------ batched answers --
def _past_end_message(size):
    # what next_set(size) / next_clear(size) raise (bitset.pyx:176-180)
    return "%d is larger than the size of this BitSet (%d)." % (size, size)


class _RunTable:
    """Set runs of one bitset (one device scan) and a device interval index over them."""

    def __init__(self, bits):
        rs, re = bits.runs(0)
        self.rs = np.asarray(rs, dtype=np.int64)
        self.re = np.asarray(re, dtype=np.int64)
        self.size = bits.size
        self.index = None
        if len(self.rs):
            self.index = IntervalIndex()
            self.index.append(self.rs.astype(np.int32), np.minimum(self.re, 2**31 - 1).astype(np.int32))
            self.index.seal()

    def overlaps(self, a, b):
        """CSR (offsets, run ids) of the runs overlapping each [a_i, b_i); rows with a_i >= b_i get none."""
        n = len(a)
        offsets = np.zeros(n + 1, dtype=np.int64)
        proper = np.nonzero(a < b)[0]
        if self.index is None or len(proper) == 0:
            return offsets, np.empty(0, dtype=np.int64)
        off, hits = self.index.find(a[proper].astype(np.int32), np.minimum(b[proper], 2**31 - 1).astype(np.int32))
        counts = np.zeros(n, dtype=np.int64)
        counts[proper] = np.diff(off)
        offsets[1:] = np.cumsum(counts)
        return offsets, hits.astype(np.int64)

    def set_pieces(self, a, b):
        """Per row: the pieces bits_set_in_range would yield, and whether it would then run off the end
        (no set bit after the last piece: `next_set` answers `size`, `next_clear(size)` raises)."""
        offsets, ids = self.overlaps(a, b)
        row = np.repeat(np.arange(len(a)), np.diff(offsets))
        ps = np.maximum(self.rs[ids], a[row])
        pe = np.minimum(self.re[ids], b[row])
        nruns = len(self.rs)
        n = np.diff(offsets)
        last_set = self.re[-1] if nruns else 0
        off_end = np.empty(len(a), dtype=bool)
        none = n == 0
        off_end[none] = (a[none] >= last_set) if nruns else True  # nothing set at or after a
        some = ~none
        if some.any():
            last = ids[offsets[1:][some] - 1]
            off_end[some] = (last == nruns - 1) & (self.re[last] <= b[some])
        return offsets, ps, pe, off_end

    def clear_pieces(self, a, b):
        """Per row: the pieces bits_clear_in_range would yield, and whether it would then call next_clear(size)."""
        offsets, ids = self.overlaps(a, b)
        n = np.diff(offsets)
        row = np.repeat(np.arange(len(a)), n)
        cs = np.maximum(self.rs[ids], a[row])
        ce = np.minimum(self.re[ids], b[row])
        # candidate gaps of row i: [a, cs_0), [ce_0, cs_1), ..., [ce_last, b)  -> n_i + 1 of them
        goff = np.zeros(len(a) + 1, dtype=np.int64)
        goff[1:] = np.cumsum(n + 1)
        total = int(goff[-1])
        gs = np.empty(total, dtype=np.int64)
        ge = np.empty(total, dtype=np.int64)
        first = goff[:-1]
        lastp = goff[1:] - 1
        gs[first] = a
        ge[lastp] = b
        inner = np.arange(len(ids)) + row  # run k of row i sits between candidate k and k + 1 of that row
        ge[inner] = cs
        gs[inner + 1] = ce
        keep = gs < ge
        grow = np.repeat(np.arange(len(a)), n + 1)[keep]
        poff = np.zeros(len(a) + 1, dtype=np.int64)
        poff[1:] = np.cumsum(np.bincount(grow, minlength=len(a)))
        tail_clear = len(self.rs) == 0 or self.re[-1] < self.size
        off_end = (a < b) & (b == self.size) & tail_clear
        return poff, gs[keep], ge[keep], off_end


class _Items:
    """The primary's items in order.  The natively parsed prefix (bxmi.tabio) is kept as arrays and a row object is only
    made when somebody asks for it -- an interval that overlaps nothing is never built; what follows the prefix, or
    everything when there is no prefix, are the objects the reader delivered."""

    def __init__(self, reader, bulk):
        self.reader, self.bulk = reader, bulk
        self.nb = bulk.n if bulk is not None else 0
        self.objs = [None] * self.nb
        self.tail_where = []
        if bulk is not None:  # headers and comments are few: made now (the header through the reader, which keeps it)
            for i in np.nonzero(bulk.kind_a != 0)[0].tolist():
                kind = bulk.kind[i]
                self.objs[i] = reader.header if kind == 3 else (Comment("") if kind == 1 else reader.parse_comment(bulk.line(i)))

    def __len__(self):
        return len(self.objs)

    def __getitem__(self, i):
        item = self.objs[i]
        if item is None:
            item = self.objs[i] = self.reader._bulk_row(self.bulk, i)
        return item

    def __iter__(self):
        return (self[i] for i in range(len(self.objs)))

    def where(self, i):
        """(line number, raw line) the reader was at when it delivered item i."""
        if i < self.nb:
            return i + 1, self.bulk.raw_line(i)
        return self.tail_where[i - self.nb]


class _Primary:
    """The primary reader, drained: items in order with the line bookkeeping the reference would have seen."""

    def __init__(self, reader):
        self.reader = reader
        self.tracks = hasattr(reader, "skip_log") and hasattr(reader, "delivered")
        self.log_from = len(reader.skip_log) if self.tracks else 0
        self.skipped0 = getattr(reader, "skipped", None)
        self.lines0 = list(getattr(reader, "skipped_lines", []) or [])
        self.base = reader.delivered if self.tracks else 0
        take = getattr(reader, "_bulk_take", None)
        self.bulk = take() if take is not None else None
        self.items = _Items(reader, self.bulk)
        for item in reader:
            self.items.objs.append(item)
            self.items.tail_where.append((getattr(reader, "linenum", None), getattr(reader, "current_line", None)))
        self.own = []  # (item index, message)

    def skip(self, i, message):
        self.own.append((i, message))

    def settle(self):
        """Replay the operation's own skips into the reader in file order
        (`primary.skipped += 1; if primary.skipped < 10: primary.skipped_lines.append(...)`, e.g. intersect.py:52-61)."""
        r = self.reader
        if not self.own or self.skipped0 is None or not hasattr(r, "skipped_lines"):
            return
        self.own.sort(key=lambda t: t[0])
        events = [(self.base + i + 0.25, self.items.where(i) + (msg,)) for i, msg in self.own]
        if self.tracks:
            events += [(d - 0.5, entry) for d, entry in r.skip_log[self.log_from:]]
        events.sort(key=lambda t: t[0])
        skipped, lines = self.skipped0, list(self.lines0)
        for _, entry in events:
            skipped += 1
            if skipped < 10:
                lines.append(entry)
        r.skipped = skipped
        r.skipped_lines[:] = lines
        if self.tracks:
            r.skip_log[self.log_from:] = [(int(t + 0.75), e) for t, e in events]


def _first_safe_bitsets(reader, lens, **kw):
    safe = BitsetSafeReaderWrapper(reader, lens=lens)
    return safe, safe.binned_bitsets(lens=lens, **kw)


def _range_error(bits, start, end):
    """The IndexError text count_range(start, end - start) would raise, or None (bitset.pyx:176-189)."""
    try:
        bits._d.check_range_count(start, end - start)
    except IndexError as e:
        return str(e)
    return None


def _rows_by_chrom(primary, bitsets, start_after_end):
    """Valid interval rows grouped by chromosome: {chrom: (item indices, starts, ends)}; invalid ones are logged."""
    groups = {}
    nb = 0
    b = primary.bulk
    if b is not None:
        # the natively parsed prefix: rows are valid by construction (start <= end), so only the range check is left
        nb = b.n
        rows = np.nonzero(b.kind_a == 0)[0]
        cid, st, en = b.chrom_a[rows], b.start_a[rows], b.end_a[rows]
        for c, name in enumerate(b.names):
            if name not in bitsets:
                continue
            sel = cid == c
            if not sel.any():
                continue
            ri, rs, re_ = rows[sel], st[sel], en[sel]
            size = bitsets[name].size
            bad = (rs < 0) | (rs >= size) | (re_ > size)
            for k in np.nonzero(bad)[0].tolist():
                primary.skip(int(ri[k]), _range_error(bitsets[name], int(rs[k]), int(re_[k])))
            ok = ~bad
            groups[name] = ([ri[ok]], [rs[ok]], [re_[ok]])
        # (dict order = first appearance among the rows that made it, as the loop below would have given)
        firsts = sorted((int(g[0][0][0]) if len(g[0][0]) else -1, name) for name, g in groups.items())
        groups = {name: groups[name] for f, name in firsts if f >= 0}
    for i in range(nb, len(primary.items)):
        item = primary.items[i]
        if not isinstance(item, GenomicInterval):
            continue
        chrom = item.chrom
        if chrom not in bitsets:
            continue
        start, end = int(item.start), int(item.end)
        if start > end and start_after_end == "skip":
            primary.skip(i, "Interval start after end!")
            continue
        if start > end and start_after_end == "drop":  # (the caller has dealt with the row already)
            continue
        if start > end and start_after_end == "warn":
            warn("Interval start after end!")
        err = _range_error(bitsets[chrom], start, end)
        if err is not None:
            primary.skip(i, err)
            continue
        g = groups.setdefault(chrom, ([], [], []))
        g[0].append(i), g[1].append(start), g[2].append(end)

    def column(parts):  # arrays of the parsed prefix first (if any), then the plain numbers of the rest
        arrays = [np.asarray(x, dtype=np.int64) if isinstance(x, np.ndarray) else None for x in parts]
        lead = [a for a in arrays if a is not None]
        rest = np.array([x for x, a in zip(parts, arrays) if a is None], dtype=np.int64)
        return np.concatenate(lead + [rest]) if lead else rest

    return {c: (column(g[0]), column(g[1]), column(g[2])) for c, g in groups.items()}


def _covered(bits, starts, ends):
    return np.asarray(bits.count_ranges(starts.astype(np.int32), (ends - starts).astype(np.int32)), dtype=np.int64)


def _emit(primary, comments, per_item, passthrough=None):
    """Yield in primary order: headers, comments, and for interval rows whatever `per_item` holds."""
    b = primary.bulk
    nb = b.n if b is not None else 0
    if nb:
        # in the parsed prefix only headers, comments, rows with output and (passthrough) rows of unknown chromosomes matter
        want = b.kind_a != 0
        if per_item:
            idx = np.fromiter((i for i in per_item if i < nb), dtype=np.int64)
            want[idx] = True
        if passthrough is not None:
            for c, name in enumerate(b.names):
                probe = GenomicInterval._from_parsed(primary.reader, "", name, 0, 0, "+")
                if passthrough(probe):
                    want |= (b.kind_a == 0) & (b.chrom_a == c)
        todo = np.nonzero(want)[0].tolist()
    else:
        todo = []
    for i in todo + list(range(nb, len(primary.items))):
        item = primary.items[i]
        if isinstance(item, Header):
            yield item
        if isinstance(item, Comment) and comments:
            yield item
        elif isinstance(item, GenomicInterval):
            if passthrough is not None and passthrough(item):
                yield item
                continue
            todo_i = per_item.get(i, ())
            if type(item) is GenomicInterval:
                if len(todo_i) == 1 and todo_i[0] == (item.start, item.end):
                    # the row survives whole: the reference yields `interval.copy()` with the same start and end -- an equal
                    # row nobody else holds -- so the row itself will do (its fields are in normal form already), and a
                    # row that came from the native parser is then printed without ever being split
                    yield item
                else:
                    for start, end in todo_i:
                        yield item._piece(start, end)
            else:
                for start, end in todo_i:
                    yield _copy_with(item, start, end)


def _copy_with(item, start, end):
    piece = item.copy()  # any other row class: the reference's three steps (intersect.py:69-72)
    piece.start = start
    piece.end = end
    return piece


def _pieces_or_whole(primary, bitsets, groups, mincols, pieces, want_set):
    """intersect (want_set) / subtract: per item index the list of (start, end) to emit."""
    out = {}
    for chrom, (idx, starts, ends) in groups.items():
        bits = bitsets[chrom]
        covered = _covered(bits, starts, ends)
        enough = covered >= mincols
        if want_set:
            cut = enough if pieces else np.zeros(len(idx), dtype=bool)
            whole = enough & ~cut
        else:
            cut = enough if pieces else np.zeros(len(idx), dtype=bool)
            whole = ~enough  # subtract.py:59-63: too little overlap -> the row survives whole; enough and not pieces -> nothing
        for k in np.nonzero(whole)[0]:
            out[int(idx[k])] = [(int(starts[k]), int(ends[k]))]
        sel = np.nonzero(cut)[0]
        if len(sel):
            table = _RunTable(bits)
            fn = table.set_pieces if want_set else table.clear_pieces
            off, ps, pe, off_end = fn(starts[sel], ends[sel])
            for j, k in enumerate(sel):
                lo, hi = int(off[j]), int(off[j + 1])
                out[int(idx[k])] = list(zip(ps[lo:hi].tolist(), pe[lo:hi].tolist()))
                if off_end[j]:
                    primary.skip(int(idx[k]), _past_end_message(bits.size))
    return out


# ------------------------------------------------------------------ the operations --
def intersect(readers, mincols=1, upstream_pad=0, downstream_pad=0, pieces=True, lens={}, comments=True):
    """operations/intersect.py:21-83.  readers[0] is kept, restricted to what the AND of readers[1:] covers."""
    primary = readers[0]
    _, bitsets = _first_safe_bitsets(readers[1], lens, upstream_pad=upstream_pad, downstream_pad=downstream_pad)
    for other in readers[2:]:
        more = other.binned_bitsets(upstream_pad=upstream_pad, downstream_pad=downstream_pad, lens=lens)
        for chrom in bitsets:
            if chrom in more:
                bitsets[chrom].iand(more[chrom])
    p = _Primary(primary)
    groups = _rows_by_chrom(p, bitsets, "skip")
    out = _pieces_or_whole(p, bitsets, groups, mincols, pieces, want_set=True)
    p.settle()
    yield from _emit(p, comments, out)


def subtract(readers, mincols=1, upstream_pad=0, downstream_pad=0, pieces=True, lens={}, comments=True):
    """operations/subtract.py:22-77.  readers[0] minus the OR of readers[1:]."""
    primary = readers[0]
    _, bitsets = _first_safe_bitsets(readers[1], lens, upstream_pad=upstream_pad, downstream_pad=downstream_pad)
    for other in readers[2:]:
        more = other.binned_bitsets(upstream_pad=upstream_pad, downstream_pad=downstream_pad, lens=lens)
        for chrom in more:
            if chrom not in bitsets:
                bitsets[chrom] = more[chrom]
            else:
                bitsets[chrom].ior(more[chrom])
    p = _Primary(primary)
    groups = _rows_by_chrom(p, bitsets, "warn")
    out = _pieces_or_whole(p, bitsets, groups, mincols, pieces, want_set=False)
    p.settle()
    yield from _emit(p, comments, out, passthrough=lambda item: item.chrom not in bitsets)


def coverage(readers, comments=True):
    """operations/coverage.py:17-77.  Appends 'bases covered' and 'fraction covered' to every row of readers[0]."""
    primary = readers[0]
    _, bitsets = _first_safe_bitsets(readers[1], {})
    for other in readers[2:]:
        more = other.binned_bitsets()
        for chrom in bitsets:
            if chrom in more:
                bitsets[chrom].ior(more[chrom])
    p = _Primary(primary)
    n_items = len(p.items)
    have = np.zeros(n_items, dtype=bool)        # rows that are written out
    bases = np.zeros(n_items, dtype=np.int64)
    length = np.full(n_items, -1, dtype=np.int64)  # -1: chromosome without a bitset -> "0", "0.0"
    b = p.bulk
    nb = b.n if b is not None else 0
    if nb:  # the natively parsed prefix: every row is valid (start <= end); those of unknown chromosomes are written with zeros
        known = np.array([name in bitsets for name in b.names] or [False])
        rows = b.kind_a == 0
        have[:nb] = rows & ~known[np.where(rows, b.chrom_a, 0)]
    for i in range(nb, n_items):  # what the reader delivered object by object: coverage.py:36-62, row by row
        item = p.items[i]
        if not isinstance(item, GenomicInterval):
            continue
        if int(item.start) > int(item.end):
            p.skip(i, "Interval start after end!")
        elif item.chrom not in bitsets:
            have[i] = True
    for chrom, (idx, starts, ends) in _rows_by_chrom(p, bitsets, "drop").items():
        have[idx] = True
        bases[idx] = _covered(bitsets[chrom], starts, ends)
        length[idx] = ends - starts
    for i, _ in p.own:  # (a row the range check refused is logged, not written)
        have[i] = False
    p.settle()
    bases_l, length_l = bases.tolist(), length.tolist()
    want = have.copy()
    if nb:
        want[:nb] |= b.kind_a != 0
    want[nb:] = True
    for i in np.nonzero(want)[0].tolist():
        item = p.items[i]
        if isinstance(item, Header):
            yield item
        if isinstance(item, Comment) and comments:
            yield item
        elif isinstance(item, GenomicInterval) and have[i]:
            c, ln = bases_l[i], length_l[i]
            fraction = 0.0 if ln < 0 else (0 if ln == 0 else float(c) / float(ln))
            if type(item) is GenomicInterval:
                item._append_fields(str(c), str(fraction))
            else:
                item.fields.append(str(c))
                item.fields.append(str(fraction))
            yield item


def _note_reader_skip(reader, message):
    # merge.py:27-35 / complement.py:50-57 / find_clusters.py:33-40:
    # `reader.skipped += 1; if reader.skipped < 10: skipped_lines.append((linenum, current_line, message))`, errors ignored
    try:
        if hasattr(reader, "note_skip"):
            reader.note_skip(reader.linenum, reader.current_line, message)
        else:
            reader.skipped += 1
            if reader.skipped < 10:
                reader.skipped_lines.append((reader.linenum, reader.current_line, message))
    except Exception:
        pass


def merge(interval, mincols=1):
    """operations/merge.py:13-37.  One row per run of the union, as a list of column strings; like the reference the
    SAME list object is yielded again and again for a chromosome, and every chromosome ends with a logged skip
    (the scan always runs off the end of the bitset)."""
    reader = BitsetSafeReaderWrapper(interval, lens={})
    bitsets = reader.binned_bitsets()
    if reader.header:
        yield reader.header
    width = max(reader.chrom_col, reader.start_col, reader.end_col) + 1
    for chrom, bits in bitsets.items():
        row = ["."] * width
        row[reader.chrom_col] = chrom
        rs, re = bits.runs(0)
        for start, end in zip(np.asarray(rs).tolist(), np.asarray(re).tolist()):
            row[reader.start_col] = str(start)
            row[reader.end_col] = str(min(end, MAX_END))
            yield row
        _note_reader_skip(reader, _past_end_message(bits.size))


def complement(reader, lens):
    """operations/complement.py:13-57.  The uncovered stretches of every chromosome seen, up to lens[chrom] (or MAX)."""
    safe, bitsets = _first_safe_bitsets(reader, lens, upstream_pad=0, downstream_pad=0)
    for bits in bitsets.values():
        bits.invert()
    width = max(safe.chrom_col, safe.start_col, safe.end_col) + 1
    for chrom, bits in bitsets.items():
        limit = lens.get(chrom, MAX)
        rs, re = bits.runs(0)
        for start, end in zip(np.asarray(rs).tolist(), np.asarray(re).tolist()):
            if start >= limit:
                break
            fields = ["."] * width
            if chrom == chrom.strip() and start <= min(end, limit):
                yield GenomicInterval._from_values(safe, fields, chrom, start, min(end, limit), "+")
                continue
            if 0 <= safe.strand_col < len(fields):  # anything unusual: through the constructor, which says what is wrong
                fields[safe.strand_col] = "+"
            fields[safe.chrom_col] = chrom
            fields[safe.start_col] = start
            fields[safe.end_col] = min(end, limit)
            yield GenomicInterval(safe, fields, safe.chrom_col, safe.start_col, safe.end_col, safe.strand_col, "+")


def find_clusters(reader, mincols=1, minregions=2):
    """operations/find_clusters.py:20-41: one ClusterTree per chromosome keyed by 0-based item number; everything that is
    not an interval row is returned in `extra`.  The trees answer on the device when asked."""
    from bx.intervals.cluster import ClusterTree

    extra, chroms = {}, {}
    for linenum, item in enumerate(reader):
        if not isinstance(item, GenomicInterval):
            extra[linenum] = item
            continue
        tree = chroms.get(item.chrom)
        if tree is None:
            tree = chroms[item.chrom] = ClusterTree(mincols, minregions)
        try:
            tree.insert(item.start, item.end, linenum)
        except OverflowError as e:
            _note_reader_skip(reader, str(e))
    return chroms, extra


def join(leftSet, rightSet, mincols=1, leftfill=True, rightfill=True):
    """operations/join.py:14-75: every left row is paired with the right rows it overlaps by at least `mincols`
    (left fields + right fields); unmatched rows are padded with "." when the fill flags ask for it.

    The reference reports the matches of one left row in the pre-order of a treap with RANDOM priorities
    (quicksect.py:39-44,113-119), i.e. in no reproducible order; here they come in the index's order (by start).
    Everything else -- which rows, the overlap arithmetic with its inclusive-range quirks (:36-50), the fill rows and
    the final in-order list of never-matched right rows (ties: later line first, quicksect.py:50-63) -- is identical.
    One batched `find` per chromosome replaces the per-row tree walks."""
    right = {}  # chrom -> (starts, ends, fields lists), in file order
    rightlen = 0
    for item in rightSet:
        if isinstance(item, GenomicInterval):
            g = right.setdefault(item.chrom, ([], [], []))
            g[0].append(item.start), g[1].append(item.end), g[2].append(item.fields)
            if rightlen == 0:
                rightlen = item.nfields
    left = list(leftSet)
    leftlen = 0
    by_chrom = {}
    for i, item in enumerate(left):
        if isinstance(item, GenomicInterval):
            if leftlen == 0:
                leftlen = item.nfields
            if item.chrom in right:
                by_chrom.setdefault(item.chrom, []).append(i)
    matches = {}  # left item index -> list of (chrom, right row number)
    for chrom, rows in by_chrom.items():
        rs, re, _ = right[chrom]
        index = IntervalIndex()
        index.append(np.array(rs, dtype=np.int64).astype(np.int32), np.array(re, dtype=np.int64).astype(np.int32))
        qs = np.array([left[i].start for i in rows], dtype=np.int64)
        qe = np.array([left[i].end for i in rows], dtype=np.int64)
        off, hits = index.find(qs.astype(np.int32), qe.astype(np.int32))
        index.close()
        hs, he = np.array(rs, dtype=np.int64)[hits], np.array(re, dtype=np.int64)[hits]
        row = np.repeat(np.arange(len(rows)), np.diff(off))
        a, b = qs[row], qe[row]
        s_in = (hs >= a) & (hs <= b)  # `item.start in range(interval.start, interval.end + 1)`
        e_in = (he >= a) & (he <= b)
        overlap = np.where(s_in & ~e_in, b - hs, np.where(e_in & ~s_in, he - a, np.where(s_in & e_in, he - hs, b - a)))
        ok = overlap >= mincols
        for k, i in enumerate(rows):
            lo, hi = int(off[k]), int(off[k + 1])
            matches[i] = (chrom, hits[lo:hi][ok[lo:hi]].tolist(), hi - lo)
    visited = {chrom: set() for chrom in right}
    for i, item in enumerate(left):
        if not isinstance(item, GenomicInterval):
            yield item
            continue
        chrom, good, found = matches.get(i, (item.chrom, [], 0))
        for j in good:
            visited[chrom].add(j)
            yield list(item) + list(right[chrom][2][j])
        if not good and rightfill:  # nothing found, or nothing met mincols
            yield list(item) + ["."] * rightlen
    if leftfill:
        for chrom, (rs, _, fields) in right.items():
            order = sorted(range(len(rs)), key=lambda j: (rs[j], -j))  # in-order of the reference's tree
            for j in order:
                if j not in visited[chrom]:
                    yield ["."] * leftlen + list(fields[j])


def base_coverage(reader):
    """operations/base_coverage.py:10-23.  Number of bases covered by the reader's intervals."""
    safe, bitsets = _first_safe_bitsets(reader, {})
    total = 0
    for bits in bitsets.values():
        err = _range_error(bits, 0, MAX_END)
        if err is not None:
            _note_reader_skip(safe, err)
            continue
        total += bits.count_range(0, MAX_END)
    return total
