"""
Delimited-text interval readers for the operations layer.

Counterparts of the reference's reader classes -- same names, iteration protocol, exception
types, message texts and skip bookkeeping (lib/bx/tabular/io.py:9-156, lib/bx/intervals/io.py:16-294),
because `bxmi.operations` must reproduce what the reference's operations observe and record
through them.  Host-side text handling only; nothing here touches the GPU.

A reader yields, in file order, `Header` (first line if it is a comment line), `Comment`
(other comment and blank lines) and `GenomicInterval` rows.  `NiceReaderWrapper` swallows
`ParseError`s and logs them in `skipped` / `skipped_lines`; `BitsetSafeReaderWrapper`
additionally drops rows that end beyond the chromosome length.
"""
from bx.bitset import MAX

FIRST_LINE_IS_HEADER = object()


class ParseError(Exception):
    """A line the reader cannot turn into a row.  The reader adds the line number afterwards (`linenum`, also accepted
    as a keyword); the text then ends in " on line N" (what tabular/io.py:9-18 prints)."""

    def __init__(self, *args, linenum=None, **_other_keywords):
        super().__init__(*args)
        self.linenum = linenum

    def _base(self):
        parts = [super().__str__()]
        if self.linenum:
            parts.append("on line %s" % (self.linenum,))
        return " ".join(parts)

    def __str__(self):
        return self._base()


class MissingFieldError(ParseError):
    pass


class FieldFormatError(ParseError):
    """'<text>[ on line N], integer expected' (intervals/io.py:20-29): `expected` names what the field should have held."""

    def __init__(self, *args, expected=None, **keywords):
        super().__init__(*args, **keywords)
        self.expected = expected

    def __str__(self):
        text = self._base()
        return "%s, %s expected" % (text, self.expected) if self.expected else text


class StrandFormatError(ParseError):
    pass


def _field(fields, key, by_name):
    """What `obj[key]` means for the row-like objects of a table: a column number indexes `fields`, a column name goes
    through `by_name`, anything else is a TypeError with the reference's text (tabular/io.py:30-40, :64-70)."""
    if isinstance(key, int):
        return fields[key]
    if isinstance(key, str):
        return by_name(key)
    raise TypeError("field indices must be integers or strings")


class Header:
    """The first line of a table when it is a comment line: column names, and `field_to_column` from a name to its
    (last) column (tabular/io.py:50-73).  `header["name"]` answers the name itself when the table has that column, else None."""

    def __init__(self, fields):
        self.set_fields(fields)

    def set_fields(self, fields):
        self.fields = fields
        self.field_to_column = {name: column for column, name in enumerate(fields)}

    def __getitem__(self, key):
        return _field(self.fields, key, lambda name: name if name in self.field_to_column else None)

    def __str__(self):
        return "#" + "\t".join(self.fields)


class Comment:
    """A comment or blank line, kept as it stood; printed with a leading '#' whether it had one or not (tabular/io.py:76-83)."""

    def __init__(self, line):
        self.line = line

    def __str__(self):
        return self.line if self.line[:1] == "#" else "#" + self.line


class TableRow:
    """One data line split into `fields`; columns by number, or by name when the reader saw a header (tabular/io.py:21-47)."""

    def __init__(self, reader, fields):
        self.reader = reader
        self.fields = fields

    def _named(self, name):
        header = self.reader.header
        if not header:
            raise TypeError("column names only supported for files with headers")
        return self.fields[header.field_to_column[name]]

    def __getitem__(self, key):
        return _field(self.fields, key, self._named)

    @property
    def fieldnames(self):
        return self.reader.header.fields

    def __str__(self):
        return "\t".join(self.fields)


_SYNCED = {"chrom": "chrom_col", "start": "start_col", "end": "end_col"}


class GenomicInterval(TableRow):
    """A row with chrom / start / end / strand views of its fields (intervals/io.py:36-104).
    Assigning one of the four attributes rewrites the field it came from."""

    def __init__(self, reader, fields, chrom_col, start_col, end_col, strand_col, default_strand, fix_strand=False):
        # Written against __dict__ directly: this constructor runs once per line of every file, and routing its ten
        # assignments through __setattr__ was half of the readers' time.  The effect is what assignment through
        # __setattr__ gives in the reference: each parsed value is written back, NORMALISED, into the field it came
        # from (" chr1 " -> "chr1", "+5" -> "5", strand "." -> the default); str(row) shows it.
        d = self.__dict__
        d["reader"], d["fields"] = reader, fields
        d["chrom_col"], d["start_col"], d["end_col"], d["strand_col"] = chrom_col, start_col, end_col, strand_col
        d["nfields"] = nfields = len(fields)
        if chrom_col >= nfields:
            raise MissingFieldError("No field for chrom_col (%d)" % chrom_col)
        d["chrom"] = fields[chrom_col] = fields[chrom_col].strip()
        if start_col >= nfields:
            raise MissingFieldError("No field for start_col (%d)" % start_col)
        try:
            start = int(fields[start_col])
        except ValueError as e:
            raise FieldFormatError("Could not parse start_col: " + str(e), expected="integer")
        d["start"], fields[start_col] = start, str(start)
        if end_col >= nfields:
            raise MissingFieldError("No field for end_col (%d)" % end_col)
        try:
            end = int(fields[end_col])
        except ValueError as e:
            raise FieldFormatError("Could not parse end_col: " + str(e), expected="integer")
        d["end"], fields[end_col] = end, str(end)
        if end < start:
            raise ParseError("Start is greater than End. Interval length is < 1.")
        strand = default_strand
        if 0 <= strand_col < nfields:
            strand = fields[strand_col]
            if strand == ".":
                strand = default_strand
            elif strand not in ("+", "-"):
                if not fix_strand:
                    raise StrandFormatError("Strand must be either '+' or '-'")
                strand = "+"
            fields[strand_col] = str(strand)
        d["strand"] = strand

    @classmethod
    def _from_parsed(cls, reader, line, chrom, start, end, strand):
        """A row the native parser (csrc/bedparse.cpp, table mode) has already accepted: its fields are in the normal
        form the constructor would write back, so nothing is parsed or rewritten here and the TAB split itself waits
        until somebody looks at `fields` (a row that is only compared and dropped never pays for it)."""
        new = object.__new__(cls)
        d = new.__dict__
        d["reader"], d["_line"] = reader, line
        d["chrom_col"], d["start_col"], d["end_col"], d["strand_col"] = reader.chrom_col, reader.start_col, reader.end_col, reader.strand_col
        d["chrom"], d["start"], d["end"], d["strand"] = chrom, start, end, strand
        return new

    def __str__(self):
        d = self.__dict__
        if "fields" not in d and "_line" in d:
            return d["_line"]  # never split, hence never changed: the line is its own normal form (_from_parsed)
        return "\t".join(self.fields)

    @classmethod
    def _from_values(cls, reader, fields, chrom, start, end, strand):
        """A NEW row made by an operation from values that are in normal form already (a chromosome name taken from a
        reader's rows, integers with start <= end, strand "+" or "-"): what the constructor would leave behind for
        `fields` holding those values, without parsing them back (complement.py:44-49 builds its rows this way)."""
        new = object.__new__(cls)
        d = new.__dict__
        d["reader"], d["fields"] = reader, fields
        cc, sc, ec, tc = reader.chrom_col, reader.start_col, reader.end_col, reader.strand_col
        d["chrom_col"], d["start_col"], d["end_col"], d["strand_col"] = cc, sc, ec, tc
        d["nfields"] = len(fields)
        d["chrom"] = fields[cc] = chrom
        d["start"], fields[sc] = start, str(start)
        d["end"], fields[ec] = end, str(end)
        if 0 <= tc < len(fields):
            fields[tc] = strand
        d["strand"] = strand
        return new

    def _append_fields(self, *more):
        """`self.fields.append(x)` for every x (what operations/coverage.py:71-72 does to a row) -- on a row that has not
        been split yet the text grows instead, which is the same thing seen through `fields`; `nfields` keeps the value
        the constructor gave it in the reference (the number of fields the row came with)."""
        d = self.__dict__
        if "fields" in d:
            d["fields"].extend(more)
            return
        line = d["_line"]
        if "nfields" not in d:
            d["nfields"] = line.count("\t") + 1
        d["_line"] = line + "\t" + "\t".join(more)

    def __getattr__(self, name):  # only reached for attributes that are not there yet
        if name == "fields":
            fields = self.__dict__["fields"] = self.__dict__["_line"].split("\t")
            return fields
        if name == "nfields":
            n = self.__dict__["nfields"] = len(self.fields)
            return n
        raise AttributeError(name)

    def __setattr__(self, name, value):
        col_attr = _SYNCED.get(name)
        if col_attr is not None:
            self.fields[getattr(self, col_attr)] = str(value)
        elif name == "strand" and 0 <= self.strand_col < self.nfields:
            self.fields[self.strand_col] = str(value)
        object.__setattr__(self, name, value)

    def copy(self):
        return GenomicInterval(self.reader, list(self.fields), self.chrom_col, self.start_col, self.end_col, self.strand_col,
                               self.strand)

    def _piece(self, start, end):
        """A copy with new start / end, for rows that come straight from a reader (fields and attributes agree):
        what copy() followed by the two assignments produces, without parsing the fields again."""
        new = object.__new__(GenomicInterval)
        d = new.__dict__
        d.update(self.__dict__)
        fields = d["fields"] = list(self.fields)
        d["start"], d["end"] = start, end
        fields[self.start_col], fields[self.end_col] = str(start), str(end)
        return new


class TableReader:
    """tabular/io.py:86-156: line classification (blank -> Comment(''), first comment line -> Header, ...)."""

    def __init__(self, input, return_header=True, return_comments=True, force_header=None, comment_lines_startswith=("#",)):
        self.input = input
        self.return_comments = return_comments
        self.return_header = return_header
        self.input_iter = iter(input)
        self.linenum = 0
        self.header = force_header
        self.comment_lines_startswith = comment_lines_startswith

    def __iter__(self):
        return self

    # -- native bulk parse (GenomicIntervalReader only; see _bulk_open) --
    _bulk = None       # parsed prefix of the input, or None
    _bulk_i = 0        # lines of it already delivered
    _bulk_tried = False

    def _bulk_open(self):
        """Hook: subclasses that know how to pre-parse their input natively return a bxmi.tabio.ParsedTable."""
        return None

    def _bulk_rest(self):
        """Switch from the parsed prefix to the per-line code for whatever follows it."""
        self.input_iter = iter(self._bulk.rest_lines(self.input))
        self._bulk = None

    def _bulk_item(self, i):
        """Line i of the parsed prefix as the object the per-line code would have produced (None = not delivered)."""
        b = self._bulk
        kind = b.kind[i]
        if kind == 0:
            return self._bulk_row(b, i)
        if kind == 1:
            return Comment("") if self.return_comments else None
        line = b.line(i)
        if kind == 3 and self.header is None:
            self.header = self.parse_header(line)
            return self.header if self.return_header else None
        return self.parse_comment(line) if self.return_comments else None

    def __next__(self):
        if not self._bulk_tried:
            self._bulk_tried = True
            self._bulk = self._bulk_open() if self.linenum == 0 else None
        while self._bulk is not None:
            i = self._bulk_i
            if i >= self._bulk.n:
                self._bulk_rest()
                break
            self._bulk_i = i + 1
            self.linenum += 1
            item = self._bulk_item(i)
            if item is not None:
                return item
        while True:
            line = next(self.input_iter)
            self.linenum += 1
            line = line.rstrip("\r\n")
            if line == "":
                if self.return_comments:
                    return Comment(line)
                continue
            first = self.linenum == 1
            if first and self.header is FIRST_LINE_IS_HEADER:
                self.header = self.parse_header(line)
                if self.return_header:
                    return self.header
                continue
            if line.startswith(tuple(self.comment_lines_startswith)):
                if first and self.header is None:
                    self.header = self.parse_header(line)
                    if self.return_header:
                        return self.header
                elif self.return_comments:
                    return self.parse_comment(line)
                continue
            try:
                return self.parse_row(line)
            except ParseError as e:
                e.linenum = self.linenum
                raise

    def parse_header(self, line):
        return Header((line[1:] if line.startswith("#") else line).split("\t"))

    def parse_comment(self, line):
        return Comment(line)

    def parse_row(self, line):
        return TableRow(self, line.split("\t"))


class GenomicIntervalReader(TableReader):
    """intervals/io.py:107-216"""

    def __init__(self, input, chrom_col=0, start_col=1, end_col=2, strand_col=5, default_strand="+", return_header=True,
                 return_comments=True, force_header=None, fix_strand=False, comment_lines_startswith=None, allow_spaces=False):
        if comment_lines_startswith is None:
            comment_lines_startswith = ["#", "track "]
        TableReader.__init__(self, input, return_header, return_comments, force_header, comment_lines_startswith)
        self.chrom_col = chrom_col
        self.start_col = start_col
        self.end_col = end_col
        self.strand_col = strand_col
        self.default_strand = default_strand
        self.fix_strand = fix_strand
        self.allow_spaces = allow_spaces

    def _bulk_open(self):
        # only for the classes of this module as they are (a subclass with its own row or header handling keeps the per-line
        # path), with the default header rule, on input whose text can be taken as a whole
        from . import tabio

        cls = type(self)
        if (cls.parse_row is not GenomicIntervalReader.parse_row or cls.parse_header is not TableReader.parse_header
                or cls.parse_comment is not TableReader.parse_comment or self.header is not None or not tabio.enabled()):
            return None
        return tabio.parse_input(self.input, self.chrom_col, self.start_col, self.end_col, self.strand_col, self.comment_lines_startswith)

    def _bulk_row(self, b, i):
        strand = b.strand[i]
        return GenomicInterval._from_parsed(self, b.line(i), b.names[b.chrom[i]], b.start[i], b.end[i],
                                            self.default_strand if strand == 0 else ("+" if strand == 43 else "-"))

    def parse_row(self, line):
        # tab first; any whitespace as a second try when allowed -- the FIRST error is the one reported
        first_error = None
        for sep in ("\t", None) if self.allow_spaces else ("\t",):
            try:
                return GenomicInterval(self, line.split(sep), self.chrom_col, self.start_col, self.end_col, self.strand_col,
                                       self.default_strand, fix_strand=self.fix_strand)
            except Exception as e:
                if first_error is None:
                    first_error = e
        raise first_error

    def binned_bitsets(self, upstream_pad=0, downstream_pad=0, lens=None):
        """One BinnedBitSet per chromosome, in first-appearance order (intervals/io.py:189-216; the pads are accepted
        and ignored there too).  Keys are the RAW chromosome fields; starts are clamped at 0, ends at the set's size."""
        from bx.bitset import BinnedBitSet

        lens = {} if lens is None else lens
        bitsets = {}
        last_chrom, last = None, None
        self._bulk_bitsets(bitsets, lens, BinnedBitSet)  # the natively parsed prefix in one go; the loop below takes the rest
        for row in self:
            if not isinstance(row, GenomicInterval):
                continue
            chrom = row[self.chrom_col]
            if chrom != last_chrom:
                if chrom not in bitsets:
                    size = lens.get(chrom, MAX)
                    try:
                        bitsets[chrom] = BinnedBitSet(size)
                    except ValueError as e:
                        raise Exception("Invalid chrom length %s in 'lens' dictionary. %s" % (str(size), str(e)))
                last_chrom, last = chrom, bitsets[chrom]
            start = max(int(row[self.start_col]), 0)
            end = min(int(row[self.end_col]), last.size)
            last.set_range(start, end - start)  # queued by the drop-in class: one launch per chromosome
        return bitsets


def _bulk_bitsets(self, bitsets, lens, BinnedBitSet):
    """binned_bitsets() over the natively parsed prefix of the input without making a row object per line: what the loop
    in binned_bitsets does row by row -- new set at a chromosome's first row, start clamped at 0, end at the set's size,
    set_range -- done per chromosome with one batch call, stopping where that loop would have raised (and raising the
    same thing), and leaving the reader as if it had iterated those lines (line number, header, and for the wrappers
    the delivered / skipped bookkeeping, in file order)."""
    import numpy as np

    if self._bulk_tried or self.linenum != 0:
        return
    self._bulk_tried = True
    b = self._bulk = self._bulk_open()
    if b is None:
        return
    kind = b.kind_a
    rows = np.nonzero(kind == 0)[0]
    chrom, st, en = b.chrom_a[rows], b.start_a[rows], b.end_a[rows]
    keep = self._bulk_keep(b, rows, chrom, en)  # None = every row is delivered
    kr = rows if keep is None else rows[keep]
    kc, ks, ke = (chrom, st, en) if keep is None else (chrom[keep], st[keep], en[keep])
    stop_line = b.n  # first line the row loop would NOT have got past
    problem = None
    if len(kr):
        nn = len(b.names)
        first = np.full(nn, len(kr), dtype=np.int64)
        np.minimum.at(first, kc, np.arange(len(kr)))
        order = [c for c in np.argsort(first, kind="stable").tolist() if first[c] < len(kr)]
        sizes = np.zeros(nn, dtype=np.int64)
        cut = len(kr)  # position (among the kept rows) of the first row that raises
        made = {}
        for c in order:
            name = b.names[c]
            try:
                size = lens.get(name, MAX)
                made[c] = BinnedBitSet(size)
                sizes[c] = made[c].size
            except ValueError as e:
                cut = int(first[c])
                problem = Exception("Invalid chrom length %s in 'lens' dictionary. %s" % (str(size), str(e)))
                break
        known = np.zeros(nn, dtype=bool)
        known[list(made)] = True
        sz = sizes[kc]
        cs = np.maximum(ks, 0)
        ce = np.minimum(ke, sz)
        cnt = ce - cs
        bad = known[kc] & ((cs >= sz) | (cnt < 0))
        pos = np.arange(len(kr))
        if bad[:cut].any():
            cut = int(np.argmax(bad[:cut]))
            problem = ("range", cut)
        for c in order:
            if c not in made or first[c] > cut or (first[c] == cut and not isinstance(problem, tuple)):
                continue
            bitsets[b.names[c]] = made[c]
            sel = (kc == c) & (pos < cut) & (cnt > 0)
            if sel.any():
                made[c].set_ranges(cs[sel].astype(np.int32), cnt[sel].astype(np.int32))
        if problem is not None:
            stop_line = int(kr[cut]) + 1
    self._bulk_account(b, keep, rows, stop_line)
    if problem is not None:
        if isinstance(problem, tuple):
            c = int(kc[cut])
            bitsets[b.names[c]].set_range(int(cs[cut]), int(cnt[cut]))  # raises what the row loop would have raised
            raise AssertionError("unreachable: the range was found invalid")
        raise problem


def _bulk_account(self, b, keep, rows, stop_line):
    """Reader state after lines [0, stop_line) of the parsed prefix have gone by (plain reader: line number and header)."""
    if b.n and b.kind[0] == 3 and self.header is None:
        self.header = self.parse_header(b.line(0))
    self.linenum = stop_line
    self._bulk_i = stop_line


def _bulk_take(self):
    """Hand the natively parsed prefix of a fresh reader to a caller that works on its arrays (bxmi.operations): the reader
    is left as if it had iterated those lines and delivered every one of them.  None if there is nothing to hand over (not
    fresh, input not parseable that way, or a reader that holds some kinds of lines back)."""
    if self._bulk_tried or self.linenum != 0 or not (self.return_header and self.return_comments):
        return None
    if type(self)._bulk_keep is not GenomicIntervalReader._bulk_keep:
        return None  # (a wrapper that drops rows: its items are not one per line)
    self._bulk_tried = True
    b = self._bulk = self._bulk_open()
    if b is None:
        return None
    self._bulk_account(b, None, None, b.n)
    return b


GenomicIntervalReader._bulk_take = _bulk_take
GenomicIntervalReader._bulk_bitsets = _bulk_bitsets
GenomicIntervalReader._bulk_account = _bulk_account
GenomicIntervalReader._bulk_keep = lambda self, b, rows, chrom, end: None


class NiceReaderWrapper(GenomicIntervalReader):
    """Skips unparsable lines and keeps count (intervals/io.py:219-265).  `current_line` is the raw line last read.
    `skip_log` (ours) keeps EVERY skip with the number of items delivered before it, so that a batched operation can
    replay its own skips into `skipped_lines` in file order."""

    def __init__(self, reader, **kwargs):
        self.outstream = kwargs.pop("outstream", None)
        self.print_delegate = kwargs.pop("print_delegate", None)
        GenomicIntervalReader.__init__(self, reader, **kwargs)
        self._current_line = None
        self.input_wrapper = None  # iter(self.input), made when the per-line code first needs it (a parsed prefix may come first)
        self.input_iter = self._tracking_lines()
        self.skipped = 0
        self.skipped_lines = []
        self.skip_log = []
        self.delivered = 0

    @property
    def current_line(self):
        """The raw line last read (with its line end), whichever path read it."""
        b = self._bulk
        if b is not None and self._bulk_i > 0:
            return b.raw_line(self._bulk_i - 1)
        return self._current_line

    @current_line.setter
    def current_line(self, value):
        self._current_line = value

    def _bulk_rest(self):
        self._current_line = self.current_line
        self.input_wrapper = iter(self._bulk.rest_lines(self.input))
        self._bulk = None

    def _bulk_account(self, b, keep, rows, stop_line):
        import numpy as np

        GenomicIntervalReader._bulk_account(self, b, keep, rows, stop_line)
        kind = b.kind_a[:stop_line]
        delivered = np.where(kind == 0, True, np.where(kind == 3, self.return_header, self.return_comments))
        if keep is not None:
            dropped = rows[~keep]
            dropped = dropped[dropped < stop_line]
            delivered[dropped] = False
            before = np.concatenate([[0], np.cumsum(delivered)])
            base = self.delivered
            for i in dropped.tolist():  # in file order, each with the count of items delivered before it
                self.delivered = base + int(before[i])
                self.note_skip(i + 1, b.raw_line(i), "Error in BitsetSafeReaderWrapper")
            self.delivered = base
        self.delivered += int(delivered.sum())

    def _tracking_lines(self):
        if self.input_wrapper is None:
            self.input_wrapper = iter(self.input)
        for self._current_line in self.input_wrapper:
            yield self._current_line

    def note_skip(self, linenum, line, message):
        self.skipped += 1
        self.skip_log.append((self.delivered, (linenum, line, message)))
        if self.skipped < 10:  # "no reason to stuff an entire bad file into memory"
            self.skipped_lines.append((linenum, line, message))

    def _next_parsed(self):
        while True:
            try:
                return GenomicIntervalReader.__next__(self)
            except ParseError as e:
                if self.outstream and self.print_delegate and callable(self.print_delegate):
                    self.print_delegate(self.outstream, e, self)
                self.note_skip(self.linenum, self.current_line, str(e))

    def __next__(self):
        item = self._next_parsed()
        self.delivered += 1
        return item


class BitsetSafeReaderWrapper(NiceReaderWrapper):
    """Re-reads `reader.input` with the reader's columns and drops rows ending beyond the chromosome
    (intervals/io.py:268-294)."""

    def __init__(self, reader, lens=None):
        NiceReaderWrapper.__init__(self, reader.input, chrom_col=reader.chrom_col, start_col=reader.start_col,
                                   end_col=reader.end_col, strand_col=reader.strand_col)
        self.lens = {} if lens is None else lens

    def _bulk_keep(self, b, rows, chrom, end):
        import numpy as np

        limit = np.array([self.lens.get(name, MAX) for name in b.names] or [MAX], dtype=np.int64)
        return end <= limit[chrom] if len(rows) else np.ones(0, dtype=bool)

    def __next__(self):
        while True:
            item = self._next_parsed()
            if isinstance(item, GenomicInterval) and item.end > self.lens.get(item.chrom, MAX):
                self.note_skip(self.linenum, self.current_line, "Error in BitsetSafeReaderWrapper")
                continue
            self.delivered += 1
            return item
