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
from itertools import count

from bx.bitset import MAX

FIRST_LINE_IS_HEADER = object()


class ParseError(Exception):
    """tabular/io.py:9-18 -- the line number is appended to the text once known."""

    def __init__(self, *args, **kwargs):
        Exception.__init__(self, *args)
        self.linenum = kwargs.get("linenum", None)

    def _base(self):
        text = Exception.__str__(self)
        return text + " on line " + str(self.linenum) if self.linenum else text

    def __str__(self):
        return self._base()


class MissingFieldError(ParseError):
    pass


class FieldFormatError(ParseError):
    """intervals/io.py:20-29 -- '<text>[ on line N], integer expected'."""

    def __init__(self, *args, **kwargs):
        ParseError.__init__(self, *args, **kwargs)
        self.expected = kwargs.get("expected", None)

    def __str__(self):
        return self._base() + ", " + self.expected + " expected" if self.expected else self._base()


class StrandFormatError(ParseError):
    pass


class Header:
    """tabular/io.py:50-73"""

    def __init__(self, fields):
        self.set_fields(fields)

    def set_fields(self, fields):
        self.fields = fields
        self.field_to_column = dict(zip(fields, count()))

    def __getitem__(self, key):
        if isinstance(key, int):
            return self.fields[key]
        if isinstance(key, str):
            return key if key in self.field_to_column else None
        raise TypeError("field indices must be integers or strings")

    def __str__(self):
        return "#" + "\t".join(self.fields)


class Comment:
    """tabular/io.py:76-83"""

    def __init__(self, line):
        self.line = line

    def __str__(self):
        return self.line if self.line.startswith("#") else "#" + self.line


class TableRow:
    """tabular/io.py:21-47"""

    def __init__(self, reader, fields):
        self.reader = reader
        self.fields = fields

    def __getitem__(self, key):
        if isinstance(key, int):
            return self.fields[key]
        if isinstance(key, str):
            if self.reader.header:
                return self.fields[self.reader.header.field_to_column[key]]
            raise TypeError("column names only supported for files with headers")
        raise TypeError("field indices must be integers or strings")

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

    def __next__(self):
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


class NiceReaderWrapper(GenomicIntervalReader):
    """Skips unparsable lines and keeps count (intervals/io.py:219-265).  `current_line` is the raw line last read.
    `skip_log` (ours) keeps EVERY skip with the number of items delivered before it, so that a batched operation can
    replay its own skips into `skipped_lines` in file order."""

    def __init__(self, reader, **kwargs):
        self.outstream = kwargs.pop("outstream", None)
        self.print_delegate = kwargs.pop("print_delegate", None)
        GenomicIntervalReader.__init__(self, reader, **kwargs)
        self.input_wrapper = iter(self.input)
        self.input_iter = self._tracking_lines()
        self.skipped = 0
        self.skipped_lines = []
        self.skip_log = []
        self.delivered = 0

    def _tracking_lines(self):
        for self.current_line in self.input_wrapper:
            yield self.current_line

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

    def __next__(self):
        while True:
            item = self._next_parsed()
            if isinstance(item, GenomicInterval) and item.end > self.lens.get(item.chrom, MAX):
                self.note_skip(self.linenum, self.current_line, "Error in BitsetSafeReaderWrapper")
                continue
            self.delivered += 1
            return item
