"""
Minimal tab-separated interval rows for the interval_join counterpart; mirrors
what lib/bx/intervals/io.py:36-103 (GenomicInterval) and lib/bx/tabular/io.py:86-156
(TableReader) do to a BED line, including which inputs make the reference fail.
"""


class ParseError(Exception):
    def __init__(self, *args, **kwargs):
        Exception.__init__(self, *args)
        self.linenum = kwargs.get("linenum", None)

    def __str__(self):
        s = Exception.__str__(self)
        return s + " on line " + str(self.linenum) if self.linenum else s


class MissingFieldError(ParseError):
    pass


class FieldFormatError(ParseError):
    def __init__(self, *args, **kwargs):
        ParseError.__init__(self, *args, **kwargs)
        self.expected = kwargs.get("expected", None)

    def __str__(self):
        s = ParseError.__str__(self)
        return s + ", " + self.expected + " expected" if self.expected else s


class StrandFormatError(ParseError):
    pass


class Row:
    __slots__ = ("fields", "chrom", "start", "end", "strand")

    def __str__(self):
        return "\t".join(self.fields)


def read_rows(lines, chrom_col=0, start_col=1, end_col=2, strand_col=5):
    """Yield Row objects; raise where interval_join.py would."""
    for linenum, line in enumerate(lines, 1):
        line = line.rstrip("\r\n")
        if line == "":
            raise AttributeError("'Comment' object has no attribute 'chrom'")  # tabular/io.py:108-110
        if line.startswith("#") or line.startswith("track "):
            kind = "Header" if linenum == 1 else "Comment"  # tabular/io.py:118-130
            raise AttributeError("'%s' object has no attribute 'chrom'" % kind)
        fields = line.split("\t")
        n = len(fields)
        try:
            if chrom_col >= n:
                raise MissingFieldError("No field for chrom_col (%d)" % chrom_col)
            r = Row()
            r.fields = fields
            r.chrom = fields[chrom_col].strip()
            if start_col >= n:
                raise MissingFieldError("No field for start_col (%d)" % start_col)
            try:
                r.start = int(fields[start_col])
            except ValueError as e:
                raise FieldFormatError("Could not parse start_col: " + str(e), expected="integer")
            if end_col >= n:
                raise MissingFieldError("No field for end_col (%d)" % end_col)
            try:
                r.end = int(fields[end_col])
            except ValueError as e:
                raise FieldFormatError("Could not parse end_col: " + str(e), expected="integer")
            if r.end < r.start:
                raise ParseError("Start is greater than End. Interval length is < 1.")
            if strand_col >= n or strand_col < 0:
                r.strand = "+"
            else:
                strand = fields[strand_col]
                if strand == ".":
                    strand = "+"
                elif strand not in ("+", "-"):
                    raise StrandFormatError("Strand must be either '+' or '-'")
                r.strand = strand
        except ParseError as e:
            e.linenum = linenum
            raise
        yield r
