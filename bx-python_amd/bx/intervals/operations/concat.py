"""
bx.intervals.operations.concat -- the reference's lib/bx/intervals/operations/concat.py:20-61: the rows of several interval
readers one after the other, in the FIRST reader's format.  No engine work (rows are copied and reshaped on the host).

`sameformat=True` (or while still inside the first input that produced a row): a row longer than the first row's field count is
cut to it (the reference pads only inside that same branch, i.e. never -- a shorter row stays shorter); otherwise a row is
rebuilt as "." fields with chrom / start / end / strand written into the FIRST reader's columns.  Headers and comments pass
through when asked for.
"""
from bx.intervals.io import GenomicInterval
from bx.tabular.io import Comment, Header


def _reshaped(row, nfields, cols):
    """`row` as the first reader's layout: dots everywhere, the four coordinates in that reader's columns."""
    chrom_col, start_col, end_col, strand_col = cols
    chrom, start, end, strand = row.chrom, row.start, row.end, row.strand
    fields = ["."] * nfields
    fields[chrom_col] = chrom
    fields[start_col] = str(start)
    fields[end_col] = str(end)
    if strand_col < len(fields):  # (strand is optional: the first format may have no such column)
        fields[strand_col] = strand
    row.fields = fields
    return row


def concat(readers, comments=True, header=True, sameformat=True):
    first = readers[0]
    cols = (first.chrom_col, first.start_col, first.end_col, first.strand_col)
    nfields = None
    in_first = True      # no row has been produced by an input that is now finished
    produced = False
    for reader in readers:
        for item in reader:
            if isinstance(item, GenomicInterval):
                if not nfields:
                    nfields = item.nfields
                row = item.copy()
                if sameformat or in_first:
                    if len(row.fields) > nfields:
                        row.fields = row.fields[:nfields]
                    produced = True
                    yield row
                else:
                    yield _reshaped(row, nfields, cols)
            elif isinstance(item, Header):
                if header:
                    yield item
            elif isinstance(item, Comment):
                if comments:
                    yield item
        if produced and in_first:
            in_first = False
