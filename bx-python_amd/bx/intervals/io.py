"""bx.intervals.io -- the reader classes of lib/bx/intervals/io.py:16-300 (bxmi.genomic holds them)."""
from bxmi.genomic import (  # noqa: F401
    BitsetSafeReaderWrapper,
    Comment,
    FieldFormatError,
    GenomicInterval,
    GenomicIntervalReader,
    Header,
    MissingFieldError,
    NiceReaderWrapper,
    ParseError,
    StrandFormatError,
    TableReader,
    TableRow,
)
