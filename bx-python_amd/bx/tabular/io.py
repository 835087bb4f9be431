"""bx.tabular.io -- the row/reader base classes of lib/bx/tabular/io.py:10-150 (bxmi.genomic holds them)."""
from bxmi.genomic import Comment, Header, ParseError, TableReader, TableRow  # noqa: F401
