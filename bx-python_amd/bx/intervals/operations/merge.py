"""bx.intervals.operations.merge -- lib/bx/intervals/operations/merge.py's entry point on the MI355X engine."""
from bxmi.operations import merge  # noqa: F401
