"""bx.intervals.operations.intersect -- lib/bx/intervals/operations/intersect.py's entry point on the MI355X engine."""
from bxmi.operations import intersect  # noqa: F401
