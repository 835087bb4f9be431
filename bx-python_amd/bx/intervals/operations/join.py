"""bx.intervals.operations.join -- lib/bx/intervals/operations/join.py's entry point on the MI355X engine."""
from bxmi.operations import join  # noqa: F401
