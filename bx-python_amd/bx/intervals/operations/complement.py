"""bx.intervals.operations.complement -- lib/bx/intervals/operations/complement.py's entry point on the MI355X engine."""
from bxmi.operations import complement  # noqa: F401
