"""bx.intervals.operations.coverage -- lib/bx/intervals/operations/coverage.py's entry point on the MI355X engine."""
from bxmi.operations import coverage  # noqa: F401
