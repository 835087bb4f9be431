"""bx.intervals.operations.subtract -- lib/bx/intervals/operations/subtract.py's entry point on the MI355X engine."""
from bxmi.operations import subtract  # noqa: F401
