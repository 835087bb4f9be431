"""bx.intervals.operations.base_coverage -- lib/bx/intervals/operations/base_coverage.py's entry point on the MI355X engine."""
from bxmi.operations import base_coverage  # noqa: F401
