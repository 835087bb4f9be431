"""bx.intervals.operations.find_clusters -- lib/bx/intervals/operations/find_clusters.py's entry point on the MI355X engine."""
from bxmi.operations import find_clusters  # noqa: F401
