"""
bx.intervals.operations -- same names as lib/bx/intervals/operations/__init__.py:6-33;
the operations themselves live in bxmi.operations (one batched engine call per chromosome).
"""
from bxmi.operations import (  # noqa: F401
    BED_DEFAULT_COLS,
    MAX_END,
    bits_clear_in_range,
    bits_set_in_range,
)
