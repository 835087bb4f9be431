"""
bx.intervals.operations -- same names as lib/bx/intervals/operations/__init__.py:6-33;
the operations themselves live in bxmi.operations (one batched engine call per chromosome).
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)  # (every operation of the reference's package is carried here, concat included since round 5)

from bxmi.operations import (  # noqa: E402,F401
    BED_DEFAULT_COLS,
    MAX_END,
    bits_clear_in_range,
    bits_set_in_range,
)
