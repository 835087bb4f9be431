"""
Tools and data structures for working with genomic intervals; mirrors
lib/bx/intervals/__init__.py:7-14 of the reference (same re-exports).
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from bx.intervals.intersection import (  # noqa: E402
    Intersecter,
    Interval,
    IntervalNode,
    IntervalTree,
)

__all__ = ["Intersecter", "Interval", "IntervalNode", "IntervalTree"]
