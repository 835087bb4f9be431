"""
bx.intervals.operations.quicksect -- the multi-chromosome interval tree of
lib/bx/intervals/operations/quicksect.py:11-126, served by the MI355X index.

Same surface: ``IntervalTree().insert(interval, linenum, other)``,
``.intersect(interval, report_func)``, ``.traverse(func)`` and the per-chromosome
objects in ``.chroms`` with ``insert(start, end, linenum, other)`` (returns the
object to store back, as the reference's treap root does), ``intersect(start,
end, report_func)`` and ``traverse(func)``.  ``report_func`` / ``func`` receive
node views carrying ``start``, ``end``, ``linenum`` and ``other``.

What differs, and why it cannot matter to a caller: the reference reports the
hits of one query in the pre-order of a treap whose priorities come from
``random.uniform`` (quicksect.py:39-44,113-119), so the order changes from run
to run; here hits come in the index's order (by start).  ``traverse`` is the
treap's in-order in both: by start, and among equal starts the later insert
first (an equal key always descends left, quicksect.py:52-69).  The tree-shape
fields (``left``, ``right``, ``priority``, ``maxend``, ``minend``) do not exist.
Coordinates are C ints, as everywhere in the engine.

``intersect_batch`` is additive: all queries of one chromosome in one launch.
"""
import numpy as np

from bx.bitset import _cint
from bx.intervals.intersection import _Core
from bxmi._ffi import as_i32

__all__ = ["IntervalTree", "IntervalNode"]


class IntervalNode:
    """One stored interval as handed to report_func / traverse callbacks (quicksect.py:35-49)."""

    __slots__ = ("start", "end", "linenum", "other")

    def __init__(self, start, end, linenum=0, other=None):
        self.start = start
        self.end = end
        self.linenum = linenum
        self.other = other

    def __repr__(self):
        return "IntervalNode(%d, %d, linenum=%r)" % (self.start, self.end, self.linenum)


class _Chrom:
    """What ``tree.chroms[name]`` holds: the reference keeps the treap root there."""

    def __init__(self):
        self._core = _Core()

    def __len__(self):
        return len(self._core)

    def insert(self, start, end, linenum=0, other=None):
        # (the device index holds C ints: a coordinate beyond them raises here, as everywhere else in the overlay,
        # instead of wrapping around on its way through ctypes)
        start, end = _cint(start), _cint(end)
        self._core.insert(start, end, IntervalNode(start, end, linenum, other))
        return self

    def intersect(self, start, end, report_func):
        for node in self._core.find(_cint(start), _cint(end)):
            report_func(node)

    def traverse(self, func):
        core = self._core
        nodes, starts = core.values, core.starts
        for i in sorted(range(len(nodes)), key=lambda j: (starts[j], -j)):
            func(nodes[i])

    def intersect_batch(self, starts, ends):
        """-> (offsets int64[nq+1], list of node views): query k's hits are nodes[offsets[k]:offsets[k+1]]."""
        core = self._core
        core._flush()
        off, hits = core.index.find(as_i32(starts), as_i32(ends))
        vals = core.values
        return off, [vals[i] for i in hits.tolist()]


class IntervalTree:
    def __init__(self):
        self.chroms = {}

    def insert(self, interval, linenum=0, other=None):
        chrom = interval.chrom
        node = self.chroms.get(chrom)
        if node is None:
            node = _Chrom()
        self.chroms[chrom] = node.insert(interval.start, interval.end, linenum, other)

    def intersect(self, interval, report_func):
        node = self.chroms.get(interval.chrom)
        if node is not None:
            node.intersect(interval.start, interval.end, report_func)

    def traverse(self, func):
        for item in self.chroms.values():
            item.traverse(func)

    def intersect_batch(self, chrom, starts, ends):
        node = self.chroms.get(chrom)
        if node is None:
            return np.zeros(len(starts) + 1, dtype=np.int64), []
        return node.intersect_batch(starts, ends)
