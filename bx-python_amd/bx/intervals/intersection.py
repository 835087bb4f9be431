"""
bx.intervals.intersection -- drop-in for the reference's Cython module
lib/bx/intervals/intersection.pyx, served by the MI355X engine
(bxmi.intervals.IntervalIndex over libbxmi.so).

Same public names and behaviour as the reference:
  Interval                     intersection.pyx:274-323
  IntervalNode                 intersection.pyx:61-268
  IntervalTree / Intersecter   intersection.pyx:325-488

The tree is not a treap here.  ``insert`` queues (start, end, payload); the
first query uploads the queue and builds the device index (radix sort into the
treap's in-order + 32-ary search levels); ``find`` is a batch-of-one of the
batched kernel and returns the stored Python objects in the reference's order.
Throughput comes from the additive batch methods ``find_batch`` / ``count_batch``.
Nothing is computed on the CPU: without libbxmi.so and a GPU these classes raise.
"""
import operator

import numpy as np

from bx.bitset import _cint
from bxmi._ffi import as_i32 as _ffi_as_i32
from bxmi.intervals import IntervalIndex

__all__ = ["Interval", "IntervalNode", "IntervalTree", "Intersecter"]


class Interval:
    """
    A stored feature: integer ``start``/``end`` plus free-form ``value``, ``chrom`` and ``strand``
    (``-1`` or ``"-"`` flips the up/downstream queries).  Same constructor, repr and comparison
    rules as the reference class (intersection.pyx:274-323).

    >>> f1 = Interval(23, 36)
    >>> f2 = Interval(34, 48, value={'chr': 12, 'anno': 'transposon'})
    >>> f2
    Interval(34, 48, value={'chr': 12, 'anno': 'transposon'})
    """

    __slots__ = ("_start", "_end", "value", "chrom", "strand")

    # `cdef public int start, end` (intersection.pyx:276): every assignment is coerced to a C int, later ones too
    start = property(lambda self: self._start, lambda self, v: setattr(self, "_start", _cint(v)))
    end = property(lambda self: self._end, lambda self, v: setattr(self, "_end", _cint(v)))

    def __init__(self, start, end, value=None, chrom=None, strand=None):
        start, end = _cint(start), _cint(end)
        assert start <= end, "start must be less than end"
        self._start = start
        self._end = end
        self.value = value
        self.chrom = chrom
        self.strand = strand

    def __repr__(self):
        fstr = "Interval(%d, %d" % (self.start, self.end)
        if self.value is not None:
            fstr += ", value=" + str(self.value)
        fstr += ")"
        return fstr

    # intersection.pyx:305-323 -- the reference's (deliberately odd) ordering
    def __lt__(self, other):
        return self.start < other.start or self.end < other.end

    def __le__(self, other):
        return self == other or self < other

    def __eq__(self, other):
        return self.start == other.start and self.end == other.end

    def __ne__(self, other):
        return self.start != other.start or self.end != other.end

    def __gt__(self, other):
        return self.start > other.start or self.end > other.end

    def __ge__(self, other):
        return self == other or self > other

    __hash__ = None


class _Core:
    """Payload list + device index shared by an IntervalTree and its IntervalNode views."""

    def __init__(self):
        self.index = IntervalIndex()
        self.values = []
        self.starts = []
        self.ends = []
        self._flushed = 0
        self._order = None

    def insert(self, start, end, value):
        self.starts.append(start)
        self.ends.append(end)
        self.values.append(value)
        self._order = None

    def __len__(self):
        return len(self.values)

    def _flush(self):
        n = len(self.values)
        if self._flushed < n:
            self.index.append(
                np.array(self.starts[self._flushed:], dtype=np.int32), np.array(self.ends[self._flushed:], dtype=np.int32)
            )
            self._flushed = n

    def find(self, start, end):
        self._flush()
        vals = self.values
        return [vals[i] for i in self.index.find_one_list(start, end)]

    def order(self):
        if self._order is None:
            self._flush()
            self._order = self.index.order().tolist()
        return self._order

    # intersection.pyx:232-260
    def left(self, position, n=1, max_dist=2500):
        n, max_dist = _cint(n), _cint(max_dist)
        self._flush()
        cand = self.index.neighbors(_cint(position - 1) + 1, max_dist, -1).tolist()
        results = [self.values[i] for i in cand]
        if len(results) == n:
            return results
        results.sort(key=operator.attrgetter("end"), reverse=True)
        return results[:n]

    def right(self, position, n=1, max_dist=2500):
        n, max_dist = _cint(n), _cint(max_dist)
        self._flush()
        cand = self.index.neighbors(_cint(position + 1) - 1, max_dist, +1).tolist()
        results = [self.values[i] for i in cand]
        if len(results) == n:
            return results
        results.sort(key=operator.attrgetter("start"))
        return results[:n]


class IntervalNode:
    """
    Node-level API of the reference (intersection.pyx:61-268), kept for code and tests that drive
    the tree through its root node.  Here a node is a *view* of a position range of the device
    index laid out as an implicit balanced search tree; the root view spans everything.
    """

    def __init__(self, start, end, interval, _core=None, _span=None, _parent=None):
        if _core is None:
            _core = _Core()
            _core.insert(_cint(start), _cint(end), interval)
        self._core = _core
        self._span = _span  # None = whole tree (root); else (lo, hi) over the in-order sequence
        self._parent = _parent

    # ---- the node's own interval -------------------------------------------
    def _bounds(self):
        return (0, len(self._core)) if self._span is None else self._span

    def _mid(self):
        lo, hi = self._bounds()
        return self._core.order()[(lo + hi) // 2]

    @property
    def start(self):
        return self._core.starts[self._mid()]

    @property
    def end(self):
        return self._core.ends[self._mid()]

    @property
    def interval(self):
        return self._core.values[self._mid()]

    @property
    def left_node(self):
        lo, hi = self._bounds()
        mid = (lo + hi) // 2
        return IntervalNode(0, 0, None, self._core, (lo, mid), self) if mid > lo else None

    @property
    def right_node(self):
        lo, hi = self._bounds()
        mid = (lo + hi) // 2
        return IntervalNode(0, 0, None, self._core, (mid + 1, hi), self) if hi > mid + 1 else None

    @property
    def root_node(self):
        return self._parent

    def __repr__(self):
        return "IntervalNode(%i, %i)" % (self.start, self.end)

    # ---- operations ----------------------------------------------------------
    def insert(self, start, end, interval):
        """Insert a new interval; returns the (possibly new) root, like intersection.pyx:103-138."""
        self._core.insert(_cint(start), _cint(end), interval)
        return self if self._span is None else IntervalNode(0, 0, None, self._core)

    def intersect(self, start, end, sort=True):
        """Same as IntervalTree.find (the `sort` argument is accepted and ignored, as in the reference)."""
        return self._core.find(_cint(start), _cint(end))

    find = intersect

    def left(self, position, n=1, max_dist=2500):
        return self._core.left(position, n, max_dist)

    def right(self, position, n=1, max_dist=2500):
        return self._core.right(position, n, max_dist)

    def traverse(self, func):
        lo, hi = self._bounds()
        core = self._core
        order = core.order()
        for k in range(lo, hi):
            func(_Leaf(core, order[k]))


class _Leaf:
    """What traverse() hands to the callback: .start/.end/.interval of one stored interval."""

    __slots__ = ("start", "end", "interval")

    def __init__(self, core, i):
        self.start = core.starts[i]
        self.end = core.ends[i]
        self.interval = core.values[i]

    def __repr__(self):
        return "IntervalNode(%i, %i)" % (self.start, self.end)


class IntervalTree:
    """
    Window-overlap queries over a set of 1-d half-open intervals; same public methods and
    results as the reference class (intersection.pyx:325-485), served from the device index.

    >>> intersecter = IntervalTree()
    >>> intersecter.insert( 0, 10, "food" )
    >>> intersecter.insert( 3, 7, dict(foo='bar') )
    >>> intersecter.find( 2, 5 )
    ['food', {'foo': 'bar'}]
    """

    def __init__(self):
        self._core = None

    def _c(self):
        if self._core is None:
            self._core = _Core()
        return self._core

    # ---- Position based interfaces -----------------------------------------
    def insert(self, start, end, value=None):
        """Store [start, end) with payload `value` (any int32 pair is accepted, like the reference)."""
        self._c().insert(_cint(start), _cint(end), value)

    add = insert

    def find(self, start, end):
        """Payloads of every stored interval with end > start_q and start < end_q, in tree order."""
        if self._core is None:
            return []
        return self._core.find(_cint(start), _cint(end))

    def before(self, position, num_intervals=1, max_dist=2500):
        """Up to `num_intervals` nearest intervals ending before `position`, within `max_dist`."""
        if self._core is None:
            return []
        return self._core.left(position, num_intervals, max_dist)

    def after(self, position, num_intervals=1, max_dist=2500):
        """Up to `num_intervals` nearest intervals starting after `position`, within `max_dist`."""
        if self._core is None:
            return []
        return self._core.right(position, num_intervals, max_dist)

    # ---- Interval-like object based interfaces -----------------------------
    def insert_interval(self, interval):
        """insert(obj.start, obj.end, obj) for any object with those two attributes."""
        self.insert(interval.start, interval.end, interval)

    add_interval = insert_interval

    def before_interval(self, interval, num_intervals=1, max_dist=2500):
        if self._core is None:
            return []
        return self._core.left(interval.start, num_intervals, max_dist)

    def after_interval(self, interval, num_intervals=1, max_dist=2500):
        if self._core is None:
            return []
        return self._core.right(interval.end, num_intervals, max_dist)

    def upstream_of_interval(self, interval, num_intervals=1, max_dist=2500):
        if self._core is None:
            return []
        if interval.strand == -1 or interval.strand == "-":
            return self._core.right(interval.end, num_intervals, max_dist)
        return self._core.left(interval.start, num_intervals, max_dist)

    def downstream_of_interval(self, interval, num_intervals=1, max_dist=2500):
        if self._core is None:
            return []
        if interval.strand == -1 or interval.strand == "-":
            return self._core.left(interval.start, num_intervals, max_dist)
        return self._core.right(interval.end, num_intervals, max_dist)

    def traverse(self, fn):
        """fn(node) for every stored interval, in tree order (node has .start/.end/.interval)."""
        if self._core is None:
            return None
        return IntervalNode(0, 0, None, self._core).traverse(fn)

    # ---- additive batch API (not in the reference) ---------------------------
    def insert_batch(self, starts, ends, values=None):
        """insert(starts[i], ends[i], values[i]) for all i (values default to None)."""
        c = self._c()
        s = _ffi_as_i32(starts).tolist()  # same range check as insert(): OverflowError beyond a C int
        e = _ffi_as_i32(ends).tolist()
        if len(s) != len(e) or (values is not None and len(values) != len(s)):
            raise ValueError("insert_batch: starts, ends and values must have the same length")
        c.starts.extend(s)
        c.ends.extend(e)
        c.values.extend(values if values is not None else [None] * len(s))
        c._order = None

    def count_batch(self, starts, ends):
        """len(find(starts[i], ends[i])) for all i -> (int32 array, total)."""
        if self._core is None:
            n = len(starts)
            return np.zeros(n, dtype=np.int32), 0
        self._core._flush()
        return self._core.index.count(starts, ends)

    def find_batch(self, starts, ends):
        """CSR (offsets, payload indices) of find(starts[i], ends[i]); map indices with .values."""
        if self._core is None:
            return np.zeros(len(starts) + 1, dtype=np.int64), np.empty(0, dtype=np.int32)
        self._core._flush()
        return self._core.index.find(starts, ends)

    @property
    def values(self):
        return self._c().values


# For backward compatibility
Intersecter = IntervalTree
