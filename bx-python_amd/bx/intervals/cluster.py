"""
Drop-in `bx.intervals.cluster` on the MI355X engine (reference: lib/bx/intervals/cluster.pyx, src/cluster.c).

`ClusterTree(mincols, minregions)` groups intervals that lie within `mincols` of each other.  The
reference keeps a self-balancing tree of clusters and merges on every insert; here inserts are
collected and the clusters of the whole set are computed in one go on the device the first time
they are asked for (sort by start, running maximum of the ends, a boundary wherever
`start - mincols > largest end so far`) -- for mincols >= 0 the reference's outcome does not
depend on the insertion order, so the results are identical.

`mincols = -1` ("overlap by one base or more") is answered too while every interval has a positive length: the committed
experiment on the reference's own C (tests/golden/cluster_negative_distance.txt, with the script that made it) finds
one answer there, the same sweep.  Distances below -1, and -1 with zero-length intervals, are refused (ValueError on query),
because the reference has no answer to reproduce: an interval can then be both "to the right of" and "to the left of" a
cluster (src/cluster.c:224-230, the first test wins), the tree stops being ordered by position, and which cluster a later
interval meets depends on the tree's shape -- i.e. on the node priorities, which come from the process-wide unseeded rand()
(src/cluster.c:66-69) and so on how many nodes any earlier tree in the process has created.
"""
from bxmi.intervals import IntervalIndex

__all__ = ["ClusterTree"]

_INT_MIN, _INT_MAX = -(2**31), 2**31 - 1


def _cint(x):
    # the C signature takes ints: anything else overflows or fails the way Cython's coercion does
    v = int(x)
    if v < _INT_MIN or v > _INT_MAX:
        raise OverflowError("value too large to convert to int")
    return v


class ClusterTree:
    """cluster.pyx:57-121"""

    def __init__(self, mincols, minregions):
        self.mincols = _cint(mincols)
        self.minregions = _cint(minregions)
        self._s, self._e, self._ids = [], [], []
        self._regions = None

    def insert(self, s, e, id):
        """Insert an interval with start, end, id as parameters (cluster.pyx:69-72)."""
        if s > e:
            raise ValueError("Interval start must be before end")
        s, e, id = _cint(s), _cint(e), _cint(id)
        self._s.append(s), self._e.append(e), self._ids.append(id)
        self._regions = None

    def _compute(self):
        if self._regions is None:
            regions = []
            if self._s:
                if self.mincols < -1 or (self.mincols == -1 and any(b <= a for a, b in zip(self._s, self._e))):
                    raise ValueError("ClusterTree with a distance below -1 (or -1 and zero-length intervals) depends on the insertion "
                                     "order and on rand() in the reference; not supported")
                ix = IntervalIndex()
                ix.append(self._s, self._e)
                starts, ends, offsets, members = ix.clusters(self.mincols, self._ids)
                members = members.tolist()
                for c, (a, b) in enumerate(zip(starts.tolist(), ends.tolist())):
                    lo, hi = int(offsets[c]), int(offsets[c + 1])
                    if hi - lo >= self.minregions:  # src/cluster.c:190 (num_ivals >= min_intervals)
                        regions.append((a, b, members[lo:hi]))
                ix.close()
            self._regions = regions
        return self._regions

    def getregions(self):
        """Clusters in ascending order of start: (start, end, [sorted ids]) (cluster.pyx:74-98)."""
        return [(a, b, list(ids)) for a, b, ids in self._compute()]

    def getlines(self):
        """The ids of all clustered intervals, cluster by cluster (cluster.pyx:100-121)."""
        return [i for _, _, ids in self._compute() for i in ids]
