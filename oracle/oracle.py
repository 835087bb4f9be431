"""
oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes bindings of the CPU restatement (oracle/liboracle.so, built from
ivtree.c + binbits.c) and, when present, of oracle/_ref/libbinbits_ref.so (the
reference's own src/binBits.c + src/kent/*.c compiled in place).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.

The Python-level argument checks restate lib/bx/bitset.pyx:177-203 (they run
before the C call in the reference as well).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_SANITIZED=1: the same sources built with -fsanitize=address,undefined (make san); the process must have
# been started with LD_PRELOAD=$(gcc -print-file-name=libasan.so) -- tests/test_oracle_golden.py does that in a child
_SAN = os.environ.get("ORACLE_SANITIZED") == "1"
_LIB = os.path.join(_HERE, "liboracle_san.so" if _SAN else "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libbinbits_ref.so")
_REF_CLUSTER = os.path.join(_HERE, "_ref", "libcluster_ref.so")

MAX_INT = 2147483647  # bitset.pyx:105
MAX = 512 * 1024 * 1024  # bitset.pyx:196

_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liboracle.so (and _ref when the reference tree is mounted)."""
    srcs = [os.path.join(_HERE, f) for f in ("ivtree.c", "binbits.c", "cluster.c")]
    stale = force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(_LIB)], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(_REF) or not os.path.exists(_REF_CLUSTER)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        vp = C.c_void_p
        L.ivt_new.restype = vp
        L.ivt_free.argtypes = [vp]
        L.ivt_size.restype = C.c_int64
        L.ivt_size.argtypes = [vp]
        L.ivt_insert.argtypes = [vp, C.c_int32, C.c_int32]
        L.ivt_insert_many.argtypes = [vp, _i32p, _i32p, C.c_int64]
        L.ivt_find.restype = C.c_int64
        L.ivt_find.argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_int64]
        L.ivt_count_batch.argtypes = [vp, _i32p, _i32p, C.c_int64, vp, C.POINTER(C.c_int64)]
        L.ivt_find_batch.restype = C.c_int64
        L.ivt_find_batch.argtypes = [vp, _i32p, _i32p, C.c_int64, _i64p, vp, C.c_int64]
        for f in (L.ivt_seek_left, L.ivt_seek_right):
            f.restype = C.c_int64
            f.argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_int64]
        L.ivt_traverse.restype = C.c_int64
        L.ivt_traverse.argtypes = [vp, vp, C.c_int64]

        L.obb_alloc.restype = vp
        L.obb_alloc.argtypes = [C.c_int32, C.c_int32]
        L.obb_free.argtypes = [vp]
        for name in ("obb_size", "obb_bin_size", "obb_nbins"):
            getattr(L, name).restype = C.c_int32
            getattr(L, name).argtypes = [vp]
        L.obb_bin_state.restype = C.c_int32
        L.obb_bin_state.argtypes = [vp, C.c_int32]
        L.obb_read.restype = C.c_int32
        L.obb_read.argtypes = [vp, C.c_int32]
        L.obb_set.argtypes = [vp, C.c_int32]
        L.obb_clear.argtypes = [vp, C.c_int32]
        L.obb_set_range.argtypes = [vp, C.c_int32, C.c_int32]
        L.obb_count_range.restype = C.c_int32
        L.obb_count_range.argtypes = [vp, C.c_int32, C.c_int32]
        L.obb_next_set.restype = C.c_int32
        L.obb_next_set.argtypes = [vp, C.c_int32]
        L.obb_next_clear.restype = C.c_int32
        L.obb_next_clear.argtypes = [vp, C.c_int32]
        L.obb_and.argtypes = [vp, vp]
        L.obb_or.argtypes = [vp, vp]
        L.obb_not.argtypes = [vp]
        L.obb_set_ranges.argtypes = [vp, _i32p, _i32p, C.c_int64]
        L.obb_count_ranges.argtypes = [vp, _i32p, _i32p, C.c_int64, _i32p]
        L.obb_runs.restype = C.c_int64
        L.obb_runs.argtypes = [vp, vp, vp, C.c_int64]
        L.obb_unpack.argtypes = [vp, _u8p]
        L.oracle_clusters.restype = C.c_int64
        L.oracle_clusters.argtypes = [_i32p, _i32p, vp, C.c_int64, C.c_int32, _i32p, _i32p, _i64p, _i32p]
        _lib = L
    return _lib


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# --------------------------------------------------------------------------- #
# Interval treap (intersection.pyx)
# --------------------------------------------------------------------------- #
class OracleIntervalTree:
    """Restated IntervalTree; payloads are insertion indices."""

    def __init__(self):
        self._h = lib().ivt_new()
        self.starts = []
        self.ends = []

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.ivt_free(self._h)
            self._h = None

    def __len__(self):
        return lib().ivt_size(self._h)

    def insert(self, start, end):
        lib().ivt_insert(self._h, start, end)
        self.starts.append(start)
        self.ends.append(end)

    def insert_many(self, starts, ends):
        s, e = _i32(starts), _i32(ends)
        lib().ivt_insert_many(self._h, s, e, len(s))
        self.starts.extend(s.tolist())
        self.ends.extend(e.tolist())

    def insert_many_arrays(self, starts, ends):
        """Same as insert_many but without keeping Python lists (large N)."""
        s, e = _i32(starts), _i32(ends)
        lib().ivt_insert_many(self._h, s, e, len(s))

    def find(self, start, end):
        n = lib().ivt_find(self._h, start, end, None, 0)
        out = np.empty(n, dtype=np.int32)
        if n:
            lib().ivt_find(self._h, start, end, out.ctypes.data, n)
        return out

    def count_batch(self, qs, qe, want_counts=True):
        qs, qe = _i32(qs), _i32(qe)
        out = np.empty(len(qs), dtype=np.int32) if want_counts else None
        tot = C.c_int64(0)
        lib().ivt_count_batch(self._h, qs, qe, len(qs), out.ctypes.data if want_counts else None, C.byref(tot))
        return out, tot.value

    def find_batch(self, qs, qe):
        qs, qe = _i32(qs), _i32(qe)
        offs = np.empty(len(qs) + 1, dtype=np.int64)
        tot = lib().ivt_find_batch(self._h, qs, qe, len(qs), offs, None, 0)
        hits = np.empty(tot, dtype=np.int32)
        if tot:
            lib().ivt_find_batch(self._h, qs, qe, len(qs), offs, hits.ctypes.data, tot)
        return offs, hits

    def traverse(self):
        n = len(self)
        out = np.empty(n, dtype=np.int32)
        lib().ivt_traverse(self._h, out.ctypes.data, n)
        return out

    def _seek(self, fn, position, max_dist):
        n = fn(self._h, position, max_dist, None, 0)
        out = np.empty(n, dtype=np.int32)
        if n:
            fn(self._h, position, max_dist, out.ctypes.data, n)
        return out.tolist()

    # intersection.pyx:232-245
    def left(self, position, n=1, max_dist=2500):
        r = self._seek(lib().ivt_seek_left, position, max_dist)
        if len(r) == n:
            return r
        r.sort(key=lambda i: self.ends[i], reverse=True)
        return r[:n]

    # intersection.pyx:247-260
    def right(self, position, n=1, max_dist=2500):
        r = self._seek(lib().ivt_seek_right, position, max_dist)
        if len(r) == n:
            return r
        r.sort(key=lambda i: self.starts[i])
        return r[:n]


# --------------------------------------------------------------------------- #
# Binned bitset (bitset.pyx over binBits.c)
# --------------------------------------------------------------------------- #
class _BinBitsBase:
    """bitset.pyx:177-241 restated on top of a C backend (ours or the reference's)."""

    def _check_index(self, index):  # bitset.pyx:177-181
        if index < 0:
            raise IndexError("BitSet index (%d) must be non-negative." % index)
        if index >= self.size:
            raise IndexError("%d is larger than the size of this BitSet (%d)." % (index, self.size))

    def _check_range_count(self, start, count):  # bitset.pyx:184-189
        self._check_index(start)
        if count < 0:
            raise IndexError("Count (%d) must be non-negative." % count)
        if start + count > self.size:
            raise IndexError("End (%d) is larger than the size of this BinnedBitSet (%d)." % (start + count, self.size))

    def _check_same(self, other):  # bitset.pyx:190-192
        if self.size != other.size:
            raise ValueError("BitSets must have the same size")


class OracleBinnedBitSet(_BinBitsBase):
    def __init__(self, size=MAX, granularity=1024):
        if size > MAX_INT:  # bitset.pyx:201-202
            raise ValueError("%d is larger than the maximum BinnedBitSet size of %d." % (size, MAX_INT))
        self._h = lib().obb_alloc(size, granularity)
        self.size = lib().obb_size(self._h)
        self.bin_size = lib().obb_bin_size(self._h)
        self.nbins = lib().obb_nbins(self._h)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.obb_free(self._h)
            self._h = None

    def __getitem__(self, index):
        self._check_index(index)
        return lib().obb_read(self._h, index)

    def set(self, index):
        self._check_index(index)
        lib().obb_set(self._h, index)

    def clear(self, index):
        self._check_index(index)
        lib().obb_clear(self._h, index)

    def set_range(self, start, count):
        self._check_range_count(start, count)
        lib().obb_set_range(self._h, start, count)

    def count_range(self, start, count):
        self._check_range_count(start, count)
        return lib().obb_count_range(self._h, start, count)

    def next_set(self, start):
        self._check_index(start)
        return lib().obb_next_set(self._h, start)

    def next_clear(self, start):
        self._check_index(start)
        return lib().obb_next_clear(self._h, start)

    def iand(self, other):
        self._check_same(other)
        lib().obb_and(self._h, other._h)

    def ior(self, other):
        self._check_same(other)
        lib().obb_or(self._h, other._h)

    def invert(self):
        lib().obb_not(self._h)

    # batch helpers (loops in C; arguments assumed valid)
    def set_ranges(self, starts, lens):
        s, l = _i32(starts), _i32(lens)
        lib().obb_set_ranges(self._h, s, l, len(s))

    def count_ranges(self, starts, lens):
        s, l = _i32(starts), _i32(lens)
        out = np.empty(len(s), dtype=np.int32)
        lib().obb_count_ranges(self._h, s, l, len(s), out)
        return out

    def runs(self):
        n = lib().obb_runs(self._h, None, None, 0)
        rs = np.empty(n, dtype=np.int32)
        re = np.empty(n, dtype=np.int32)
        if n:
            lib().obb_runs(self._h, rs.ctypes.data, re.ctypes.data, n)
        return rs, re

    def states(self):
        return np.array([lib().obb_bin_state(self._h, i) for i in range(self.nbins)], dtype=np.uint8)

    def unpack(self):
        out = np.empty(self.size, dtype=np.uint8)
        lib().obb_unpack(self._h, out)
        return out


class _RefBinBitsStruct(C.Structure):  # src/binBits.h:7-13
    _fields_ = [("size", C.c_int), ("bin_size", C.c_int), ("nbins", C.c_int), ("bins", C.c_void_p)]


_ref = None


def have_ref():
    return os.path.exists(_REF)


def ref_lib():
    """The reference's own C (binBits.h:15-26), compiled in place by `make ref`."""
    global _ref
    if _ref is None:
        R = C.CDLL(_REF)
        sp = C.POINTER(_RefBinBitsStruct)
        R.binBitsAlloc.restype = sp
        R.binBitsAlloc.argtypes = [C.c_int, C.c_int]
        R.binBitsFree.argtypes = [sp]
        R.binBitsReadOne.restype = C.c_int
        R.binBitsReadOne.argtypes = [sp, C.c_int]
        R.binBitsSetOne.argtypes = [sp, C.c_int]
        R.binBitsClearOne.argtypes = [sp, C.c_int]
        R.binBitsSetRange.argtypes = [sp, C.c_int, C.c_int]
        R.binBitsCountRange.restype = C.c_int
        R.binBitsCountRange.argtypes = [sp, C.c_int, C.c_int]
        R.binBitsFindSet.restype = C.c_int
        R.binBitsFindSet.argtypes = [sp, C.c_int]
        R.binBitsFindClear.restype = C.c_int
        R.binBitsFindClear.argtypes = [sp, C.c_int]
        R.binBitsAnd.argtypes = [sp, sp]
        R.binBitsOr.argtypes = [sp, sp]
        R.binBitsNot.argtypes = [sp]
        _ref = R
    return _ref


class RefBinnedBitSet(_BinBitsBase):
    """bitset.pyx's BinnedBitSet re-hosted on the reference's compiled C."""

    def __init__(self, size=MAX, granularity=1024):
        if size > MAX_INT:
            raise ValueError("%d is larger than the maximum BinnedBitSet size of %d." % (size, MAX_INT))
        self._p = ref_lib().binBitsAlloc(size, granularity)
        self.size = self._p.contents.size
        self.bin_size = self._p.contents.bin_size
        self.nbins = self._p.contents.nbins

    def __del__(self):
        if getattr(self, "_p", None) and _ref is not None:
            _ref.binBitsFree(self._p)
            self._p = None

    def __getitem__(self, index):
        self._check_index(index)
        return ref_lib().binBitsReadOne(self._p, index)

    def set(self, index):
        self._check_index(index)
        ref_lib().binBitsSetOne(self._p, index)

    def clear(self, index):
        self._check_index(index)
        ref_lib().binBitsClearOne(self._p, index)

    def set_range(self, start, count):
        self._check_range_count(start, count)
        ref_lib().binBitsSetRange(self._p, start, count)

    def count_range(self, start, count):
        self._check_range_count(start, count)
        return ref_lib().binBitsCountRange(self._p, start, count)

    def next_set(self, start):
        self._check_index(start)
        return ref_lib().binBitsFindSet(self._p, start)

    def next_clear(self, start):
        self._check_index(start)
        return ref_lib().binBitsFindClear(self._p, start)

    def iand(self, other):
        self._check_same(other)
        ref_lib().binBitsAnd(self._p, other._p)

    def ior(self, other):
        self._check_same(other)
        ref_lib().binBitsOr(self._p, other._p)

    def invert(self):
        ref_lib().binBitsNot(self._p)


# ---------------------------------------------------------------- ClusterTree (src/cluster.c, cluster.pyx) --
def cluster_regions(starts, ends, ids, max_dist, min_intervals=0):
    """getregions() of a ClusterTree(max_dist, min_intervals) holding the given intervals (oracle/cluster.c)."""
    s, e = _i32(starts), _i32(ends)
    n = len(s)
    idv = _i32(ids) if ids is not None else None
    c_start, c_end = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.int32)
    c_off, members = np.zeros(n + 1, np.int64), np.empty(max(n, 1), np.int32)
    nc = lib().oracle_clusters(s, e, idv.ctypes.data if idv is not None else None, n, int(max_dist), c_start, c_end, c_off, members)
    if nc < 0:
        raise ValueError("oracle_clusters: bad input (negative max_dist is insertion-order dependent in the reference)")
    out = []
    for c in range(nc):
        lo, hi = int(c_off[c]), int(c_off[c + 1])
        if hi - lo >= min_intervals:
            out.append((int(c_start[c]), int(c_end[c]), members[lo:hi].tolist()))
    return out


class _RefInterval(C.Structure):  # src/cluster.h: struct_interval
    pass


_RefInterval._fields_ = [("start", C.c_int), ("end", C.c_int), ("id", C.c_int), ("next", C.POINTER(_RefInterval))]


class _RefClusterNode(C.Structure):  # src/cluster.h: struct_clusternode (leading fields)
    _fields_ = [("start", C.c_int), ("end", C.c_int), ("priority", C.c_int), ("interval_head", C.POINTER(_RefInterval)),
                ("interval_tail", C.POINTER(_RefInterval)), ("num_ivals", C.c_int)]


class _RefClusterTree(C.Structure):  # src/cluster.h: struct_clustertree
    _fields_ = [("max_dist", C.c_int), ("min_intervals", C.c_int), ("root", C.c_void_p)]


class _RefTreeItr(C.Structure):
    pass


_RefTreeItr._fields_ = [("next", C.POINTER(_RefTreeItr)), ("node", C.POINTER(_RefClusterNode))]

_ref_cluster = None


def have_ref_cluster():
    return os.path.exists(_REF_CLUSTER)


def ref_cluster_regions(triples, max_dist, min_intervals):
    """cluster.pyx's getregions() re-hosted on the reference's compiled src/cluster.c: inserts in the given order."""
    global _ref_cluster
    if _ref_cluster is None:
        R = C.CDLL(_REF_CLUSTER)
        R.create_clustertree.restype = C.POINTER(_RefClusterTree)
        R.create_clustertree.argtypes = [C.c_int, C.c_int]
        R.clusternode_insert.restype = C.c_void_p
        R.clusternode_insert.argtypes = [C.POINTER(_RefClusterTree), C.c_void_p, C.c_int, C.c_int, C.c_int]
        R.clusteritr.restype = C.POINTER(_RefTreeItr)
        R.clusteritr.argtypes = [C.POINTER(_RefClusterTree)]
        R.freeclusteritr.argtypes = [C.POINTER(_RefTreeItr)]
        R.free_tree.argtypes = [C.POINTER(_RefClusterTree)]
        _ref_cluster = R
    R = _ref_cluster
    tree = R.create_clustertree(int(max_dist), int(min_intervals))
    for s, e, i in triples:
        tree.contents.root = R.clusternode_insert(tree, tree.contents.root, int(s), int(e), int(i))
    out = []
    itr = R.clusteritr(tree)
    head = itr
    while itr:
        node = itr.contents.node.contents
        ids, iv = [], node.interval_head
        while iv:
            ids.append(iv.contents.id)
            iv = iv.contents.next
        out.append((node.start, node.end, sorted(ids)))
        itr = itr.contents.next
    if head:
        R.freeclusteritr(head)
    R.free_tree(tree)
    return out
