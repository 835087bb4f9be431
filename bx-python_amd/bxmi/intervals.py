"""
Batch interval index on the MI355X -- the engine under
``bx.intervals.intersection.IntervalTree`` (reference: lib/bx/intervals/intersection.pyx).

``IntervalIndex`` keeps intervals as int32 SoA arrays in HBM and answers whole
batches of ``find`` queries per kernel launch.  Payloads never leave the host:
the device returns *insertion indices* in the reference's result order.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import as_i32, call, ptr


class IntervalIndex:
    """SoA interval index; insertion index == payload id.

    append() -> seal() -> count()/find() ; appending again un-seals, the next
    query re-seals (the index is rebuilt from the device-resident arrays).
    """

    def __init__(self):
        _ffi.require_gpu()
        h = C.c_void_p()
        call("bxmi_ivl_create", C.byref(h))
        self._h = h
        self._n = 0
        self._sealed = False
        self._one_buf = np.empty(4096, dtype=np.int32)

    def close(self):
        if getattr(self, "_h", None):
            _ffi.load().bxmi_ivl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self._n

    # ---- build -------------------------------------------------------------
    def append(self, starts, ends):
        """IntervalTree.insert(start, end, i) for each pair, in order (intersection.pyx:388-395)."""
        s, e = as_i32(starts), as_i32(ends)
        if s.shape != e.shape or s.ndim != 1:
            raise ValueError("starts and ends must be 1-d arrays of equal length")
        if len(s):
            call("bxmi_ivl_append", self._h, ptr(s), ptr(e), len(s))
            self._n += len(s)
            self._sealed = False

    def append_dev(self, starts_ptr, ends_ptr, n, stream=None):
        call("bxmi_ivl_append_dev", self._h, starts_ptr, ends_ptr, n, stream)
        self._n += n
        self._sealed = False

    def seal(self, stream=None):
        call("bxmi_ivl_seal", self._h, stream)
        self._sealed = True

    def _ready(self):
        if not self._sealed:
            self.seal()

    @property
    def has_reversed(self):
        self._ready()
        f = C.c_int(0)
        call("bxmi_ivl_has_reversed", self._h, C.byref(f))
        return bool(f.value)

    def slice_state(self):
        """(state, unit_keys[7]) of the slice search stage: 1 = usable, -1 = a bucket's keys do not fit the LDS, 0 = undecided."""
        self._ready()
        st, need = C.c_int(0), (C.c_int64 * 7)()
        call("bxmi_ivl_slice_state", self._h, C.byref(st), need)
        return st.value, list(need)

    def flat_state(self):
        """(state, hard_cells) of the flat walk on cell images: 1 = usable, -1 = the index does not qualify, 0 = undecided."""
        self._ready()
        st, hc = C.c_int(0), C.c_int64(0)
        call("bxmi_ivl_flat_state", self._h, C.byref(st), C.byref(hc))
        return st.value, hc.value

    def sparse_state(self):
        """(state, hard_cells, cell_log2) of the offset-cell images a sparse index takes: 1 = usable, -1 = the index does not
        qualify, 0 = undecided; cell_log2 = the cell width chosen from the density (6..8)."""
        self._ready()
        st, hc, k = C.c_int(0), C.c_int64(0), C.c_int(0)
        call("bxmi_ivl_sparse_state", self._h, C.byref(st), C.byref(hc), C.byref(k))
        return st.value, hc.value, k.value

    def count_width(self):
        """(bits, wide_counts): 8 or 16 bits per count between the search and the un-permute kernel of a flat-walk pass, and how
        many counts did not fit 8 bits so far (as last mirrored to the host)."""
        self._ready()
        bits, wide = C.c_int(0), C.c_int64(0)
        call("bxmi_ivl_count_width", self._h, C.byref(bits), C.byref(wide))
        return bits.value, wide.value

    def order_state(self):
        """(skipping, answers_seen): whether large count batches currently go without the order check, and how many order
        reports the host has read."""
        self._ready()
        sk, seen = C.c_int(0), C.c_int64(0)
        call("bxmi_ivl_order_state", self._h, C.byref(sk), C.byref(seen))
        return sk.value, seen.value

    def dense_state(self):
        """(state, [most keys of a block, most overflow entries of a unit]) of the dense-image search stage: 1 = usable,
        -1 = the index does not fit the format, 0 = undecided."""
        self._ready()
        st, worst = C.c_int(0), (C.c_int64 * 2)()
        call("bxmi_ivl_dense_state", self._h, C.byref(st), worst)
        return st.value, list(worst)

    def order(self):
        """Insertion indices in the treap's in-order (== IntervalTree.traverse order)."""
        self._ready()
        out = np.empty(self._n, dtype=np.int32)
        if self._n:
            call("bxmi_ivl_order", self._h, ptr(out))
        return out

    # ---- queries -----------------------------------------------------------
    def count(self, qs, qe, want_counts=True):
        """len(find(qs[i], qe[i])) for every i  ->  (int32[nq] or None, int total)."""
        self._ready()
        qs, qe = as_i32(qs), as_i32(qe)
        if qs.shape != qe.shape or qs.ndim != 1:
            raise ValueError("qs and qe must be 1-d arrays of equal length")
        counts = np.empty(len(qs), dtype=np.int32) if want_counts else None
        total = C.c_int64(0)
        call("bxmi_ivl_count", self._h, ptr(qs), ptr(qe), len(qs), ptr(counts), C.byref(total))
        return counts, total.value

    def find(self, qs, qe, cap_hint=None):
        """Batched find(): CSR (offsets int64[nq+1], hits int32[total]) of insertion indices."""
        self._ready()
        qs, qe = as_i32(qs), as_i32(qe)
        if qs.shape != qe.shape or qs.ndim != 1:
            raise ValueError("qs and qe must be 1-d arrays of equal length")
        nq = len(qs)
        offsets = np.zeros(nq + 1, dtype=np.int64)
        cap = int(cap_hint) if cap_hint is not None else max(1024, 8 * nq)
        total = C.c_int64(0)
        for _ in range(2):
            hits = np.empty(cap, dtype=np.int32)
            rc = call("bxmi_ivl_find", self._h, ptr(qs), ptr(qe), nq, ptr(offsets), ptr(hits), cap, C.byref(total),
                      allow=(_ffi.ERANGE,))
            if rc == _ffi.OK:
                return offsets, hits[: total.value]
            cap = total.value
        raise _ffi.BxmiError(_ffi.ERANGE, "find: hit buffer still too small")

    def find_one(self, qs, qe):
        """find() for one query -> int32 array of insertion indices (one launch + one sync)."""
        self._ready()
        n = C.c_int64(0)
        buf = self._one_buf
        rc = call("bxmi_ivl_find_one", self._h, int(qs), int(qe), ptr(buf), len(buf), C.byref(n), allow=(_ffi.ERANGE,))
        if rc == _ffi.ERANGE:
            buf = self._one_buf = np.empty(int(n.value) * 2, dtype=np.int32)
            call("bxmi_ivl_find_one", self._h, int(qs), int(qe), ptr(buf), len(buf), C.byref(n))
        return buf[: n.value].copy()  # the buffer is reused by the next call

    def find_one_list(self, qs, qe):
        """find() for one query -> Python list of insertion indices: the per-call path of the drop-in tree (one launch,
        the answer polled from host-visible memory); nothing is looked up or allocated that can be kept between calls."""
        if not self._sealed:
            self.seal()
        fast = self.__dict__.get("_fo")
        if fast is None or fast[1] is not self._one_buf:
            n = C.c_int64(0)
            fast = self._fo = (_ffi.load().bxmi_ivl_find_one, self._one_buf, self._one_buf.ctypes.data, n, C.byref(n))
        fn, buf, addr, n, ref = fast
        rc = fn(self._h, qs, qe, addr, len(buf), ref)
        if rc:
            if rc != _ffi.ERANGE:
                _ffi.check(rc)
            self._one_buf = np.empty(int(n.value) * 2, dtype=np.int32)
            return self.find_one_list(qs, qe)
        return buf[: n.value].tolist()

    def count_dev(self, qs_ptr, qe_ptr, nq, counts_ptr, total_ptr, stream=None):
        """Device-pointer form used by bench.py / the sharded driver (no host sync)."""
        self._ready()
        call("bxmi_ivl_count_dev", self._h, qs_ptr, qe_ptr, nq, counts_ptr, total_ptr, stream)

    @staticmethod
    def count_multi_dev(indexes, qs_ptrs, qe_ptrs, nqs, counts_ptrs, total_ptrs=None, stream=None):
        """count_dev for several indexes at once (one per chromosome): same results, one fused pass for those that qualify.
        All arguments are per-index lists (device pointers as ints, total_ptrs entries may be None)."""
        n = len(indexes)
        for ix in indexes:
            ix._ready()
        H = (C.c_void_p * n)(*[ix._h for ix in indexes])
        Q = (C.c_void_p * n)(*qs_ptrs)
        E = (C.c_void_p * n)(*qe_ptrs)
        N = (C.c_int64 * n)(*nqs)
        K = (C.c_void_p * n)(*counts_ptrs)
        T = (C.c_void_p * n)(*total_ptrs) if total_ptrs is not None else None
        call("bxmi_ivl_count_multi_dev", H, n, Q, E, N, K, T, stream)

    def find_dev(self, qs_ptr, qe_ptr, nq, offsets_ptr, hits_ptr, cap, stream=None):
        self._ready()
        total = C.c_int64(0)
        rc = call("bxmi_ivl_find_dev", self._h, qs_ptr, qe_ptr, nq, offsets_ptr, hits_ptr, cap, C.byref(total), stream,
                  allow=(_ffi.ERANGE,))
        return rc, total.value

    def neighbors(self, position, max_dist, direction, cap=4096):
        """Candidate list of IntervalNode.left (direction<0) / right (direction>0), intersection.pyx:192-229."""
        self._ready()
        n_out = C.c_int64(0)
        for _ in range(2):
            out = np.empty(max(cap, 1), dtype=np.int32)
            call("bxmi_ivl_neighbors", self._h, int(position), int(max_dist), int(direction), ptr(out), len(out), C.byref(n_out))
            if n_out.value <= len(out):
                return out[: n_out.value]
            cap = n_out.value
        raise _ffi.BxmiError(_ffi.ERANGE, "neighbors: buffer still too small")

    def clusters(self, max_dist, ids=None):
        """ClusterTree's grouping (cluster.pyx:57-121): intervals chained by gaps <= max_dist.
        -> (starts int32[c], ends int32[c], offsets int64[c+1], members int32[n]); members of a cluster are ids in
        ascending order (ids[insertion index], or the insertion index when ids is None)."""
        self._ready()
        n = self._n
        starts = np.empty(max(n, 1), dtype=np.int32)
        ends = np.empty(max(n, 1), dtype=np.int32)
        offsets = np.zeros(n + 1, dtype=np.int64)
        members = np.empty(max(n, 1), dtype=np.int32)
        nc = C.c_int64(0)
        if ids is not None:
            ids = as_i32(ids)
            if len(ids) != n:
                raise ValueError("ids must have one entry per interval")
        call("bxmi_ivl_clusters", self._h, ptr(ids) if ids is not None else None, int(max_dist), C.byref(nc), ptr(starts), ptr(ends),
             ptr(offsets), ptr(members))
        c = nc.value
        return starts[:c], ends[:c], offsets[: c + 1], members[:n]
