"""
Multi-GPU sharding of the hot path.

The path is embarrassingly parallel: intervals on different chromosomes never
interact, and queries against one chromosome are independent.  So
  * chromosomes are dealt to ranks by LPT bin packing (largest first, each to
    the least loaded rank) -- deterministic, identical on every rank;
  * a single-chromosome workload splits its QUERIES into contiguous blocks and
    replicates the (small) target index on every GPU;
  * the only collective is an int64 sum of per-chromosome overlap counts
    (torch.distributed all_reduce: RCCL over xGMI on GPUs, gloo in CPU tests).
One process per GPU; results that are lists are concatenated on the host.
"""
import numpy as np


def lpt_assign(weights, nranks):
    """weights: {key: cost}.  Returns a list (len nranks) of key lists, heaviest keys first."""
    if nranks < 1:
        raise ValueError("nranks must be >= 1")
    loads = [0] * nranks
    out = [[] for _ in range(nranks)]
    for key, w in sorted(weights.items(), key=lambda kv: (-kv[1], str(kv[0]))):
        r = min(range(nranks), key=lambda i: (loads[i], i))
        out[r].append(key)
        loads[r] += w
    return out


def balance(weights, assignment):
    """max load / mean load of an assignment (1.0 = perfect)."""
    loads = [sum(weights[k] for k in part) for part in assignment]
    return max(loads) / (sum(loads) / len(loads)) if sum(loads) else 1.0


def query_block(n, rank, world):
    """Contiguous [lo, hi) block of n queries owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_counts(per_key, keys, device=None):
    """Sum {key: int} over all ranks -> {key: int} for every key in `keys` (same order on all ranks).

    Ranks contribute 0 for chromosomes they do not own.  Without an initialised
    process group this is the identity (single GPU)."""
    import torch
    import torch.distributed as dist

    vec = torch.tensor([int(per_key.get(k, 0)) for k in keys], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return dict(zip(keys, vec.tolist()))


class Comm:
    """The C ABI's collective (include/bxmi.h: bxmi_comm_* / bxmi_allreduce_i64, RCCL underneath): what a host without
    torch would use.  `exchange(bytes_or_None) -> bytes` carries rank 0's 128-byte id to the other ranks -- any channel
    the launcher offers; with torch.distributed initialised, `Comm.from_torch()` broadcasts it there."""

    def __init__(self, rank, world, exchange):
        import ctypes as C

        from . import _ffi

        ident = C.create_string_buffer(128)
        if rank == 0:
            _ffi.call("bxmi_comm_unique_id", ident)
        raw = exchange(ident.raw if rank == 0 else None)
        ident = C.create_string_buffer(bytes(raw), 128)
        h = C.c_void_p()
        _ffi.call("bxmi_comm_create", C.byref(h), ident, rank, world)
        self._h, self.rank, self.world = h, rank, world

    @classmethod
    def from_torch(cls):
        import torch
        import torch.distributed as dist

        rank, world = dist.get_rank(), dist.get_world_size()

        def exchange(raw):
            box = [raw]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        return cls(rank, world, exchange)

    def allreduce_i64(self, dev_ptr, n, stream=None):
        """In place on `n` int64 at device address `dev_ptr`, ordered on `stream`."""
        from . import _ffi

        _ffi.call("bxmi_allreduce_i64", self._h, dev_ptr, n, stream)

    def close(self):
        from . import _ffi

        if self._h:
            _ffi.call("bxmi_comm_destroy", self._h)
            self._h = None


def gather_concat(arr, rank_order_key=None):
    """Host-side concatenation of per-rank int arrays in rank order (hit lists / per-query counts)."""
    import torch.distributed as dist

    a = np.ascontiguousarray(arr)
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return a
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, a)
    return np.concatenate(parts) if parts else a


def count_genome(targets, queries, rank=0, world=1, device=None, counter=None, weights=None):
    """Whole-genome overlap count, sharded by chromosome.

    targets / queries: {chrom: (start int32[], end int32[])}.  `counter(ts, te, qs, qe) -> (counts, total)`
    defaults to the MI355X engine (IntervalIndex.count); tests inject a stand-in.
    `weights` ({chrom: cost}) overrides the default cost targets + queries -- a rank that generated only its own
    chromosomes passes the planned sizes so that every rank deals the chromosomes identically.
    Returns ({chrom: total overlaps}, {chrom: per-query int32 counts} for the chromosomes this rank owns)."""
    chroms = [c for c in queries if c in targets]
    if weights is None:
        weights = {c: len(targets[c][0]) + len(queries[c][0]) for c in chroms}
    mine = lpt_assign({c: weights[c] for c in chroms}, world)[rank]
    totals, per_query = {}, {}
    if counter is None:
        per_query, totals = _count_owned(targets, queries, mine)
    else:
        for c in mine:
            counts, total = counter(targets[c][0], targets[c][1], queries[c][0], queries[c][1])
            totals[c] = total
            per_query[c] = counts
    return allreduce_counts(totals, chroms, device), per_query


def _count_owned(targets, queries, mine):
    """The MI355X engine on this rank's chromosomes: one index per chromosome, all of them queried in ONE fused pass
    (bxmi_ivl_count_multi_dev) -> ({chrom: int32 counts}, {chrom: total})."""
    from . import _ffi
    from .intervals import IntervalIndex

    own = [c for c in mine if len(queries[c][0])]
    ixs, dq, dc, dt = [], [], [], _ffi.DeviceArray(8 * max(1, len(own)))
    dt.zero()
    for c in own:
        ix = IntervalIndex()
        ix.append(targets[c][0], targets[c][1])
        ix.seal()
        ixs.append(ix)
        qs, qe = _ffi.as_i32(queries[c][0]), _ffi.as_i32(queries[c][1])
        dq.append((_ffi.DeviceArray.from_numpy(qs), _ffi.DeviceArray.from_numpy(qe)))
        dc.append(_ffi.DeviceArray(4 * len(qs)))
    IntervalIndex.count_multi_dev(ixs, [a.ptr for a, _ in dq], [b.ptr for _, b in dq], [len(queries[c][0]) for c in own], [d.ptr for d in dc],
                                  [dt.ptr + 8 * i for i in range(len(own))], None)
    _ffi.call("bxmi_synchronize", None)
    tot = dt.to_numpy(np.int64, len(own)) if own else np.zeros(0, np.int64)
    per_query = {c: d.to_numpy(np.int32, len(queries[c][0])) for c, d in zip(own, dc)}
    totals = {c: int(t) for c, t in zip(own, tot)}
    for c in mine:
        if c not in per_query:
            per_query[c], totals[c] = np.zeros(0, np.int32), 0
    for ix in ixs:
        ix.close()
    return per_query, totals
