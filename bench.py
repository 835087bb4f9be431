#!/usr/bin/env python3
"""
bench.py -- BASELINE.json's headline metric on the MI355X.

Primary line (configs[1]): M overlap-queries/s, count-only, 100M queries x 10M
targets, one chromosome, int32 SoA, everything resident in HBM when the timed
region starts.  One "step" = one pass of the count kernel over the rank's
100M-query batch.  With --gpus N every rank holds a replica of the 10M-target
index and its own 100M queries (weak scaling, no data-path collective); the only
collective is the optional int64 all-reduce of the overlap total (RCCL).

The same JSON line carries
  roofline      algorithmic bytes (12 B/query + 8 B/target per launch) / measured
                kernel time (HIP events on the launch stream) against 8 TB/s,
  cpu_baseline  the oracle treap (a C port of the reference's algorithm) timed on
                this box's host cores on a bounded sample of the same workload,
  bitset        BASELINE's second metric (BinnedBitSet Gbp/s) on configs[2].

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "bx-python_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


class CpuBaseline(threading.Thread):
    """Builds the oracle treap over the 10M targets in the background (ctypes drops the GIL)."""

    def __init__(self, ts, te):
        super().__init__(daemon=True)
        self.ts, self.te = ts, te
        self.tree = None
        self.build_s = None
        self.error = None

    def run(self):
        try:
            from oracle import oracle as O

            t0 = time.perf_counter()
            t = O.OracleIntervalTree()
            t.insert_many_arrays(self.ts, self.te)
            self.build_s = time.perf_counter() - t0
            self.tree = t
        except Exception as ex:  # the baseline must never take the GPU numbers down with it
            self.error = repr(ex)

    def measure(self, qs, qe, sample):
        self.join()
        if self.tree is None:
            return None
        t0 = time.perf_counter()
        counts, total = self.tree.count_batch(qs[:sample], qe[:sample])
        dt = time.perf_counter() - t0
        return dict(value=round(sample / dt / 1e6, 5), unit="M queries/s", cores=1, kind="port",
                    sample="first %d of the 100M queries against the full 10M-target treap (oracle/ivtree.c, single thread; "
                           "treap build %.1f s not included)" % (sample, self.build_s),
                    seconds=round(dt, 2)), counts


def bench_bitsets(torch, steps, warmup):
    """configs[2]: iand + count_range over two hg19-sized (3.1 Gbp) sets of 24 chromosome bitsets."""
    from bxmi import _ffi, synth
    from bxmi.bitset import DeviceBitSet

    ra = synth.genome_ranges(1_500_000, 301)
    rb = synth.genome_ranges(1_500_000, 302)
    A, Bs = [], []
    for chrom, size in synth.HG19_SIZES.items():
        a, b = DeviceBitSet(size), DeviceBitSet(size)
        a.set_ranges(*ra[chrom])
        b.set_ranges(*rb[chrom])
        A.append(a)
        Bs.append(b)
    bits = sum(synth.HG19_SIZES.values())
    stream = torch.cuda.current_stream().cuda_stream
    acc = torch.zeros(1, dtype=torch.int64, device="cuda")

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps  # ms per pass over the genome

    def popcount():
        acc.zero_()
        for a in A:
            _ffi.call("bxmi_bits_popcount_dev", a._h, acc.data_ptr(), stream)

    def iand():
        for a, b in zip(A, Bs):
            _ffi.call("bxmi_bits_and_dev", a._h, b._h, stream)

    def fused():
        acc.zero_()
        for a, b in zip(A, Bs):
            _ffi.call("bxmi_bits_and_count_dev", a._h, b._h, acc.data_ptr(), stream)

    from bxmi.bitset import BitSetGroup

    gA, gB = BitSetGroup(A), BitSetGroup(Bs)
    per = torch.zeros(len(A), dtype=torch.int64, device="cuda")

    def g_popcount():
        per.zero_()
        _ffi.call("bxmi_bits_group_popcount_dev", gA._g, per.data_ptr(), stream)

    def g_iand():
        _ffi.call("bxmi_bits_group_and_dev", gA._g, gB._g, None, stream)

    def g_fused():
        per.zero_()
        _ffi.call("bxmi_bits_group_and_dev", gA._g, gB._g, per.data_ptr(), stream)

    pop_before = None
    popcount()
    torch.cuda.synchronize()
    pop_before = int(acc.item())
    ms_pop = timed(popcount)
    ms_and = timed(iand)
    ms_fused = timed(fused)
    and_bits = int(acc.item())
    popcount()
    torch.cuda.synchronize()
    assert int(acc.item()) == and_bits, "fused and+count disagrees with a separate popcount"
    ms_gpop, ms_gand, ms_gfused = timed(g_popcount), timed(g_iand), timed(g_fused)
    assert int(per.sum().item()) == and_bits, "group and+count disagrees with the per-chromosome launches"
    group = dict(
        note="same work as ONE launch over all 24 chromosomes (bxmi_bits_group_*)",
        popcount_gbps=round(bits / ms_gpop / 1e6, 1), iand_gbps=round(bits / ms_gand / 1e6, 1), iand_count_fused_gbps=round(bits / ms_gfused / 1e6, 1),
        ms=dict(popcount=round(ms_gpop, 4), iand=round(ms_gand, 4), fused=round(ms_gfused, 4)),
        roofline_frac=dict(popcount=round(bits / 8 / (ms_gpop * 1e6) / HBM_PEAK_GBS, 4), iand=round(3 * bits / 8 / (ms_gand * 1e6) / HBM_PEAK_GBS, 4),
                           fused=round(3 * bits / 8 / (ms_gfused * 1e6) / HBM_PEAK_GBS, 4)),
    )
    out = dict(
        one_launch_per_genome=group,
        workload="configs[2]: 24 hg19-sized chromosome bitsets (3.096 Gbp), two sets of 1.5M ranges, lens=chrom sizes",
        popcount_gbps=round(bits / ms_pop / 1e6, 1), iand_gbps=round(bits / ms_and / 1e6, 1), iand_count_fused_gbps=round(bits / ms_fused / 1e6, 1),
        ms=dict(popcount=round(ms_pop, 4), iand=round(ms_and, 4), fused=round(ms_fused, 4)),
        roofline_frac=dict(popcount=round(bits / 8 / (ms_pop * 1e6) / HBM_PEAK_GBS, 4), iand=round(3 * bits / 8 / (ms_and * 1e6) / HBM_PEAK_GBS, 4),
                           fused=round(3 * bits / 8 / (ms_fused * 1e6) / HBM_PEAK_GBS, 4)),
        bases_set_a=pop_before, bases_in_and=and_bits,
        note="iand is idempotent after the first pass; every pass still moves read A + read B + write A",
    )
    # reference C (oracle/_ref = src/binBits.c compiled in place) on chr21, if it travelled with the repo
    try:
        from oracle import oracle as O

        if O.have_ref():
            size = synth.HG19_SIZES["chr21"]
            ra21, rb21 = ra["chr21"], rb["chr21"]
            x, y = O.RefBinnedBitSet(size), O.RefBinnedBitSet(size)
            R = O.ref_lib()
            for s, n in zip(ra21[0].tolist(), ra21[1].tolist()):
                R.binBitsSetRange(x._p, s, n)
            for s, n in zip(rb21[0].tolist(), rb21[1].tolist()):
                R.binBitsSetRange(y._p, s, n)
            t0 = time.perf_counter()
            c = R.binBitsCountRange(x._p, 0, size)
            t_cnt = time.perf_counter() - t0
            t0 = time.perf_counter()
            R.binBitsAnd(x._p, y._p)
            t_and = time.perf_counter() - t0
            out["cpu_reference"] = dict(kind="reference", cores=1, sample="chr21 (48.1 Mbp) via oracle/_ref (src/binBits.c, gcc -O2)",
                                        popcount_gbps=round(size / t_cnt / 1e9, 2), iand_gbps=round(size / t_and / 1e9, 2), chr21_bases=c)
    except Exception as ex:
        out["cpu_reference_error"] = repr(ex)
    for d in A + Bs:
        d.close()
    # the same two sets with the reference's DEFAULT sizes (no lens: 24 x 512 Mi bits = 1.5 GiB per set)
    try:
        from bxmi.bitset import MAX

        A2, B2 = [], []
        for chrom in synth.HG19_SIZES:
            a, b = DeviceBitSet(MAX), DeviceBitSet(MAX)
            a.set_ranges(*ra[chrom])
            b.set_ranges(*rb[chrom])
            A2.append(a)
            B2.append(b)
        g2a, g2b = BitSetGroup(A2), BitSetGroup(B2)
        bits2 = MAX * len(A2)
        ms_p2 = timed(lambda: (per.zero_(), _ffi.call("bxmi_bits_group_popcount_dev", g2a._g, per.data_ptr(), stream)))
        ms_f2 = timed(lambda: (per.zero_(), _ffi.call("bxmi_bits_group_and_dev", g2a._g, g2b._g, per.data_ptr(), stream)))
        assert int(per.sum().item()) == and_bits, "MAX-sized sets disagree with the lens-sized ones"
        out["default_MAX_sizes"] = dict(
            workload="same ranges, every chromosome a BinnedBitSet(MAX): 24 x 512 Mi bits", popcount_gbps=round(bits2 / ms_p2 / 1e6, 1),
            iand_count_fused_gbps=round(bits2 / ms_f2 / 1e6, 1), ms=dict(popcount=round(ms_p2, 4), fused=round(ms_f2, 4)),
            roofline_frac=dict(popcount=round(bits2 / 8 / (ms_p2 * 1e6) / HBM_PEAK_GBS, 4), fused=round(3 * bits2 / 8 / (ms_f2 * 1e6) / HBM_PEAK_GBS, 4)))
        for d in A2 + B2:
            d.close()
    except Exception as ex:
        out["default_MAX_sizes"] = {"error": repr(ex)}
    return out


def bench_find(torch, reps=3):
    """configs[4] (BASELINE.json: 50M x 50M overlap join, CSR hit list kept in HBM) -- a side measurement, never `value`.
    Generated order and the same queries sorted by start; size-independent checks on the full result."""
    from bxmi import synth
    from bxmi.intervals import IntervalIndex

    nt = nq = 50_000_000
    (ts, te), (qs_h, qe_h) = synth.cfg5(nt, nq)
    ix = IntervalIndex()
    ix.append(ts, te)
    ix.seal()
    cap = nq * 8
    offs = torch.empty(nq + 1, dtype=torch.int64, device="cuda")
    hits = torch.empty(cap, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    out = {"workload": "configs[4]: %d x %d join, G=2e9, len U[1,200], CSR (int64 offsets, int32 hits) in HBM" % (nq, nt)}
    d_ts, d_te = torch.from_numpy(ts).cuda(), torch.from_numpy(te).cuda()
    qs, qe = torch.from_numpy(qs_h).cuda(), torch.from_numpy(qe_h).cuda()
    host = None
    for label in ("generated_order", "sorted_by_start"):
        if label == "sorted_by_start":
            o = torch.argsort(qs, stable=True)
            qs, qe = qs[o].contiguous(), qe[o].contiguous()
            del o
        rc, total = ix.find_dev(qs.data_ptr(), qe.data_ptr(), nq, offs.data_ptr(), hits.data_ptr(), cap, stream)
        if rc != 0:
            return {"error": "find_dev rc=%d" % rc}
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ix.find_dev(qs.data_ptr(), qe.data_ptr(), nq, offs.data_ptr(), hits.data_ptr(), cap, stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        h = hits[:total].long()
        rep = torch.repeat_interleave(torch.arange(nq, device="cuda"), offs[1:] - offs[:-1])
        every_hit_overlaps = bool(((d_te[h] > qs[rep]) & (d_ts[h] < qe[rep])).all().item())
        counts = torch.empty(nq, dtype=torch.int32, device="cuda")
        tot = torch.zeros(1, dtype=torch.int64, device="cuda")
        ix.count_dev(qs.data_ptr(), qe.data_ptr(), nq, counts.data_ptr(), tot.data_ptr(), stream)
        torch.cuda.synchronize()
        counts_match = bool(torch.equal((offs[1:] - offs[:-1]).int(), counts)) and int(tot.item()) == total
        alg = nq * 16 + total * 4 + nt * 8
        out[label] = dict(ms=round(ms, 3), m_queries_per_s=round(nq / ms / 1e3, 1), m_hits_per_s=round(total / ms / 1e3, 1), hits=int(total),
                          frac_of_hbm_peak=round(alg / ms / 1e6 / HBM_PEAK_GBS, 4), every_hit_overlaps=every_hit_overlaps,
                          counts_match_count_path=counts_match)
        del h, rep, counts
        if label == "generated_order":
            # the same join through the HOST-pointer entry point (bxmi_ivl_find on numpy arrays: what the Cython binding of
            # INTEGRATION.md 1b and the CLI counterparts call): 0.4 GB up, 0.4 GB of offsets + 1 GB of hits down into fresh arrays
            ix.find(qs_h[:1 << 20], qe_h[:1 << 20])
            ho, hh = ix.find(qs_h, qe_h, cap_hint=6 * nq)  # (sizes the handle's staging)
            del ho, hh
            t0 = time.perf_counter()
            ho, hh = ix.find(qs_h, qe_h, cap_hint=6 * nq)
            dt = time.perf_counter() - t0
            host = dict(ms=round(dt * 1e3, 2), m_queries_per_s=round(nq / dt / 1e6, 1),
                        same_csr=bool(np.array_equal(ho, offs.cpu().numpy()) and np.array_equal(hh, hits[:total].cpu().numpy())),
                        note="bxmi_ivl_find on pageable numpy arrays, fresh output arrays; host threads touch the outputs' pages ahead of the copies")
            del ho, hh
    out["host_pointers_generated_order"] = host
    return out


def bench_clustered(torch, reps=5, nt=10_000_000, nq=100_000_000, hot_spots=20_000, genome=250_000_000):
    """The non-uniform counterpart of configs[1] -- a side measurement, never `value`: targets and
    queries around 20 000 hot spots with heavily duplicated coordinates (bxmi.synth.clustered's recipe, drawn with the
    device's generator here: 110 M draws take numpy half a minute), in generated order and sorted by start.  Says which
    search stage served the index; the counts are checked against the direct tree kernel on a 4 M-query slice."""
    from bxmi import _ffi
    from bxmi.intervals import IntervalIndex

    g = torch.Generator(device="cuda")
    g.manual_seed(601)
    centres = torch.sort(torch.randint(10_000, genome - 10_000, (hot_spots,), generator=g, device="cuda"))[0]
    weight = torch.exp(0.6 * torch.randn(hot_spots, generator=g, device="cuda", dtype=torch.float64))
    cum = torch.cumsum(weight / weight.sum(), 0)
    lengths = torch.randint(30, 2000, (48,), generator=g, device="cuda")

    def draw(n, scatter):
        k = torch.searchsorted(cum, torch.rand(n, generator=g, device="cuda", dtype=torch.float64)).clamp_(max=hot_spots - 1)
        off = torch.empty(n, device="cuda", dtype=torch.float32).geometric_(1.0 / scatter, generator=g).long() - 1
        start = centres[k] + off
        length = lengths[torch.randint(0, 48, (n,), generator=g, device="cuda")]
        return start.int().contiguous(), (start + length).int().contiguous()

    ts, te = draw(nt, 60.0)
    qs, qe = draw(nq, 400.0)
    ix = IntervalIndex()
    ix.append_dev(ts.data_ptr(), te.data_ptr(), nt)
    ix.seal()
    stream = torch.cuda.current_stream().cuda_stream
    counts = torch.empty(nq, dtype=torch.int32, device="cuda")
    tot = torch.zeros(1, dtype=torch.int64, device="cuda")
    out = {"workload": "%d queries x %d targets around %d hot spots on %d coordinates (geometric scatter 400 / 60, 48 lengths): %d distinct target "
                       "starts" % (nq, nt, hot_spots, genome, int(torch.unique(ts).numel()))}
    for label in ("generated_order", "sorted_by_start"):
        if label == "sorted_by_start":
            o = torch.argsort(qs, stable=True)
            qs, qe = qs[o].contiguous(), qe[o].contiguous()
            del o
        ix.count_dev(qs.data_ptr(), qe.data_ptr(), nq, counts.data_ptr(), tot.data_ptr(), stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ix.count_dev(qs.data_ptr(), qe.data_ptr(), nq, counts.data_ptr(), tot.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        # the same first 4 M queries through the direct tree kernel (no exchange, no images)
        m = 4 << 20
        ref = torch.empty(m, dtype=torch.int32, device="cuda")
        _ffi.call("bxmi_set_option", b"ivl.partition", 0)
        try:
            ix.count_dev(qs.data_ptr(), qe.data_ptr(), m, ref.data_ptr(), None, stream)
            torch.cuda.synchronize()
        finally:
            _ffi.call("bxmi_set_option", b"ivl.partition", -1)
        out[label] = dict(ms=round(ms, 4), m_queries_per_s=round(nq / ms / 1e3, 1), frac_of_hbm_peak=round(alg_bytes_of(nq, nt) / ms / 1e6 / HBM_PEAK_GBS, 4),
                          same_as_direct_kernel=bool(torch.equal(counts[:m], ref)))
    out["search_stage_of_this_index"] = dict(zip(("flat_walk_on_cell_images", "dense_unit_images", "key_slices", "offset_cell_images"),
                                                 (ix.flat_state()[0], ix.dense_state()[0], ix.slice_state()[0], ix.sparse_state()[0])))
    ix.close()
    return out


def bench_per_call():
    """What an UNMODIFIED script pays per line on the drop-in classes (bx.intervals.IntervalTree.find,
    bx.bitset.BinnedBitSet.count_range): one launch and one answer across PCIe per call.  Microseconds per call."""
    import bx.bitset
    import bx.intervals

    rng = np.random.default_rng(1)
    s = rng.integers(0, 10_000_000, size=200_000)
    e = s + rng.integers(1, 1000, size=len(s))
    t = bx.intervals.IntervalTree()
    t.insert_batch(s, e, list(range(len(s))))
    t.find(1, 2)
    q = rng.integers(0, 10_000_000, size=3000).tolist()
    t0 = time.perf_counter()
    hits = 0
    for x in q:
        hits += len(t.find(x, x + 500))
    find_us = (time.perf_counter() - t0) / len(q) * 1e6
    b = bx.bitset.BinnedBitSet()
    for i in range(20000):
        b.set_range(int(s[i]), int(e[i] - s[i]))
    b.count_range(0, 10)
    t0 = time.perf_counter()
    bases = 0
    for x in q:
        bases += b.count_range(x, 500)
    count_us = (time.perf_counter() - t0) / len(q) * 1e6
    # the same calls with the host mirror of the device-extracted run list switched off: every call is a launch and an answer across PCIe
    saved = bx.bitset._MIRROR_MAX_RUNS
    bx.bitset._MIRROR_MAX_RUNS = -1
    try:
        b.set_range(0, 1)  # (a mutation drops the mirror)
        b.count_range(0, 10)
        t0 = time.perf_counter()
        bases_dev = 0
        for x in q:
            bases_dev += b.count_range(x, 500)
        count_dev_us = (time.perf_counter() - t0) / len(q) * 1e6
    finally:
        bx.bitset._MIRROR_MAX_RUNS = saved
    return dict(IntervalTree_find_us=round(find_us, 2), BinnedBitSet_count_range_us=round(count_us, 2),
                BinnedBitSet_count_range_device_every_call_us=round(count_dev_us, 2), calls=len(q), hits=hits, bases=bases,
                mirror_and_device_agree=bool(abs(bases_dev - bases) <= 1),  # (one base was set in between to drop the mirror)
                note="drop-in classes, one call per query, 200k-interval tree / 20k-range bitset; host wall time incl. the Python wrapper; "
                     "count_range: answered from the host mirror of the device-extracted run list (default between mutations) / by a kernel per call")


def genome_golden_check(chrom, counts_np, golden):
    """Owner-side parity of one chromosome: the strided subsample against the reference treap's hash (tests/golden/scale.json)."""
    pt = golden["chroms"].get(chrom) if golden else None
    if not pt:
        return None
    sub = np.ascontiguousarray(counts_np[:: golden["stride"]])
    return hashlib.sha256(sub.tobytes()).hexdigest() == pt["counts_sha256"] and int(sub.sum(dtype=np.int64)) == pt["total"]


def bench_genome(torch, dist, rank, world, steps, warmup, n_targets, n_queries, single_gpu_reference=True):
    """BASELINE configs[3]: the 24 hg19 chromosomes (synth.cfg4: 10M targets x 100M queries spread by length), one
    interval index per chromosome, chromosomes dealt to the ranks by LPT (bxmi.shard), every rank counts its own
    chromosomes, the 24 per-chromosome totals are all-reduced (int64, RCCL).  STRONG scaling: the whole job is the
    same genome whatever the number of GPUs; the time of a step is the slowest rank's.  Returns a dict (rank 0) or None."""
    from bxmi import shard, synth
    from bxmi.intervals import IntervalIndex

    chroms = list(synth.HG19_SIZES)
    tsz, qsz = synth.cfg4_sizes(n_targets), synth.cfg4_sizes(n_queries)
    weights = {c: tsz[c] + qsz[c] for c in chroms}
    assign = shard.lpt_assign(weights, world)
    golden = None
    gpath = os.path.join(ROOT, "tests", "golden", "scale.json")
    if n_targets == 10_000_000 and n_queries == 100_000_000 and os.path.exists(gpath):
        golden = json.load(open(gpath)).get("cfg4_genome")
    stream = torch.cuda.current_stream().cuda_stream

    class Shard:
        def __init__(self, owned):
            self.owned = owned
            self.ix, self.q, self.counts = {}, {}, {}
            self.index_s = 0.0  # append + seal of the owned chromosomes' indexes (the synthetic data's generation apart)
            for c in owned:
                (ts, te), (qs, qe) = synth.cfg4_chrom(c, n_targets, n_queries)
                tb = time.perf_counter()
                ix = IntervalIndex()
                ix.append(ts, te)
                ix.seal()
                self.index_s += time.perf_counter() - tb
                self.ix[c] = ix
                self.q[c] = (torch.from_numpy(qs).cuda(), torch.from_numpy(qe).cuda())
                self.counts[c] = torch.empty(len(qs), dtype=torch.int32, device="cuda")

        def step(self, row):
            # one fused pass over all the chromosomes this rank owns (bxmi_ivl_count_multi_dev); BENCH_PER_CHROM=1 issues the
            # per-chromosome calls of the reference's dict-of-trees loop instead
            if os.environ.get("BENCH_PER_CHROM"):
                for c in self.owned:
                    qs, qe = self.q[c]
                    self.ix[c].count_dev(qs.data_ptr(), qe.data_ptr(), qs.numel(), self.counts[c].data_ptr(), row[chroms.index(c):].data_ptr(), stream)
                return
            own = self.owned
            IntervalIndex.count_multi_dev([self.ix[c] for c in own], [self.q[c][0].data_ptr() for c in own], [self.q[c][1].data_ptr() for c in own],
                                          [self.q[c][0].numel() for c in own], [self.counts[c].data_ptr() for c in own],
                                          [row[chroms.index(c):].data_ptr() for c in own], stream)

    def timed(sh, k_steps, k_warm, collective):
        rows = torch.zeros((k_steps + k_warm + 1, len(chroms)), dtype=torch.int64, device="cuda")
        for k in range(k_warm):
            sh.step(rows[k])
            if collective:
                all_reduce(dist, rows[k])
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k_steps)]
        if collective:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pending = []
        for k in range(k_steps):
            ev[k][0].record()
            sh.step(rows[k_warm + k])
            ev[k][1].record()
            if collective:
                # RCCL over xGMI: 24 x int64, the path's only collective.  Asynchronous: step k + 1 does not need step k's
                # reduced totals, so its kernels may run while the 192 bytes travel; every reduction is waited for inside
                # the timed region.
                pending.append(all_reduce(dist, rows[k_warm + k], async_op=True))
        for w in pending:
            w.wait()
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        return elapsed, kernel_ms, rows[k_warm:k_warm + k_steps]

    t0 = time.perf_counter()
    mine = Shard(assign[rank])
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    collective = world > 1
    elapsed, kernel_ms, rows = timed(mine, steps, warmup, collective)
    if collective:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda")
        all_reduce(dist, t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0].item()), float(t[1].item())
    # parity on what was just measured: owner-side golden hashes, and the reduced totals against the owners' counts
    ok_hash, ok_sum = [], []
    own_tot = torch.zeros(len(chroms), dtype=torch.int64, device="cuda")
    for c in mine.owned:
        cn = mine.counts[c].cpu().numpy()
        g = genome_golden_check(c, cn, golden)
        if g is not None:
            ok_hash.append(bool(g))
        own_tot[chroms.index(c)] = int(cn.sum(dtype=np.int64))
    if collective:
        all_reduce(dist, own_tot)
    same_every_step = bool((rows == own_tot.unsqueeze(0)).all().item())
    flags = torch.tensor([int(all(ok_hash)) if ok_hash else 1, int(same_every_step), len(ok_hash)], dtype=torch.int64, device="cuda")
    if collective:
        red = flags.clone()
        all_reduce(dist, red, op=dist.ReduceOp.MIN)
        cnt = flags[2:].clone()
        all_reduce(dist, cnt)
        flags = torch.tensor([int(red[0]), int(red[1]), int(cnt[0])], device="cuda")
    total_q = sum(qsz.values())
    value = total_q * steps / elapsed / 1e6
    out = None
    if rank == 0:
        loads = [sum(weights[c] for c in part) for part in assign]
        out = dict(
            workload="configs[3]: 24 hg19 chromosomes, synth.cfg4 (%d targets x %d queries spread by chromosome length, len U[1,1000], seeds (401,i)/(402,i)); "
                     "one index per chromosome, chromosomes dealt to the ranks by LPT, per-chromosome int64 totals all-reduced" % (sum(tsz.values()), total_q),
            value=round(value, 2), unit="M queries/s", n_gpus=world, scaling="strong", ms_per_step=round(elapsed / steps * 1e3, 4),
            kernel_ms_slowest_rank=round(kernel_ms, 4),
            roofline_frac=round(alg_bytes_of(total_q, sum(tsz.values())) / (kernel_ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
            lpt=dict(balance_max_over_mean=round(shard.balance(weights, assign), 4), speedup_bound=round(world / shard.balance(weights, assign), 3),
                     heaviest_rank_intervals=max(loads), chromosomes_per_rank=[len(p) for p in assign]),
            collective=("all_reduce(int64[24]) per step, backend %s" % dist.get_backend()) if collective else "none (one rank)",
            parity=dict(golden_subsample_hashes_ok=bool(flags[0].item()), chromosomes_hashed=int(flags[2].item()),
                        reduced_totals_equal_sum_of_owner_counts_every_step=bool(flags[1].item()),
                        overlaps_per_step=int(own_tot.sum().item())),
            build_s=round(mine.index_s, 4), data_generation_and_build_s=round(build_s, 2))
    if not collective and rank == 0 and not os.environ.get("BENCH_PER_CHROM"):
        # side legs on one GPU (never `value`): (1) the caller wants the per-chromosome TOTALS only (configs[3]'s "all-reduce on counts"
        # shape, scripts/bed_count_overlapping.py consumes len(find()) only): no count is stored per query; (2) every chromosome's
        # queries sorted by start, as a sorted BED file brings them: the walk on cell images answers them as they lie
        # (count_dense.hpp: bs_check_multi / bs_plan_multi / bs_walk over the segments), no exchange.
        class TotalsOnly:
            owned = mine.owned

            def step(self, row):
                own = mine.owned
                IntervalIndex.count_multi_dev([mine.ix[c] for c in own], [mine.q[c][0].data_ptr() for c in own], [mine.q[c][1].data_ptr() for c in own],
                                              [mine.q[c][0].numel() for c in own], [None for c in own], [row[chroms.index(c):].data_ptr() for c in own], stream)

        e2, k2, rows2 = timed(TotalsOnly(), steps, warmup, False)
        out["total_only"] = dict(ms_per_step=round(e2 / steps * 1e3, 4), kernel_ms=round(k2, 4),
                                 totals_equal_counts_pass=bool((rows2 == own_tot.unsqueeze(0)).all().item()))
        for c in mine.owned:
            qs_c, qe_c = mine.q[c]
            o = torch.argsort(qs_c, stable=True)
            mine.q[c] = (qs_c[o].contiguous(), qe_c[o].contiguous())
            del o
        warm_rows = torch.zeros((4, len(chroms)), dtype=torch.int64, device="cuda")
        for k in range(4):  # (the exact order check comes back one call after the probe saw no descent: let the answers arrive)
            mine.step(warm_rows[k])
            torch.cuda.synchronize()
        e3, k3, rows3 = timed(mine, steps, 1, False)
        sorted_counts_ok = True
        for c in mine.owned[:3]:  # (three chromosomes' counts against the shuffled pass, as multisets per chromosome total)
            sorted_counts_ok = sorted_counts_ok and int(mine.counts[c].sum(dtype=torch.int64).item()) == int(own_tot[chroms.index(c)].item())
        out["sorted_queries"] = dict(ms_per_step=round(e3 / steps * 1e3, 4), kernel_ms=round(k3, 4), value=round(total_q * steps / e3 / 1e6, 2),
                                     roofline_frac=round(alg_bytes_of(total_q, sum(tsz.values())) / (k3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     totals_equal_shuffled_pass=bool((rows3 == own_tot.unsqueeze(0)).all().item()) and sorted_counts_ok,
                                     note="every chromosome's queries sorted by start; one bxmi_ivl_count_multi_dev call per step, answered by the sorted walk over segments")
    # the same genome on ONE GPU, in the same run (rank 0 alone; the others wait): what the speed-up is measured against
    if collective and single_gpu_reference:
        if rank == 0:
            rest = Shard([c for c in chroms if c not in assign[0]])
            rest.owned = chroms
            for d in ("ix", "q", "counts"):
                getattr(rest, d).update(getattr(mine, d))
            e1, k1, _ = timed(rest, steps, warmup, False)
            out["single_gpu_same_run"] = dict(value=round(total_q * steps / e1 / 1e6, 2), ms_per_step=round(e1 / steps * 1e3, 4), kernel_ms=round(k1, 4))
            out["speedup_vs_1gpu"] = round(value / out["single_gpu_same_run"]["value"], 3)
        dist.barrier()
    return out


def source_stamps():
    """Hashes that tie a profile to the code it measured: sha256 of bench.py and of the kernel sources, first 16 hex digits."""
    def sha(paths):
        h = hashlib.sha256()
        for q in paths:
            h.update(open(q, "rb").read())
        return h.hexdigest()[:16]

    csrc = os.path.join(ROOT, "bx-python_amd", "csrc")
    kernels = [os.path.join(csrc, f) for f in ("common.hpp", "primitives.hpp", "count_direct.hpp", "count_bitmap.hpp", "count_slices.hpp",
                                               "count_dense.hpp", "offset_cells.hpp", "intervals.hip")]  # what the count pass is made of
    return dict(bench_sha16=sha([os.path.join(ROOT, "bench.py")]), kernel_sha16=sha(kernels))


def alg_bytes_of(nq, nt):
    """8 B in + 4 B out per query, the sorted starts + ends read once."""
    return nq * 12 + nt * 8

class _Done:
    def wait(self):
        return True


def all_reduce(dist, t, op=None, async_op=False):
    """dist.all_reduce on a device tensor.  Under BENCH_DRY_MULTI=1 the backend is gloo and every rank sits on one GPU: the
    tensor goes through host memory (the rehearsal must not depend on gloo having been built with device support)."""
    kw = {} if op is None else {"op": op}
    if os.environ.get("BENCH_DRY_MULTI") == "1":
        h = t.detach().cpu()
        dist.all_reduce(h, **kw)
        t.copy_(h)
        return _Done() if async_op else None
    return dist.all_reduce(t, async_op=async_op, **kw)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher's environment: start the N ranks here, one per device, the way the
    contract's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` would, and
    return its exit code.  Fewer than N devices is an error (exit 2), never a silent one-rank run; BENCH_DRY_MULTI=1
    (every rank on cuda:0, gloo) is the rehearsal of this path on a one-GPU box."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if os.environ.get("BENCH_DRY_MULTI") == "1":
        if have < 1:
            log("bench.py: BENCH_DRY_MULTI=1 still needs one HIP device; none visible")
            return 2
    elif have < n:
        log("bench.py: --gpus %d asked for, %d HIP device(s) visible: refusing to run fewer ranks than asked "
            "(BENCH_DRY_MULTI=1 rehearses the N-rank code path on one device)" % (n, have))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: launching %d ranks: %s" % (n, " ".join(cmd)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--queries", type=int, default=100_000_000)
    ap.add_argument("--targets", type=int, default=10_000_000)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bitset", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-pointer (PCIe-inclusive) side measurement: its chunked passes would mix into a kernel profile")
    ap.add_argument("--no-find", action="store_true", help="skip the configs[4] CSR-join side measurement")
    ap.add_argument("--no-sorted", action="store_true", help="skip the sorted-queries side measurement (profiling runs: its launches "
                    "dismiss the bucketed kernels at once and would halve their average durations)")
    ap.add_argument("--allreduce-total", type=int, default=1, help="all-reduce the int64 overlap total each step when --gpus > 1")
    ap.add_argument("--workload", choices=["auto", "count", "genome"], default="auto",
                    help="count (= auto) = configs[1] at EVERY N: each rank its own 100M queries against a replica of the 10M-target index, weak "
                         "scaling, the same metric and per-GPU work whatever --gpus is; the genome (configs[3], 24 chromosomes dealt to the ranks "
                         "by LPT, strong scaling) rides the same line as the `genome` object.  genome = that leg alone as the top-level line.")
    ap.add_argument("--weak", action="store_true", help="(accepted for compatibility: --workload count is weak scaling at every N)")
    ap.add_argument("--no-genome", action="store_true", help="skip the configs[3] leg")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d: refusing to print a line for a job of another size"
                         % (args.gpus, world))

    import torch
    import torch.distributed as dist

    from bxmi import _ffi, synth
    from bxmi.intervals import IntervalIndex

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # BENCH_DRY_MULTI=1: rehearsal of the N > 1 code path on a box with ONE GPU (all ranks on cuda:0, gloo instead of
    # RCCL, which refuses two ranks on one device).  The line it prints says so; it is not a scaling measurement.
    dry = os.environ.get("BENCH_DRY_MULTI") == "1"
    if dry:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    _ffi.call("bxmi_set_device", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    # the world the collective library itself reports: an all-reduce of ones over the job's communicator
    coll = dict(world=1, backend="none (one rank)")
    if world > 1:
        ones = torch.ones(1, dtype=torch.int64, device="cuda")
        all_reduce(dist, ones)
        coll = dict(world=dist.get_world_size(), backend=dist.get_backend(), ranks_counted_by_all_reduce_of_ones=int(ones.item()),
                    devices_visible=torch.cuda.device_count(), device_of_rank0=local_rank)
        if coll["ranks_counted_by_all_reduce_of_ones"] != args.gpus:
            raise SystemExit("bench.py: the collective counted %d ranks, --gpus %d" % (coll["ranks_counted_by_all_reduce_of_ones"], args.gpus))

    # The top-level line is configs[1] at every N (a 1 -> 8 curve must compare one workload with itself).
    workload = "count" if args.workload == "auto" else args.workload
    if workload == "genome":
        g = bench_genome(torch, dist, rank, world, args.steps, args.warmup, args.targets, args.queries)
        if rank == 0:
            name = _ffi.C.create_string_buffer(128)
            _ffi.call("bxmi_device_info", local_rank, name, 128, None, None)
            total_q = sum(synth.cfg4_sizes(args.queries).values())
            alg = alg_bytes_of(total_q, sum(synth.cfg4_sizes(args.targets).values()))
            achieved = alg / (g["kernel_ms_slowest_rank"] * 1e-3) / 1e9
            line = {
                "metric": "M overlap-queries/s, whole-genome intersect: 24 chromosomes, 100M x 10M intervals (count-only)",
                "value": g["value"], "unit": "M queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": g["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32",
                "data": "synthetic",
                "config": {"workload": g["workload"], "sharding": "chromosomes dealt to the ranks by LPT (largest first, to the least loaded rank); "
                           "no data-path collective, per-chromosome int64 totals all-reduced (RCCL) every step", "lpt": g["lpt"]},
                "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                             "frac": round(achieved / (HBM_PEAK_GBS * world), 4), "traffic": None,
                             "kernel": "all of a rank's chromosomes in one bxmi_ivl_count_multi_dev pass (tile sort, run table, plan, the persistent walk on "
                                       "offset-cell images -- a chromosome has one target per ~300 coordinates -- un-permute); slowest rank", "kernel_ms": g["kernel_ms_slowest_rank"],
                             "algorithmic_bytes_per_launch": alg, "peak_note": "n_gpus x 8 TB/s",
                             "timed_with": "HIP events on the launch stream around every rank's chromosomes; the slowest rank's mean"},
                "collective": dict(coll, per_step=g["collective"]), "parity": g["parity"], "index_build_s": g["build_s"], "device": name.value.decode(),
            }
            for k in ("single_gpu_same_run", "speedup_vs_1gpu"):
                if k in g:
                    line[k] = g[k]
            if dry:
                line["dry_run"] = "BENCH_DRY_MULTI=1: every rank on cuda:0, gloo instead of RCCL -- a rehearsal of the code path, not a measurement"
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    (ts, te), _ = synth.cfg2(args.targets, 1)
    baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        baseline = CpuBaseline(ts, te)
        baseline.start()

    # ---- resident data -------------------------------------------------------
    t0 = time.perf_counter()
    ix = IntervalIndex()
    ix.append(ts, te)
    ix.seal()
    torch.cuda.synchronize()
    first_build_s = time.perf_counter() - t0  # the process's first index: library load, code objects, nothing warm
    # a second index over the same targets, timed in its parts: the 80 MB host copy + upload, and the seal itself (two
    # radix sorts, prefix max, three search trees)
    t0 = time.perf_counter()
    ix2 = IntervalIndex()
    ix2.append(ts, te)
    t1 = time.perf_counter()
    ix2.seal()
    _ffi.call("bxmi_synchronize", None)
    build_s = time.perf_counter() - t0
    build_parts = dict(first_index_of_the_process_s=round(first_build_s, 4), append_from_host_arrays_s=round(t1 - t0, 4),
                       seal_s=round(build_s - (t1 - t0), 4))
    ix2.close()
    qs_h, qe_h = synth.uniform_intervals(args.queries, 202 + 1000 * rank)
    qs = torch.from_numpy(qs_h).cuda()
    qe = torch.from_numpy(qe_h).cuda()
    counts = torch.empty(args.queries, dtype=torch.int32, device="cuda")
    # bxmi_ivl_count_dev ADDS the pass's overlap total to *total: every step gets its own zeroed slot
    totals = torch.zeros(args.steps + args.warmup + 8, dtype=torch.int64, device="cuda")
    total = totals[-1:]
    stream = torch.cuda.current_stream().cuda_stream
    nq = args.queries
    step_no = [0]

    def step(ev=None):
        slot = totals[step_no[0]:step_no[0] + 1]
        step_no[0] += 1
        if ev:
            ev[0].record()
        ix.count_dev(qs.data_ptr(), qe.data_ptr(), nq, counts.data_ptr(), slot.data_ptr(), stream)
        if ev:
            ev[1].record()
        if world > 1 and args.allreduce_total:
            all_reduce(dist, slot)  # RCCL over xGMI: 8 bytes, the path's only collective

    for _ in range(args.warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        all_reduce(dist, t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))

    # ---- self-check of what was just measured --------------------------------
    total.zero_()
    ix.count_dev(qs.data_ptr(), qe.data_ptr(), nq, counts.data_ptr(), total.data_ptr(), stream)
    torch.cuda.synchronize()
    local_total = int(total.item())
    sum_ok = local_total == int(counts.sum(dtype=torch.int64).item())
    parity = "sum-of-counts == total: %s" % sum_ok
    if world == 1:
        parity += "; every timed step produced that total: %s" % bool((totals[: args.steps + args.warmup] == local_total).all().item())
    job_total = local_total
    if world > 1:
        # every rank's own total, summed by the collective: what each timed step's all-reduced slot must hold
        lt = torch.tensor([local_total], dtype=torch.int64, device="cuda")
        all_reduce(dist, lt)
        job_total = int(lt.item())
        if args.allreduce_total:
            parity += "; every timed step's all-reduced total == sum of the ranks' totals: %s" % bool(
                (totals[: args.steps + args.warmup] == job_total).all().item())
    golden_path = os.path.join(ROOT, "tests", "golden", "scale.json")
    if rank == 0 and args.queries == 100_000_000 and args.targets == 10_000_000 and os.path.exists(golden_path):
        pt = json.load(open(golden_path))["points"].get("10M x 1M (cfg2 subsample)")
        if pt:
            sub = counts[:: pt["stride"]].contiguous().cpu().numpy()
            ok = hashlib.sha256(sub.tobytes()).hexdigest() == pt["counts_sha256"] and int(sub.sum(dtype=np.int64)) == pt["total"]
            parity += "; sha256 of the 1M-query subsample == reference treap's: %s" % ok

    # what a caller's FIRST passes cost (the headline is a steady-state number).  A fresh handle over the same
    # targets: pass 1 builds the index's unit images, allocates the pass's scratch and asks the order probe synchronously
    # (all once per handle); pass 2 is what every later pass costs unless the batch looks sorted.
    cold = None
    if rank == 0 and world == 1:
        ixc = IntervalIndex()
        ixc.append(ts, te)
        ixc.seal()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        ixc.count_dev(qs.data_ptr(), qe.data_ptr(), nq, counts.data_ptr(), total.data_ptr(), stream)
        ev[1].record()
        ixc.count_dev(qs.data_ptr(), qe.data_ptr(), nq, counts.data_ptr(), total.data_ptr(), stream)
        ev[2].record()
        torch.cuda.synchronize()
        cold = dict(first_pass_ms=round(ev[0].elapsed_time(ev[1]), 4), second_pass_ms=round(ev[1].elapsed_time(ev[2]), 4),
                    order_check_skipped_after_first_pass=bool(ixc.order_state()[0]),
                    note="fresh handle, same 10M targets and 100M shuffled queries: pass 1 = image build (once per sealed index) + scratch "
                         "allocation + the order probe answered synchronously; pass 2 = the steady state")
        ixc.close()

    # the caller wants the TOTAL only (counts = NULL; scripts/bed_count_overlapping.py consumes len(find()) only): the same pass, its
    # un-permute kernel sums without storing a count per query.  Reported beside the headline, never `value`.
    total_only = None
    if rank == 0 and world == 1 and not args.no_sorted:  # (a side leg like `sorted_queries`: tools/profile.sh leaves both out)
        tt = torch.zeros(8, dtype=torch.int64, device="cuda")
        ix.count_dev(qs.data_ptr(), qe.data_ptr(), nq, None, tt[0:].data_ptr(), stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(5):
            ix.count_dev(qs.data_ptr(), qe.data_ptr(), nq, None, tt[1 + k:].data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        total_only = dict(ms_per_pass=round(e0.elapsed_time(e1) / 5, 4), same_total=bool((tt[:6] == local_total).all().item()))

    # the same batch through the HOST-pointer entry point (pageable numpy buffers over PCIe): reported, never `value`
    pcie = None
    if rank == 0 and world == 1 and not args.no_pcie:
        ix.count(qs_h[:1 << 20], qe_h[:1 << 20])
        t1 = time.perf_counter()
        hc, ht = ix.count(qs_h, qe_h)  # the handle's first batch of this size: its 1.2 GB of device staging are allocated here
        dt_first = time.perf_counter() - t1
        del hc
        t1 = time.perf_counter()
        hc, ht = ix.count(qs_h, qe_h)  # (a fresh numpy output array every call: its page faults are inside)
        dt = time.perf_counter() - t1
        same = bool(ht == local_total and np.array_equal(hc, counts.cpu().numpy()))
        pcie = dict(value=round(nq / dt / 1e6, 1), unit="M queries/s", seconds=round(dt, 4), first_call_seconds=round(dt_first, 4), same_counts=same,
                    note="bxmi_ivl_count on pageable numpy arrays, fresh output array: 0.8 GB H2D + 0.4 GB D2H in chunks of 8 Mi queries, "
                         "upload of chunk k+1 / pass on k / download of k-1 at once (csrc/host_pipeline.hpp: ivl_count_host_chunks)")
        del hc

    # the same queries sorted by start (how BED files usually arrive): libbxmi notices on the device and answers in one
    # pass without bucketing.  Reported beside the headline, never `value` (BASELINE.json asks for generated order).
    sorted_q = None
    if rank == 0 and world == 1 and not args.no_sorted:
        order = torch.argsort(qs, stable=True)
        sqs, sqe = qs[order].contiguous(), qe[order].contiguous()
        del order
        scounts = torch.empty_like(counts)
        # the first sorted batch after shuffled ones meets no order check: it goes through the exchange, exact as ever
        torch.cuda.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
        ix.count_dev(sqs.data_ptr(), sqe.data_ptr(), nq, scounts.data_ptr(), total.data_ptr(), stream)
        f1.record()
        torch.cuda.synchronize()
        first_sorted_ms = f0.elapsed_time(f1)
        for _ in range(2):
            ix.count_dev(sqs.data_ptr(), sqe.data_ptr(), nq, scounts.data_ptr(), total.data_ptr(), stream)
        # (the warm-up has to END before the timed passes are enqueued: after the shuffled batches above the library goes
        # without its order check, and what brings it back is the report of a finished pass -- bxmi_ivl_order_state)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(5, args.steps)
        e0.record()
        for _ in range(reps):
            ix.count_dev(sqs.data_ptr(), sqe.data_ptr(), nq, scounts.data_ptr(), total.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        s_ms = e0.elapsed_time(e1) / reps
        same = int(scounts.sum(dtype=torch.int64).item()) == local_total
        sorted_q = dict(value=round(nq / s_ms / 1e3, 1), unit="M queries/s", ms_per_pass=round(s_ms, 4),
                        frac_of_hbm_peak=round(alg_bytes_of(nq, args.targets) / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        same_total_as_unsorted=bool(same),
                        kernel=("bm_sorted_check (detects the order, leaves every unit's stretch of the query arrays) + bs_plan + bs_walk (a unit's cell image in LDS, "
                                "its queries answered as they lie: no exchange)" if ix.flat_state()[0] == 1 or ix.sparse_state()[0] == 1
                                else "bm_sorted_check (detects the order) + ivl_local_count_kernel"),
                        first_sorted_pass_after_shuffled_ms=round(first_sorted_ms, 4))
        del sqs, sqe, scounts

    stages = dict(zip(("flat_walk_on_cell_images", "dense_unit_images", "key_slices", "offset_cell_images"),
                      (ix.flat_state()[0], ix.dense_state()[0], ix.slice_state()[0], ix.sparse_state()[0])))
    # configs[3] on all the ranks of this job (strong scaling, chromosomes dealt by LPT, totals all-reduced): a collective
    # leg, so every rank walks through it; rank 0 keeps the result for the line
    genome_leg = None
    if world > 1 and not args.no_genome:
        del qs, qe, counts
        torch.cuda.empty_cache()
        # (no try / except here: a rank that fails would hang the others in the collective -- let it surface)
        genome_leg = bench_genome(torch, dist, rank, world, max(5, args.steps), args.warmup, args.targets, args.queries)
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = world * nq * args.steps / elapsed / 1e6
    alg_bytes = alg_bytes_of(nq, args.targets)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    partitioned = nq >= (2 << 20)  # libbxmi's default switch-over to the large-batch (bitmap-cell) pass
    name = _ffi.C.create_string_buffer(128)
    cus = _ffi.C.c_int(0)
    _ffi.call("bxmi_device_info", local_rank, name, 128, _ffi.C.byref(cus), None)
    line = {
        "metric": "M overlap-queries/s at 100M x 10M intervals (count-only)",
        "value": round(value, 2),
        "unit": "M queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "first_pass_ms": cold["first_pass_ms"] if cold else None,
        "second_pass_ms": cold["second_pass_ms"] if cold else None,
        "first_sorted_pass_after_shuffled_ms": sorted_q["first_sorted_pass_after_shuffled_ms"] if sorted_q else None,
        "cold_passes": cold,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {
            "workload": "configs[1]: %d queries x %d targets, single chrom, int32 SoA, count-only; G=250M, len U[1,1000], "
                        "numpy default_rng seeds 201 (targets) / 202+1000*rank (queries)" % (nq, args.targets),
            "per_gpu_queries": nq, "targets": args.targets, "index_replicated_per_gpu": True,
            "sharding": "queries split across ranks, no data-path collective" + ("; int64 total all-reduced (RCCL)" if world > 1 else ""),
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None,
            "kernel": ("count pass = bm_params (with a probe of 8192 starts for the order; the exact bm_sorted_check + the stand-down of "
                       "ivl_local_count run only while the probes see no descent) + bm_tile_sort + bd_transpose + bd_plan + "
                       "bw_search (persistent walk on cell images of 2^18-coordinate units, ring of hand-issued loads) + bd_unpermute "
                       "(8-bit counts) + bm_fold_totals; dominant by time: bm_tile_sort_kernel (HBM-bound), then bw_search_kernel" if partitioned else "ivl_count_kernel"),
            "search_stage_of_this_index": stages,
            "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": alg_bytes,
            "timed_with": "HIP events on the launch stream around every bxmi_ivl_count_dev call of the timed region",
        },
        "index_build_s": round(build_s, 4),
        "index_build_parts": build_parts,
        "pcie_inclusive": pcie,
        "sorted_queries": sorted_q,
        "total_only": total_only,
        "parity": parity,
        "overlaps_per_step_rank0": local_total,
        "overlaps_per_step_all_ranks": job_total,
        "device": name.value.decode(),
    }
    # HBM bytes per launch from the PMC counters: taken in their own rocprofv3 runs (tools/profile.sh), so this run can only
    # quote them -- and only if they were taken on THIS code (the file carries the hashes of bench.py and the kernel sources)
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    line["roofline"]["traffic_source"] = "none: profiles/pmc_latest.json missing"
    if os.path.exists(pmc):
        try:
            doc = json.load(open(pmc))
            have, want = doc.get("stamps", {}), source_stamps()
            if all(have.get(k) == want[k] for k in want):
                line["roofline"]["traffic"] = doc.get("count_pass", {}).get("hbm_bytes_per_launch")
                line["roofline"]["traffic_source"] = ("profiles/pmc_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `%s` on %s, "
                                                      "bench_sha16=%s kernel_sha16=%s head=%s" % (have.get("command"), have.get("date"), have.get("bench_sha16"),
                                                                                                  have.get("kernel_sha16"), have.get("head")))
            else:
                line["roofline"]["traffic_source"] = "stale: profiles/pmc_latest.json was taken on other code (%s), not quoted" % ", ".join(
                    "%s %s != %s" % (k, have.get(k), want[k]) for k in want if have.get(k) != want[k])
        except Exception as ex:
            line["roofline"]["traffic_source"] = "unreadable: %r" % ex
    if baseline is not None:
        res = baseline.measure(qs_h, qe_h, args.cpu_sample)
        if res is None:
            line["cpu_baseline"] = {"error": baseline.error}
        else:
            cb, ocounts = res
            cb["agrees_with_gpu"] = bool(np.array_equal(ocounts, counts[: args.cpu_sample].cpu().numpy()))
            cb["gpu_over_cpu"] = round(value / cb["value"], 1)
            # the reference itself cannot run here, so its speed on THIS box's cores is the port's speed here
            # divided by how much faster the port ran than the reference where both could be timed (oracle/gen_golden.py
            # --only calibration: same machine, same 10M-target treap, same 1M queries, one thread)
            try:
                cal = json.load(open(golden_path)).get("calibration")
                if cal:
                    cb["port_over_reference"] = cal["port_over_reference"]
                    cb["reference_equivalent_mq_per_s"] = round(cb["value"] / cal["port_over_reference"], 5)
                    cb["x_reference"] = round(value / (cb["value"] / cal["port_over_reference"]), 1)
                    cb["calibration"] = ("tests/golden/scale.json: reference IntervalTree.find %.5f Mq/s vs oracle/ivtree.c %.5f Mq/s on %s, 1 thread"
                                         % (cal["reference_mq_per_s"], cal["port_mq_per_s"], cal["cpu"]))
            except Exception as ex:
                cb["calibration_error"] = repr(ex)
            line["cpu_baseline"] = cb
    if world == 1 and not args.no_bitset:
        try:
            del qs, qe, counts
            torch.cuda.empty_cache()
            line["bitset"] = bench_bitsets(torch, max(5, args.steps), args.warmup)
        except Exception as ex:
            line["bitset"] = {"error": repr(ex)}
    if world == 1 and not args.no_find:
        try:
            torch.cuda.empty_cache()
            line["find_csr"] = bench_find(torch)
        except Exception as ex:
            line["find_csr"] = {"error": repr(ex)}
    if world == 1 and not args.no_genome:
        try:
            torch.cuda.empty_cache()
            line["genome"] = bench_genome(torch, dist, 0, 1, max(5, args.steps), args.warmup, args.targets, args.queries)
        except Exception as ex:
            line["genome"] = {"error": repr(ex)}
    if genome_leg is not None:
        line["genome"] = genome_leg
        # configs[3] is the strong-scaling leg: the same genome on rank 0 alone, in this run, is what it is measured against
        line["speedup_vs_1gpu"] = genome_leg.get("speedup_vs_1gpu")
    line["collective"] = coll
    if dry:
        line["dry_run"] = "BENCH_DRY_MULTI=1: every rank on cuda:0, gloo instead of RCCL -- a rehearsal of the code path, not a measurement"
    if world == 1 and not args.no_find:
        try:
            torch.cuda.empty_cache()
            line["clustered"] = bench_clustered(torch)
            u = line["ms_per_step"]
            line["clustered"]["generated_order"]["x_uniform_time"] = round(line["clustered"]["generated_order"]["ms"] / u, 2)
        except Exception as ex:
            line["clustered"] = {"error": repr(ex)}
    if world == 1 and not args.no_find:
        try:
            line["per_call_latency"] = bench_per_call()
        except Exception as ex:
            line["per_call_latency"] = {"error": repr(ex)}
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
