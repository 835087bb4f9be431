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
    return out


def alg_bytes_of(nq, nt):
    """8 B in + 4 B out per query, the sorted starts + ends read once."""
    return nq * 12 + nt * 8


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
    ap.add_argument("--no-find", action="store_true", help="skip the configs[4] CSR-join side measurement")
    ap.add_argument("--no-sorted", action="store_true", help="skip the sorted-queries side measurement (profiling runs: its launches "
                    "dismiss the bucketed kernels at once and would halve their average durations)")
    ap.add_argument("--allreduce-total", type=int, default=1, help="all-reduce the int64 overlap total each step when --gpus > 1")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log("note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    from bxmi import _ffi, synth
    from bxmi.intervals import IntervalIndex

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    _ffi.call("bxmi_set_device", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

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
    build_s = time.perf_counter() - t0
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
            dist.all_reduce(slot)  # RCCL over xGMI: 8 bytes, the path's only collective

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
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
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
    golden_path = os.path.join(ROOT, "tests", "golden", "scale.json")
    if rank == 0 and args.queries == 100_000_000 and args.targets == 10_000_000 and os.path.exists(golden_path):
        pt = json.load(open(golden_path))["points"].get("10M x 1M (cfg2 subsample)")
        if pt:
            sub = counts[:: pt["stride"]].contiguous().cpu().numpy()
            ok = hashlib.sha256(sub.tobytes()).hexdigest() == pt["counts_sha256"] and int(sub.sum(dtype=np.int64)) == pt["total"]
            parity += "; sha256 of the 1M-query subsample == reference treap's: %s" % ok

    # the same batch through the HOST-pointer entry point (pageable numpy buffers over PCIe): reported, never `value`
    pcie = None
    if rank == 0 and world == 1:
        ix.count(qs_h[:1 << 20], qe_h[:1 << 20])
        t1 = time.perf_counter()
        hc, ht = ix.count(qs_h, qe_h)
        dt = time.perf_counter() - t1
        pcie = dict(value=round(nq / dt / 1e6, 1), unit="M queries/s", seconds=round(dt, 3), same_counts=bool(ht == local_total),
                    note="bxmi_ivl_count on host arrays: 0.8 GB H2D + 0.4 GB D2H through pageable memory included")
        del hc

    # the same queries sorted by start (how BED files usually arrive): libbxmi notices on the device and answers in one
    # pass without bucketing.  Reported beside the headline, never `value` (BASELINE.json asks for generated order).
    sorted_q = None
    if rank == 0 and world == 1 and not args.no_sorted:
        order = torch.argsort(qs, stable=True)
        sqs, sqe = qs[order].contiguous(), qe[order].contiguous()
        del order
        scounts = torch.empty_like(counts)
        for _ in range(2):
            ix.count_dev(sqs.data_ptr(), sqe.data_ptr(), nq, scounts.data_ptr(), total.data_ptr(), stream)
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
                        same_total_as_unsorted=bool(same), kernel="part_hist (detects the order) + ivl_local_count_kernel")
        del sqs, sqe, scounts

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = world * nq * args.steps / elapsed / 1e6
    alg_bytes = alg_bytes_of(nq, args.targets)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    partitioned = nq >= (4 << 20)  # libbxmi's default switch-over to the bucketed large-batch path
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
            "kernel": ("count pass = part_hist + column scan + part_scatter + part_count_cells + part_gather (dominant: part_scatter_kernel)"
                       if partitioned else "ivl_count_kernel"),
            "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": alg_bytes,
            "timed_with": "HIP events on the launch stream around every bxmi_ivl_count_dev call of the timed region",
        },
        "index_build_s": round(build_s, 3),
        "pcie_inclusive": pcie,
        "sorted_queries": sorted_q,
        "parity": parity,
        "overlaps_per_step_rank0": local_total,
        "device": name.value.decode(),
    }
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc):
        try:
            line["roofline"]["traffic"] = json.load(open(pmc)).get("count_pass", {}).get("hbm_bytes_per_launch")
        except Exception:
            pass
    if baseline is not None:
        res = baseline.measure(qs_h, qe_h, args.cpu_sample)
        if res is None:
            line["cpu_baseline"] = {"error": baseline.error}
        else:
            cb, ocounts = res
            cb["agrees_with_gpu"] = bool(np.array_equal(ocounts, counts[: args.cpu_sample].cpu().numpy()))
            cb["gpu_over_cpu"] = round(value / cb["value"], 1)
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
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
