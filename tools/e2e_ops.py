#!/usr/bin/env python3
"""End-to-end wall time of the operations layer (bxmi.operations) on a synthetic pair of interval files:
batched engine calls vs the per-row pattern of the reference's operations code running on the drop-in
BinnedBitSet (count_range + next_set/next_clear per row).  Same readers, same secondary bitsets; the outputs
are compared.  N rows per file (env N, default 200000)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import numpy as np

from bxmi import genomic, operations, synth

N = int(os.environ.get("N", 200_000))


def make(seed, tag):
    rng = np.random.default_rng(seed)
    chroms = list(synth.HG19_SIZES)
    ch = rng.integers(0, len(chroms), size=N)
    s = rng.integers(0, 40_000_000, size=N)
    e = s + rng.integers(1, 1000, size=N)
    return ["%s\t%d\t%d\t%s%d\t0\t+\n" % (chroms[c], a, b, tag, i) for i, (c, a, b) in enumerate(zip(ch.tolist(), s.tolist(), e.tolist()))]


primary, secondary = make(21, "p"), make(22, "s")


def per_row_intersect(readers, mincols=1):
    """The reference's row loop (operations/intersect.py:38-83) on the drop-in classes: three or more engine calls per row."""
    safe = genomic.BitsetSafeReaderWrapper(readers[1], lens={})
    bitsets = safe.binned_bitsets()
    for row in readers[0]:
        if not isinstance(row, genomic.GenomicInterval) or row.chrom not in bitsets:
            continue
        bits = bitsets[row.chrom]
        try:
            if bits.count_range(row.start, row.end - row.start) >= mincols:
                for a, b in operations.bits_set_in_range(bits, row.start, row.end):
                    piece = row.copy()
                    piece.start, piece.end = a, b
                    yield piece
        except IndexError:
            continue


def readers():
    return [genomic.NiceReaderWrapper(primary), genomic.GenomicIntervalReader(secondary)]


list(operations.intersect([genomic.NiceReaderWrapper(primary[:1000]), genomic.GenomicIntervalReader(secondary[:1000])]))  # warm up
for name, batched, slow in (("intersect", lambda: operations.intersect(readers()), lambda: per_row_intersect(readers())),):
    t0 = time.perf_counter()
    a = [str(x) for x in batched()]
    t1 = time.perf_counter()
    b = [str(x) for x in slow()]
    t2 = time.perf_counter()
    print("%-10s N=%d rows/file  batched %.2f s   per-row %.2f s   pieces=%d  same output: %s" % (name, N, t1 - t0, t2 - t1, len(a), a == b))
for name, fn in (("subtract", lambda: operations.subtract(readers())), ("coverage", lambda: operations.coverage(readers())),
                 ("merge", lambda: operations.merge(genomic.GenomicIntervalReader(secondary))),
                 ("complement", lambda: operations.complement(genomic.GenomicIntervalReader(secondary), dict(synth.HG19_SIZES)))):
    t0 = time.perf_counter()
    n = sum(1 for _ in fn())
    print("%-10s N=%d rows/file  batched %.2f s   items=%d" % (name, N, time.perf_counter() - t0, n))
t0 = time.perf_counter()
print("base_coverage = %d  (%.2f s)" % (operations.base_coverage(genomic.GenomicIntervalReader(secondary)), time.perf_counter() - t0))
