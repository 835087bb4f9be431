"""
Synthetic workloads of BASELINE.md / SURVEY.md §8(d).

Everything is drawn from ``numpy.random.default_rng(seed)`` (PCG64) so that the
build container (where the real reference produced the golden hashes), the CPU
tests and the GPU box all see the same arrays.
"""
import numpy as np

# test_data/epo_tests/hg19.chrom.sizes of the reference, 24 main chromosomes
# (values re-typed; sum 3 095 677 412, every chromosome < 512 Mi).
HG19_SIZES = {
    "chr1": 249250621, "chr2": 243199373, "chr3": 198022430, "chr4": 191154276,
    "chr5": 180915260, "chr6": 171115067, "chr7": 159138663, "chr8": 146364022,
    "chr9": 141213431, "chr10": 135534747, "chr11": 135006516, "chr12": 133851895,
    "chr13": 115169878, "chr14": 107349540, "chr15": 102531392, "chr16": 90354753,
    "chr17": 81195210, "chr18": 78077248, "chr19": 59128983, "chr20": 63025520,
    "chr21": 48129895, "chr22": 51304566, "chrX": 155270560, "chrY": 59373566,
}  # fmt: skip


def uniform_intervals(n, seed, genome=250_000_000, max_len=1000):
    """start ~ U[0, genome-max_len), len ~ U[1, max_len]  ->  int32 (start, end)."""
    rng = np.random.default_rng(seed)
    start = rng.integers(0, genome - max_len, size=n, dtype=np.int64)
    length = rng.integers(1, max_len + 1, size=n, dtype=np.int64)
    return start.astype(np.int32), (start + length).astype(np.int32)


def cfg1(n=10_000):
    """chr1, 10k targets (seed 101) + 10k queries (seed 102), G=250M, len U[1,1000]."""
    return uniform_intervals(n, 101), uniform_intervals(n, 102)


def cfg2(n_targets=10_000_000, n_queries=100_000_000, target_seed=201, query_seed=202):
    """One chromosome, G=250M, len U[1,1000]; targets seed 201, queries seed 202."""
    return uniform_intervals(n_targets, target_seed), uniform_intervals(n_queries, query_seed)


def cfg5(n_targets=50_000_000, n_queries=50_000_000):
    """Join workload: G=2e9, len U[1,200]  (~5 hits per query at 50M x 50M)."""
    return (
        uniform_intervals(n_targets, 501, genome=2_000_000_000, max_len=200),
        uniform_intervals(n_queries, 502, genome=2_000_000_000, max_len=200),
    )


def clustered(n_targets=10_000_000, n_queries=100_000_000, hot_spots=20_000, genome=250_000_000, seed=601, sort_queries=False):
    """Non-uniform counterpart of cfg2: everything sits around `hot_spots` places (exons, peaks).  A
    target's start is its hot spot + a small offset drawn from a geometric distribution (many targets share a start),
    its length one of a few dozen values, so coordinates are duplicated heavily; queries are drawn the same way with
    wider scatter.  -> ((target_start, target_end), (query_start, query_end)) int32, queries in generated order (or
    sorted by start, as a sorted BED file arrives)."""
    rng = np.random.default_rng(seed)
    centres = np.sort(rng.integers(10_000, genome - 10_000, size=hot_spots, dtype=np.int64))
    # some hot spots are several times busier than others
    weight = rng.lognormal(0.0, 0.6, size=hot_spots)
    weight /= weight.sum()
    lengths = rng.integers(30, 2000, size=48, dtype=np.int64)

    cum = np.cumsum(weight)

    def draw(n, scatter):
        k = np.minimum(np.searchsorted(cum, rng.random(n)), hot_spots - 1)  # (rng.choice(p=...) is ten times slower)
        start = centres[k] + rng.geometric(1.0 / scatter, size=n) - 1
        length = lengths[rng.integers(0, len(lengths), size=n)]
        return start.astype(np.int32), (start + length).astype(np.int32)

    ts, te = draw(n_targets, 60.0)
    qs, qe = draw(n_queries, 400.0)
    if sort_queries:
        o = np.argsort(qs, kind="stable")
        qs, qe = qs[o], qe[o]
    return (ts, te), (qs, qe)


def cfg4_sizes(n_total, sizes=None):
    """Intervals per chromosome for a set of n_total placed proportionally to chromosome length (rounded per chromosome)."""
    sizes = sizes or HG19_SIZES
    total = sum(sizes.values())
    return {chrom: int(round(n_total * size / total)) for chrom, size in sizes.items()}


def cfg4_chrom(chrom, n_targets=10_000_000, n_queries=100_000_000, sizes=None):
    """SURVEY 8(d) "cfg 4" (BASELINE configs[3]): the cfg-2 distributions (len U[1,1000]) on one of the 24 hg19
    chromosomes, N and Q proportional to its length; targets seed (401, i), queries seed (402, i) with i the
    chromosome's position in HG19_SIZES -- per-chromosome seeds, so a rank generates only what it owns.
    -> ((target_start, target_end), (query_start, query_end)) int32."""
    sizes = sizes or HG19_SIZES
    i = list(sizes).index(chrom)
    nt, nq = cfg4_sizes(n_targets, sizes)[chrom], cfg4_sizes(n_queries, sizes)[chrom]
    return uniform_intervals(nt, [401, i], genome=sizes[chrom]), uniform_intervals(nq, [402, i], genome=sizes[chrom])


def cfg4(n_targets=10_000_000, n_queries=100_000_000, chroms=None, sizes=None):
    """{chrom: cfg4_chrom(chrom)} for `chroms` (default: all 24)."""
    sizes = sizes or HG19_SIZES
    return {c: cfg4_chrom(c, n_targets, n_queries, sizes) for c in (chroms or list(sizes))}


def genome_ranges(n_total, seed, sizes=None, max_len=2000):
    """cfg 3/4: ranges placed proportionally to chromosome length.

    Returns {chrom: (start int32[], len int32[])}, every range inside its chromosome.
    """
    sizes = sizes or HG19_SIZES
    rng = np.random.default_rng(seed)
    total = sum(sizes.values())
    out = {}
    for chrom, size in sizes.items():
        n = int(round(n_total * size / total))
        start = rng.integers(0, size - max_len, size=n, dtype=np.int64)
        length = rng.integers(1, max_len + 1, size=n, dtype=np.int64)
        out[chrom] = (start.astype(np.int32), length.astype(np.int32))
    return out


def bed_lines(chrom, start, end, prefix="n"):
    """6-column BED text lines as cfg 1 defines them (chrom s e name 0 +)."""
    return ["%s\t%d\t%d\t%s%d\t0\t+\n" % (chrom, s, e, prefix, i) for i, (s, e) in enumerate(zip(start.tolist(), end.tolist()))]
