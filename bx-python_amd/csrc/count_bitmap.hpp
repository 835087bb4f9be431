// count_bitmap.hpp -- the large-batch count pass, second generation ("bm_*" kernels).
// Included by intervals.hip (needs IndexDev, global_rank_lt, count_one_global, the block scan).
//
// What it computes is unchanged: count(q) = #{start < qe} - #{end <= qs} for a proper query on an index without
// reversed targets (lib/bx/intervals/intersection.pyx:180-189 reports exactly the targets with end > qs and
// start < qe); everything else -- zero-length / reversed queries, queries off the coordinate grid, counts that do
// not fit 16 bits -- is flagged and recomputed from the sealed index by the last kernel, so the pass stays exact.
//
// Why a second path: the first one (part_* kernels) moves every query into GLOBAL bucket order and its count back,
// 47 B of HBM traffic per query against 12 algorithmic.  Here
//   1. bm_tile_sort_kernel   one workgroup orders a tile of 16384 / 32768 queries by coordinate bucket INSIDE LDS and writes it
//                            back in place of the tile: a 4-byte record per query (17-bit offset in the bucket, 15-bit
//                            length), the 16-bit slot of every query in the tile's sorted order, and the tile's 2048
//                            bucket offsets.  No global histogram, no prefix over tiles, every byte coalesced;
//   2. bm_transpose_kernel,  the (tile, bucket) run table turned bucket-major, and a plan that cuts every bucket's
//      bm_plan_kernel        tiles into work items of ~64 Ki queries (one item per bucket on uniform input);
//   3. the search            one of the stages of count_dense.hpp / count_slices.hpp: a workgroup keeps a UNIT (2^f neighbouring
//                            buckets) of the index in LDS -- cell images (a rank is ONE ds_read_b64, a mask, a popcount and an
//                            add), dense images or staged key slices -- walks the unit's runs through the tiles and leaves the
//                            counts beside the records;
//   4. bm_unpermute_kernel   per tile: counts pulled through the 16-bit slots back into query order, escapes recomputed.
// HBM bytes per query with cell images and 8-bit counts: 8 (queries) + 4 + 4 (records out and in) + 2 + 2 (slots) + 1 + 1 (counts)
// + 4 (result) = 26, plus the padding of the runs (4.4 %), the images (129 MB per pass) and the tables.
// count_slices.hpp also holds the two kernels that turn the pass into find().
#pragma once

namespace bxmi {

constexpr int BM_NB = PT_NB;             // coordinate buckets (the grid of the first-generation path: same geometry)
constexpr unsigned BM_REC_ESC = 0xFFFFFFFFu;
constexpr int BM_MIN_SHIFT = 5;          // at least one whole cell per bucket
constexpr int BM_GROUP_TILES = 64;       // tiles per plan group (granularity of work-item boundaries)
constexpr int BM_CHUNK = 65536;          // queries per search work item (soft: a single group is never split)
constexpr int BM_HARD = 127;             // duplicate descriptor value of a hard cell
constexpr int BM_LONG_CAP = 480;         // long runs a search workgroup remembers for its cooperative finish

struct BmGeom {
    int32_t cmin;    // first coordinate of bucket 0
    int32_t cmax;    // largest end of the index (no target reaches past it)
    int32_t shift;   // bucket width = 1 << shift
    int32_t nce;     // cells of the ends' image, sentinel included:   (W >> 5) + 2
    int32_t ncs;     // cells of the starts' image, sentinel included: ((W + BM_MARGIN) >> 5) + 1
    int32_t stride;  // cells per bucket image in global memory (nce + ncs rounded up to 16 bytes)
    // record format: offset in the low `rshift` bits, length above; the offset is relative to the first coordinate of
    // the record's UNIT = 2^f neighbouring buckets.  The image pass has f = 0 and rshift = 17; the slice pass
    // (count_slices.hpp) picks f per batch, rshift = max(17, shift + f), and uses nce / ncs / dshift for its directories.
    int32_t f;
    int32_t rshift;
    int32_t dshift;
};

__device__ __forceinline__ unsigned bm_len_esc(const BmGeom &g) { return (1u << (32 - g.rshift)) - 1u; }


// One batch may cover several sealed indexes at once (a genome: one index per chromosome, bxmi_ivl_count_multi_dev):
// every index with its queries is a SEGMENT.  Tiles are numbered across the whole batch, a segment owns a range of
// them that starts on a plan-group boundary, and every kernel below finds its geometry, arrays and images through the
// segment of the tile (or work item) it is looking at.  A plain bxmi_ivl_count_dev call is a batch of one segment.
struct BmSeg {
    BmGeom g;
    const int32_t *qs, *qe;   // the segment's queries
    int32_t *counts;          // and where their counts go
    int64_t nq;
    int64_t tile0, ntiles;    // first tile of the segment in the batch's numbering, tiles that hold queries
    int64_t tile_end;         // first tile of the next segment (tile0 + ntiles rounded up to a plan group)
    const unsigned char *dimages;  // dense stage: the index's unit images (count_dense.hpp)
    const unsigned char *pimages;  // flat walk on cell images: the index's unit images (count_dense.hpp, bp_*)
    const int4 *smeta;        // slice pass: ranks at every bucket boundary (SlMeta, count_slices.hpp)
    IndexDev ix;              // the sealed index (escapes, hard cells)
    const int32_t *e_sorted;
};

// The batch's parameter block is written by a kernel from its own arguments (stream-ordered, no host staging that a
// following batch could overwrite while this one is still queued): up to BM_PAR_CHUNK segments per launch.
constexpr int BM_PAR_CHUNK = 16;
struct BmSegChunk {
    BmSeg seg[BM_PAR_CHUNK];
    unsigned long long *total[BM_PAR_CHUNK];  // where each segment's overlap total is accumulated (may be NULL)
};

// The order PROBE (a workgroup of 256 threads): two stretches of 4096 consecutive starts.  A descent in them says "shuffled"
// for certain; none says "could be sorted", and the host keeps / brings back the exact check.
__device__ __forceinline__ bool bm_probe_descent(const int32_t *__restrict__ qs, int64_t nq)
{
    bool descent = false;
#pragma unroll
    for (int part = 1; part <= 2; part++) {
        const int64_t at = ((nq / 3 * part) & ~(int64_t)15) + 16 * (int64_t)threadIdx.x;
        if (threadIdx.x < 256 && at + 17 <= nq) {  // (256 threads look; a wider workgroup joins the barrier only)
            const int4 *p = reinterpret_cast<const int4 *>(qs + at);
            const int4 a = p[0], b = p[1], d = p[2], e = p[3];
            const int nxt = qs[at + 16];
            descent |= a.x > a.y || a.y > a.z || a.z > a.w || a.w > b.x || b.x > b.y || b.y > b.z || b.z > b.w || b.w > d.x || d.x > d.y ||
                       d.y > d.z || d.z > d.w || d.w > e.x || e.x > e.y || e.y > e.z || e.z > e.w || e.w > nxt;
        }
    }
    return __syncthreads_or(descent);
}

// a handle's first large batch asks the probe alone and waits for the answer (bm_count_segments)
__global__ __launch_bounds__(256) void bm_probe_kernel(const int32_t *__restrict__ qs, int64_t nq, unsigned *__restrict__ answer)
{
    const bool d = bm_probe_descent(qs, nq);
    if (threadIdx.x == 0) *answer = d ? 1u : 0u;
}

// Where the parameter block of a batch goes (device memory), and what is zeroed with it.
struct BmParOut {
    BmSeg *segs;
    unsigned long long **totals;
    unsigned short *tile_seg;
    unsigned long long *zero_u64;  // the segments' partial totals, the order flag, the item counters
    int n_zero;
    int *n_items;
    unsigned *probe;               // NULL, or where the order probe leaves "descent seen"
    unsigned long long *order_host;  // ... and, in host memory for the next calls, (order_seq << 1) | descent seen
    unsigned long long order_seq;
};

// A batch of at most BM_PAR_CHUNK segments whose first kernel is the tile sort (no order check in front of it) needs no
// parameter kernel: the tile sort takes the segments BY VALUE, every workgroup finds its own in the kernel arguments, and the
// first workgroup writes the block for the kernels behind it (they start after the tile sort has ended), zeroes what they
// accumulate into and asks the order probe -- bm_params_kernel's work, 5.7 us of launch on a 0.67 ms pass (a quarter of an
// eight-GPU share's fixed costs), on the side of one workgroup of 3052.
__device__ __forceinline__ void bm_write_params(const BmSegChunk &c, int npar, const BmParOut &po)
{
    const int T = (int)blockDim.x;
    for (int i = threadIdx.x; i < po.n_zero; i += T) po.zero_u64[i] = 0ull;
    if (threadIdx.x == 0) *po.n_items = 0;
    if (po.probe) {
        const bool d = bm_probe_descent(c.seg[0].qs, c.seg[0].nq);
        if (threadIdx.x == 0) {
            if (d) *po.probe = 1u;  // (behind the probe's barrier: after the zeroing above)
            if (po.order_host) *po.order_host = (po.order_seq << 1) | (d ? 1ull : 0ull);
        }
    }
    if ((int)threadIdx.x < npar) {
        po.segs[threadIdx.x] = c.seg[threadIdx.x];
        po.totals[threadIdx.x] = c.total[threadIdx.x];
    }
    for (int i = 0; i < npar; i++)
        for (int64_t t = c.seg[i].tile0 + threadIdx.x; t < c.seg[i].tile_end; t += T) po.tile_seg[t] = (unsigned short)i;
}

// The first launch of a batch also zeroes what the later kernels accumulate into (the segments' partial totals with the
// order flag behind them, the plan's item count): two memsets less on the stream.
__global__ __launch_bounds__(256) void bm_params_kernel(BmSegChunk c, int first, BmSeg *__restrict__ segs, unsigned long long **__restrict__ totals,
                                                        unsigned short *__restrict__ tile_seg, unsigned long long *__restrict__ zero_u64, int n_zero,
                                                        int *__restrict__ n_items, unsigned *__restrict__ probe = nullptr,
                                                        unsigned long long *__restrict__ order_host = nullptr, unsigned long long order_seq = 0)
{
    if (first == 0 && blockIdx.x == 0) {
        for (int i = threadIdx.x; i < n_zero; i += 256) zero_u64[i] = 0ull;
        if (threadIdx.x == 0) *n_items = 0;
        if (probe) {
            // No order check in this pass (bm_count_segments stopped launching it after shuffled batches): a PROBE instead.
            // *probe (zeroed above) = 1: descent seen; the same into host memory, for the next calls on the handle.
            const bool d = bm_probe_descent(c.seg[0].qs, c.seg[0].nq);
            if (threadIdx.x == 0) {
                if (d) *probe = 1u;  // (behind the probe's barrier: after the zeroing above)
                if (order_host) *order_host = (order_seq << 1) | (d ? 1ull : 0ull);
            }
        }
    }
    const BmSeg &sg = c.seg[blockIdx.x];
    const int id = first + (int)blockIdx.x;
    if (threadIdx.x == 0) {
        segs[id] = sg;
        totals[id] = c.total[blockIdx.x];
    }
    for (int64_t t = sg.tile0 + threadIdx.x; t < sg.tile_end; t += 256) tile_seg[t] = (unsigned short)id;
}

struct OpMin {
    template <typename T>
    __device__ __forceinline__ T operator()(T a, T b) const
    {
        return a < b ? a : b;
    }
};

__device__ __forceinline__ int bm_rank_lt64(const int32_t *__restrict__ a, int n, long long x)
{
    int l = 0, h = n;
    while (l < h) {
        int mid = (int)(((unsigned)l + (unsigned)h) >> 1);
        if ((long long)a[mid] < x)
            l = mid + 1;
        else
            h = mid;
    }
    return l;
}

// ---------------------------------------------------------------------------
// BED files usually arrive sorted, and a sorted batch needs no exchange at all (ivl_local_count_kernel answers it as it
// lies).  Every kernel of this pass reads the flag this one leaves: 1 = a descent was seen, go on; 0 = sorted, stand
// down.  A workgroup that sees the flag already raised leaves at once, so a shuffled batch costs a few microseconds
// (every workgroup finds a descent in its first 4096 starts) and a sorted one a single read of the starts.
// The flag is read and written with ORDINARY accesses on purpose: raising it with device-scope (write-through) stores
// from 2048 workgroups serialises them at the memory side (measured: 76 us for a shuffled batch); ordinary stores of the
// same value meet in each XCD's L2 and are written back at the kernel boundary, which is all the next kernel needs.
// BOUNDS (a sorted batch on cell images, count_dense.hpp bs_*): the same read of the starts also says where every UNIT's
// queries begin -- bounds[u] = first query whose start is not below the unit's first coordinate, u = 0 .. units (the last
// one: where the grid ends).  In a sorted batch every boundary lies between exactly one pair of neighbouring starts (or in
// front of the first / behind the last), so every entry is written exactly once; an unsorted batch leaves garbage nobody reads.
struct BmBounds {
    unsigned *bounds;  // NULL: not asked for
    int cmin, ulog, units;
};
// boundaries at or below x: u = 0 .. result - 1
__device__ __forceinline__ int bm_bounds_upto(const BmBounds &B, int x)
{
    if (x < B.cmin) return 0;
    const unsigned u = (((unsigned)x - (unsigned)B.cmin) >> B.ulog) + 1u;
    return u < (unsigned)(B.units + 1) ? (int)u : B.units + 1;
}

// One chunk of 256 x 16 starts (thread t: starts [base + 16 t, base + 16 t + 16] incl. the neighbour behind them): does this thread see a descent?
// BOUNDS: also write the unit boundaries that lie between two of its starts.
constexpr int BM_CHECK_CH = 256 * 16;
template <bool BOUNDS>
__device__ __forceinline__ bool bm_check_chunk(const int32_t *__restrict__ qs, int64_t nq, int64_t c, const BmBounds &B)
{
    const int64_t base = c * BM_CHECK_CH + 16 * (int64_t)threadIdx.x;
    bool descent = false;
    if (base + 17 <= nq) {
        const int4 *p = reinterpret_cast<const int4 *>(qs + base);
        const int4 a = p[0], b = p[1], d = p[2], e = p[3];
        const int nxt = qs[base + 16];
        descent = a.x > a.y || a.y > a.z || a.z > a.w || a.w > b.x || b.x > b.y || b.y > b.z || b.z > b.w || b.w > d.x || d.x > d.y ||
                  d.y > d.z || d.z > d.w || d.w > e.x || e.x > e.y || e.y > e.z || e.z > e.w || e.w > nxt;
        if (BOUNDS && !descent && bm_bounds_upto(B, a.x) != bm_bounds_upto(B, nxt)) {  // (rare: a unit is ~100 000 queries wide)
            const int v[17] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w, e.x, e.y, e.z, e.w, nxt};
#pragma unroll
            for (int i = 0; i < 16; i++)
                for (int u = bm_bounds_upto(B, v[i]); u < bm_bounds_upto(B, v[i + 1]); u++) B.bounds[u] = (unsigned)(base + i + 1);
        }
    } else {
        for (int64_t i = base; i + 1 < nq && i < base + 16; i++) {
            descent |= qs[i] > qs[i + 1];
            if (BOUNDS && qs[i] <= qs[i + 1])
                for (int u = bm_bounds_upto(B, qs[i]); u < bm_bounds_upto(B, qs[i + 1]); u++) B.bounds[u] = (unsigned)(i + 1);
        }
    }
    return descent;
}

// in front of the first start, behind the last (one thread)
__device__ __forceinline__ void bm_bounds_outer(const int32_t *__restrict__ qs, int64_t nq, const BmBounds &B)
{
    const int first = bm_bounds_upto(B, qs[0]), last = bm_bounds_upto(B, qs[nq - 1]);
    for (int u = 0; u < first; u++) B.bounds[u] = 0u;
    for (int u = last; u <= B.units; u++) B.bounds[u] = (unsigned)nq;
}

template <bool BOUNDS>
__global__ __launch_bounds__(256) void bm_sorted_check_kernel(const int32_t *__restrict__ qs, int64_t nq, unsigned *__restrict__ unsorted, BmBounds B)
{
    if (BOUNDS && blockIdx.x == 0 && threadIdx.x == 0) bm_bounds_outer(qs, nq, B);
    for (int64_t c = blockIdx.x; c * BM_CHECK_CH < nq; c += gridDim.x) {
        if (__hip_atomic_load(unsorted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return;
        const bool descent = bm_check_chunk<BOUNDS>(qs, nq, c, B);
        if (__syncthreads_or(descent)) {
            if (threadIdx.x == 0) __hip_atomic_store(unsorted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
        }
    }
}

// ---------------------------------------------------------------------------
// pass 1: order a tile by bucket inside LDS
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned bm_bucket_of(int qs, const BmGeom &g)
{
    if (qs < g.cmin) return 0;
    const unsigned b = ((unsigned)qs - (unsigned)g.cmin) >> g.shift;
    return b < (unsigned)(BM_NB - 1) ? b : (unsigned)(BM_NB - 1);
}

// SUB = 2 (find() through the exchange, find_exchange.hpp): the tile is ordered by HALF buckets -- 2 * BM_NB keys, the same
// grid one bit finer -- so that a (tile, bucket) run is itself ordered by half and the fill half can serve every half
// bucket from its own LDS window of (end, index) pairs.  Needs g.shift >= 1.
template <int SUB>
__device__ __forceinline__ unsigned bm_sort_key_of(int qs, const BmGeom &g)
{
    if (SUB == 1) return bm_bucket_of(qs, g);
    if (qs < g.cmin) return 0;
    const unsigned b = ((unsigned)qs - (unsigned)g.cmin) >> (g.shift - 1);
    return b < (unsigned)(2 * BM_NB - 1) ? b : (unsigned)(2 * BM_NB - 1);
}

// The record of a query the search kernel can answer from its bucket's image; anything else becomes an escape record.
__device__ __forceinline__ unsigned bm_record_of(int qs, int qe, const BmGeom &g)
{
    const unsigned rel = (unsigned)qs - (unsigned)g.cmin;
    const unsigned len = (unsigned)qe - (unsigned)qs;
    const bool ok = qs >= g.cmin && (rel >> g.shift) < (unsigned)BM_NB && qe > qs && len < bm_len_esc(g);
    return ok ? ((rel & ((1u << (g.shift + g.f)) - 1u)) | (len << g.rshift)) : BM_REC_ESC;
}

// PAD (the flat walk of count_dense.hpp): the records of a UNIT (2^f buckets, f from the segment's geometry; the host asks
// for PAD only when a unit is at least a thread's BPT buckets) start on a multiple of four slots -- up to three escape
// records fill the gap behind every unit -- so that every 16-byte slot of the tile-sorted array belongs to ONE unit: the
// search stores whole slots and knows how many memory operations it has in flight.  A tile then takes up to
// BM_PAD_ROOM more slots: its stride in the record / count arrays is TILE + BM_PAD_ROOM, its used length goes to `tend`.
// (4096 slots of room + 544: a tile's stride is then 149 632 bytes = 18 x 8 KiB + 17 lines -- with a stride of whole 8 KiB the runs
// of one unit in consecutive tiles, which is what a search workgroup reads, fall into the same few sets of the CU's L1:
// measured 238 -> 226 us for the search kernel)
constexpr int BM_PAD_ROOM = 4096 + 544;
// diagnostics (compile time, wrong results): price the phases of the tile sort -- 1 = no LDS atomics, 2 = no copy of the
// sorted tile to HBM, 3 = no placement (LDS scatter + slot stores)
#ifndef BM_TS_EXP
#define BM_TS_EXP 0
#endif

// TOT (a batch that wants its overlap TOTAL only, on cell images: bm_count_segments): nobody will put counts back into query
// order, so the slots are not written (2 of the kernel's 14 bytes per query), and `tesc[tile]` says whether the tile holds a
// query the images cannot answer (an escape record: bm_escape_totals_kernel then reads that tile's queries again).
template <int THREADS, int ITEMS, bool PAD = false, int SUB = 1, bool TOT = false>
__global__ __launch_bounds__(THREADS) void bm_tile_sort_kernel(const BmSeg *__restrict__ segs, const unsigned short *__restrict__ tile_seg,
                                                               unsigned *__restrict__ recs /* [ntiles][TILE (+ BM_PAD_ROOM)], tile-sorted */,
                                                               unsigned short *__restrict__ slots /* [nq] slot of every query in its tile */,
                                                               unsigned short *__restrict__ tbl /* [ntiles][BM_NB] first slot of every bucket */,
                                                               const unsigned *__restrict__ gate /* NULL, or 0 = sorted batch: stand down */,
                                                               unsigned *__restrict__ tend /* PAD: [ntiles] slots used */,
                                                               unsigned short *__restrict__ tbl2 /* SUB = 2: [ntiles][2 * BM_NB] first slot of every half bucket */,
                                                               const BmSegChunk par, const int npar /* 0: segs / tile_seg are in memory already */, const BmParOut po,
                                                               unsigned *__restrict__ tesc = nullptr /* TOT: [ntiles] 1 = the tile holds an escape record */)
{
    constexpr int TILE = THREADS * ITEMS;
    if (gate && *gate == 0) return;
    static_assert(!TOT || (PAD && SUB == 1), "total-only batches: the persistent walk on padded runs");
    constexpr int NBK = BM_NB * SUB;     // sort keys: buckets, or half buckets
    constexpr int BPT = NBK / THREADS;   // keys per thread in the scan
    static_assert(NBK % THREADS == 0 && (BPT == 2 || BPT == 4), "2 or 4 buckets per thread");
    static_assert(SUB == 1 || (SUB == 2 && !PAD && BPT == 4), "half buckets: packed runs, 1024-thread shapes");
    constexpr int STAGED = PAD ? TILE + 3 * THREADS : TILE;  // (at most one unit per thread: three pad slots each)
    constexpr int STRIDE = PAD ? TILE + BM_PAD_ROOM : TILE;
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    unsigned *staged = reinterpret_cast<unsigned *>(dyn);                          // [STAGED] records in sorted order
    unsigned *cnt = staged + STAGED;                                               // [NBK]
    unsigned short *toff = reinterpret_cast<unsigned short *>(cnt + NBK);          // [NBK]
    unsigned *scan_tmp = reinterpret_cast<unsigned *>(toff + NBK);                 // [16]
    const int64_t tile = blockIdx.x;
    BmGeom g;
    const int32_t BX_GLOBAL *__restrict__ qs, *__restrict__ qe;  // this tile's queries (as_global: common.hpp)
    int64_t nq, ltile, ntiles_seg;
    if (npar) {
        if (blockIdx.x == 0) bm_write_params(par, npar, po);  // (nothing below depends on it: later kernels read the block)
        int i = 0;
        while (i + 1 < npar && tile >= par.seg[i].tile_end) i++;
        g = par.seg[i].g;
        ltile = tile - par.seg[i].tile0, ntiles_seg = par.seg[i].ntiles;
        qs = as_global(par.seg[i].qs) + ltile * TILE, qe = as_global(par.seg[i].qe) + ltile * TILE;
        nq = par.seg[i].nq - ltile * TILE;
    } else {
        const BmSeg &sg = segs[tile_seg[tile]];
        g = sg.g;
        ltile = tile - sg.tile0, ntiles_seg = sg.ntiles;
        qs = as_global(sg.qs) + ltile * TILE, qe = as_global(sg.qe) + ltile * TILE;
        nq = sg.nq - ltile * TILE;
    }
    if (ltile >= ntiles_seg) return;  // padding up to the next plan group
    recs += tile * STRIDE, slots += tile * TILE;  // scratch is laid out by the batch's tile numbering
    const int64_t base = 0;
    const int n = (int)(nq < TILE ? nq : TILE);
    int n_out = n;  // slots of the sorted tile (PAD: the units' gaps included)
    for (int i = threadIdx.x; i < NBK; i += THREADS) cnt[i] = 0;
    __syncthreads();
    unsigned br[ITEMS];  // bucket << 16 | rank inside the (tile, bucket) run
    int4 vs[ITEMS / 4], ve[ITEMS / 4];
    if (n == TILE) {
#pragma unroll
        for (int j = 0; j < ITEMS / 4; j++) vs[j] = load_int4(qs + base + 4 * (j * THREADS + (int)threadIdx.x));
#pragma unroll
        for (int j = 0; j < ITEMS / 4; j++) ve[j] = load_int4(qe + base + 4 * (j * THREADS + (int)threadIdx.x));
#pragma unroll
        for (int j = 0; j < ITEMS / 4; j++) {
            const unsigned bx = bm_sort_key_of<SUB>(vs[j].x, g), by = bm_sort_key_of<SUB>(vs[j].y, g), bz = bm_sort_key_of<SUB>(vs[j].z, g),
                           bw = bm_sort_key_of<SUB>(vs[j].w, g);
            // a sorted batch puts the wave's 256 consecutive queries in one bucket: one lane adds for all of them
            // (256 same-address LDS atomics serialise otherwise)
            const unsigned b0 = (unsigned)__builtin_amdgcn_readfirstlane((int)bx);
            if (__all(bx == b0 && by == b0 && bz == b0 && bw == b0)) {
                unsigned r0 = 0;
                if (lane_id() == 0) r0 = atomicAdd(&cnt[b0], 256u);
                r0 = (unsigned)__builtin_amdgcn_readfirstlane((int)r0) + 4u * (unsigned)lane_id();
                br[4 * j + 0] = (b0 << 16) | (r0 + 0);
                br[4 * j + 1] = (b0 << 16) | (r0 + 1);
                br[4 * j + 2] = (b0 << 16) | (r0 + 2);
                br[4 * j + 3] = (b0 << 16) | (r0 + 3);
            } else {
                if (BM_TS_EXP == 1) {
                    const unsigned fake = 4u * (unsigned)(j * THREADS + (int)threadIdx.x);
                    br[4 * j + 0] = (bx << 16) | (fake + 0u), br[4 * j + 1] = (by << 16) | (fake + 1u);
                    br[4 * j + 2] = (bz << 16) | (fake + 2u), br[4 * j + 3] = (bw << 16) | (fake + 3u);
                } else {
                br[4 * j + 0] = (bx << 16) | atomicAdd(&cnt[bx], 1u);
                br[4 * j + 1] = (by << 16) | atomicAdd(&cnt[by], 1u);
                br[4 * j + 2] = (bz << 16) | atomicAdd(&cnt[bz], 1u);
                br[4 * j + 3] = (bw << 16) | atomicAdd(&cnt[bw], 1u);
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const int k = j * THREADS + threadIdx.x;
            if (k < n) {
                const unsigned b = bm_sort_key_of<SUB>(qs[base + k], g);
                br[j] = (b << 16) | atomicAdd(&cnt[b], 1u);
            }
        }
    }
    __syncthreads();
    {
        unsigned c[BPT], sum = 0;
#pragma unroll
        for (int u = 0; u < BPT; u++) {
            c[u] = cnt[BPT * threadIdx.x + u];
            sum += c[u];
        }
        unsigned tot;
        unsigned exc;
        if (PAD) {
            const int lane = lane_id();
            const int lanes = (1 << g.f) / BPT;  // lanes per unit: a power of two, 1 .. 32
            unsigned usum = sum;                 // the unit's queries, in every lane of the unit
            for (int d = 1; d < lanes; d <<= 1) usum += (unsigned)__shfl_xor((int)usum, d, 64);
            const unsigned padded = (usum + 3u) & ~3u;
            const bool leader = (lane & (lanes - 1)) == 0;
            const unsigned before = block_exclusive_scan(leader ? padded : 0u, OpSum(), 0u, scan_tmp, &tot);
            const unsigned ubase = (unsigned)__shfl((int)before, lane & ~(lanes - 1), 64);  // the unit's first slot: its leader's prefix
            unsigned inc = sum;  // inclusive prefix of the lanes' sums inside the unit
            for (int d = 1; d < lanes; d <<= 1) {
                const unsigned up = (unsigned)__shfl_up((int)inc, d, 64);
                if ((lane & (lanes - 1)) >= d) inc += up;
            }
            exc = ubase + inc - sum;
            if (leader)
                for (unsigned k = usum; k < padded; k++) staged[ubase + k] = BM_REC_ESC;  // (nobody's slot: answered, never read)
            n_out = BM_TS_EXP == 1 ? n : (int)tot;
            if (threadIdx.x == 0) tend[tile] = tot;
        } else {
            exc = block_exclusive_scan(sum, OpSum(), 0u, scan_tmp, &tot);
        }
        unsigned short o[BPT];
#pragma unroll
        for (int u = 0; u < BPT; u++) {
            o[u] = (unsigned short)exc;
            toff[BPT * threadIdx.x + u] = o[u];
            exc += c[u];
        }
        if (SUB == 2) {
            // the thread's four half buckets are two whole buckets: the bucket table as every other kernel reads it, and the finer one
            *reinterpret_cast<unsigned *>(tbl + tile * BM_NB + 2 * threadIdx.x) = (unsigned)o[0] | ((unsigned)o[2 % BPT] << 16);
            *reinterpret_cast<uint2 *>(tbl2 + tile * NBK + BPT * threadIdx.x) =
                make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2 % BPT] | ((unsigned)o[BPT - 1] << 16));
        } else {
        unsigned short *row = tbl + tile * BM_NB + BPT * threadIdx.x;
        if (BPT == 4)
            *reinterpret_cast<uint2 *>(row) = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[BPT - 1] << 16));
        else
            *reinterpret_cast<unsigned *>(row) = (unsigned)o[0] | ((unsigned)o[1] << 16);
        }
    }
    __syncthreads();
    bool esc = false;  // TOT: one of this thread's queries became an escape record
    if (n == TILE) {
        uint2 *l4 = reinterpret_cast<uint2 *>(slots + base);  // four 16-bit slots per 8-byte store
#pragma unroll
        for (int j = 0; j < ITEMS / 4; j++) {
            const unsigned s0 = toff[br[4 * j + 0] >> 16] + (br[4 * j + 0] & 0xffffu), s1 = toff[br[4 * j + 1] >> 16] + (br[4 * j + 1] & 0xffffu);
            const unsigned s2 = toff[br[4 * j + 2] >> 16] + (br[4 * j + 2] & 0xffffu), s3 = toff[br[4 * j + 3] >> 16] + (br[4 * j + 3] & 0xffffu);
            if (BM_TS_EXP == 3) {  // (keep the inputs alive)
                if ((s0 ^ s1 ^ s2 ^ s3 ^ (unsigned)vs[j].x ^ (unsigned)ve[j].y) == 0x9e3779b9u) staged[0] = s0;
                continue;
            }
            const unsigned r0 = bm_record_of(vs[j].x, ve[j].x, g), r1 = bm_record_of(vs[j].y, ve[j].y, g);
            const unsigned r2 = bm_record_of(vs[j].z, ve[j].z, g), r3 = bm_record_of(vs[j].w, ve[j].w, g);
            staged[s0] = r0, staged[s1] = r1, staged[s2] = r2, staged[s3] = r3;
            if (TOT)
                esc |= r0 == BM_REC_ESC || r1 == BM_REC_ESC || r2 == BM_REC_ESC || r3 == BM_REC_ESC;
            else
                l4[j * THREADS + threadIdx.x] = make_uint2(s0 | (s1 << 16), s2 | (s3 << 16));
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const int k = j * THREADS + threadIdx.x;
            if (k < n) {
                const unsigned s = toff[br[j] >> 16] + (br[j] & 0xffffu);
                const unsigned r = bm_record_of(qs[base + k], qe[base + k], g);
                staged[s] = r;
                if (TOT)
                    esc |= r == BM_REC_ESC;
                else
                    slots[base + k] = (unsigned short)s;
            }
        }
    }
    if (TOT) {
        const int any = __syncthreads_or(esc ? 1 : 0);
        if (threadIdx.x == 0) tesc[tile] = (unsigned)(any != 0);
    } else {
        __syncthreads();
    }
    int4 *out = reinterpret_cast<int4 *>(recs + base);
    const int n4 = BM_TS_EXP == 2 ? 0 : (n_out + 3) >> 2;  // (the scratch is padded to whole tiles)
    for (int i = threadIdx.x; i < n4; i += THREADS) out[i] = reinterpret_cast<const int4 *>(staged)[i];
}

// (Round 4 tried this kernel persistent and software-pipelined -- a workgroup per CU that asks for the next tile's queries
// while it places the current one, same register count: 333 us against 322.  On 128 workgroups it takes 484 us, i.e. a tile
// costs a CU 20 us alone and 28 us on a full chip: the kernel is bound by what the chip's DRAM gives this mix of five
// streams per workgroup (4.5 TB/s), not by the phases of a workgroup.  Not kept.  Non-temporal stores of the sorted tile
// and of the slots, non-temporal loads of the queries: 299-321 us against 318, the pass within 1 % either way.  Not kept.)

// ---------------------------------------------------------------------------
// pass 2: the run table bucket-major, and the work plan
// ---------------------------------------------------------------------------
// tbl[tile][bucket] (16-bit first slots) -> runT[bucket][tile] = first slot | length << 16, a 64 x 64 patch per workgroup;
// grpcnt[group][bucket] = queries of the bucket in the 64 tiles of the group.
// (NBK = 2 * BM_NB, grpcnt = NULL: the same transposition of the half-bucket table of find(), find_exchange.hpp)
template <int NBK = BM_NB>
__global__ __launch_bounds__(256) void bm_transpose_kernel(const unsigned short *__restrict__ tbl, const BmSeg *__restrict__ segs,
                                                           const unsigned short *__restrict__ tile_seg, int tile_log2,
                                                           unsigned *__restrict__ runT /* [NBK][ntp] */, int64_t ntp,
                                                           unsigned *__restrict__ grpcnt /* [ngroups][NBK], may be NULL */, const unsigned *__restrict__ gate)
{
    __shared__ unsigned short t[BM_GROUP_TILES][66];
    if (gate && *gate == 0) return;
    const int grp = blockIdx.x, b0 = blockIdx.y * 64;
    {
        const int r = threadIdx.x >> 2, q = threadIdx.x & 3;  // 4 threads per tile row, 16 buckets each
        const int64_t tile = (int64_t)grp * BM_GROUP_TILES + r;
        const BmSeg &sg = segs[tile_seg[tile]];  // (a plan group never straddles two segments)
        const bool live = tile - sg.tile0 < sg.ntiles;
        const int64_t left = sg.nq - ((tile - sg.tile0) << tile_log2);
        const unsigned ntile = !live ? 0u : (left < ((int64_t)1 << tile_log2) ? (unsigned)left : 1u << tile_log2);
        const unsigned short *row = tbl + tile * NBK + b0 + 16 * q;
        uint4 a = make_uint4(0, 0, 0, 0), c = a;
        if (live) {
            a = *reinterpret_cast<const uint4 *>(row);
            c = *reinterpret_cast<const uint4 *>(row + 8);
        }
        const unsigned w[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            t[r][16 * q + 2 * i] = (unsigned short)(w[i] & 0xffffu);
            t[r][16 * q + 2 * i + 1] = (unsigned short)(w[i] >> 16);
        }
        // (a full tile's total is 1 << 16 when the tile has 65536 queries: lengths are taken modulo 2^16 below, see host)
        if (q == 3) t[r][64] = (unsigned short)(b0 + 64 < NBK ? (live ? row[16] : 0) : ntile);
    }
    __syncthreads();
    {
        const int c = threadIdx.x >> 2, q = threadIdx.x & 3;  // 4 threads per bucket row, 16 tiles each
        unsigned *dst = runT + (int64_t)(b0 + c) * ntp + (int64_t)grp * BM_GROUP_TILES + 16 * q;
        unsigned sum = 0;
        unsigned v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const unsigned off = t[16 * q + i][c], len = (unsigned)(unsigned short)(t[16 * q + i][c + 1] - t[16 * q + i][c]);
            v[i] = off | (len << 16);
            sum += len;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) reinterpret_cast<uint4 *>(dst)[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        if (q == 0 && grpcnt) grpcnt[(int64_t)grp * NBK + b0 + c] = sum;
    }
}

// Work items: consecutive tile groups of one bucket, closed before they would pass BM_CHUNK queries.
// items[i] = {bucket, first tile, last tile + 1, queries}; n_items[0] = number of items.
// One workgroup, a thread per pair of buckets; the group counts are pulled 16 at a time with independent loads (one
// thread's walk down its two columns is otherwise a chain of ~2 x ngroups dependent L2 round trips: measured 62 us).
constexpr int BM_PLAN_BATCH = 16;
// UNITS = what one work item searches: 2 = a thread walks two buckets, each its own unit (one bucket per search
// workgroup); 1 = the two buckets form ONE unit (the search workgroup holds both images, see bm_search_kernel<PAIR>).
// An item never crosses from one segment (index) into the next.  items[i] = {bucket | segment << 16, first tile,
// last tile + 1, queries}.
template <bool EMIT, int UNITS>
__device__ __forceinline__ void bm_plan_walk(const unsigned *__restrict__ grpcnt, int ngroups, const BmSeg *__restrict__ segs,
                                             const unsigned short *__restrict__ tile_seg, int b0, int chunk, int (&cnt)[2], int4 *__restrict__ items,
                                             int out0, int out1)
{
    unsigned acc[2] = {0, 0};
    int g_first[2] = {0, 0};
    int out[2] = {out0, out1};
    int seg = tile_seg[0];
    int64_t t_last = EMIT ? segs[seg].tile0 + segs[seg].ntiles : 0;  // of the current segment (kept in registers: a load per item otherwise)
    cnt[0] = cnt[1] = 0;
    auto close = [&](int u, int g_end) {  // the item of unit u that ends before group g_end (same segment as `seg`)
        if (EMIT) {
            const int64_t t_end = (int64_t)g_end * BM_GROUP_TILES;
            items[out[u]++] = make_int4((b0 + u) | (seg << 16), g_first[u] * BM_GROUP_TILES, (int)(t_end < t_last ? t_end : t_last), (int)acc[u]);
        }
        cnt[u]++;
        acc[u] = 0;
    };
    for (int g0 = 0; g0 < ngroups; g0 += BM_PLAN_BATCH) {
        uint2 v[BM_PLAN_BATCH];
        int sg[BM_PLAN_BATCH];
#pragma unroll
        for (int i = 0; i < BM_PLAN_BATCH; i++) {
            const int gi = g0 + i < ngroups ? g0 + i : ngroups - 1;  // a valid address: no branch around the loads
            v[i] = *reinterpret_cast<const uint2 *>(grpcnt + (int64_t)gi * BM_NB + b0);
            sg[i] = tile_seg[(int64_t)gi * BM_GROUP_TILES];
        }
#pragma unroll
        for (int i = 0; i < BM_PLAN_BATCH; i++) {
            const int gi = g0 + i;
            if (gi >= ngroups) break;
            const unsigned cc[2] = {UNITS == 1 ? v[i].x + v[i].y : v[i].x, v[i].y};
            if (sg[i] != seg) {  // the next index starts here
#pragma unroll
                for (int u = 0; u < UNITS; u++)
                    if (acc[u] > 0) close(u, gi);
                seg = sg[i];
                if (EMIT) t_last = segs[seg].tile0 + segs[seg].ntiles;
            }
#pragma unroll
            for (int u = 0; u < UNITS; u++) {
                if (acc[u] > 0 && acc[u] + cc[u] > (unsigned)chunk) close(u, gi);
                if (acc[u] == 0) g_first[u] = gi;
                acc[u] += cc[u];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < UNITS; u++)
        if (acc[u] > 0) close(u, ngroups);
}

// BM_PLAN_BLOCKS workgroups of BM_PLAN_THREADS threads share the 1024 bucket pairs (one workgroup read the whole group
// table through one CU: 45 us for 100 M queries); each reserves room for its items with one atomic on the item count,
// which the host zeroes before the launch.  The order of the workgroups' ranges in the item list is not fixed; inside
// a range neighbouring buckets stay neighbours, which is what the search's XCD-aware item mapping wants.
constexpr int BM_PLAN_BLOCKS = 8, BM_PLAN_THREADS = BM_NB / 2 / BM_PLAN_BLOCKS;

template <int UNITS>
__global__ __launch_bounds__(BM_PLAN_THREADS) void bm_plan_kernel(const unsigned *__restrict__ grpcnt, int ngroups, const BmSeg *__restrict__ segs,
                                                                  const unsigned short *__restrict__ tile_seg, int chunk, int4 *__restrict__ items,
                                                                  int *__restrict__ n_items, const unsigned *__restrict__ gate)
{
    __shared__ int scan_tmp[16];
    __shared__ int s_base;
    if (gate && *gate == 0) return;
    const int b0 = 2 * (int)(blockIdx.x * BM_PLAN_THREADS + threadIdx.x);
    int cnt[2], again[2];
    bm_plan_walk<false, UNITS>(grpcnt, ngroups, segs, tile_seg, b0, chunk, cnt, nullptr, 0, 0);
    int tot;
    const int at = block_exclusive_scan(cnt[0] + cnt[1], OpSum(), 0, scan_tmp, &tot);
    if (threadIdx.x == 0) s_base = atomicAdd(n_items, tot);
    __syncthreads();
    const int base = s_base;
    bm_plan_walk<true, UNITS>(grpcnt, ngroups, segs, tile_seg, b0, chunk, again, items, base + at, base + at + cnt[0]);  // bucket b0's items, then bucket b0 + 1's
}

// ---------------------------------------------------------------------------
// pass 3: search
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) const unsigned long long *lds_cell_p;  // a cell: bitmap in the low word, meta in the high word

__device__ __forceinline__ int bm_hard_rank(lds_cell_p cells, unsigned rel, const int32_t *__restrict__ a, int slice_lo, long long lo)
{
    const unsigned c = rel >> 5;
    const int r0 = slice_lo + (int)((unsigned)(cells[c] >> 32) & 0xFFFFFu), r1 = slice_lo + (int)((unsigned)(cells[c + 1] >> 32) & 0xFFFFFu);
    const long long key = lo + (long long)rel;
    if (key > INT_MAX) return r1 - slice_lo;  // every int32 key is below it
    return global_rank_lt(a, r0, r1, (int)key) - slice_lo;
}

// (Round 2's search kernels on images of single buckets / bucket pairs -- bm_search_kernel, bm_search_pipe_kernel -- lived here
// until round 4: every index they served qualifies for the cell images of 2^18-coordinate units as well (count_dense.hpp,
// bp_* / bw_*: the same 8-byte cells, runs twice as long), so no input selected them any more.)
typedef int bm_v4i __attribute__((ext_vector_type(4)));

template <int U>
struct BmRound {
    unsigned first[U];  // first record of each of the group's runs
    unsigned lens[U];   // records of the item's first bucket | all records << 16
    unsigned rec[U];    // this lane's record of the first pass
    unsigned lf_at;     // this lane's record of the first leftover pass (absolute index), ~0u = none
    unsigned lf_rec;
    unsigned lf_total;  // leftover records of the whole group
    unsigned listed;    // bit u: run u is long and waits in the workgroup's list (its leftovers are not the group's)
    bool lf_second;     // that record belongs to the item's second bucket
};

// ---------------------------------------------------------------------------
// pass 4: counts back into query order
// ---------------------------------------------------------------------------
// A query the search kernel did not answer.  Proper queries entirely left of the first start or right of the last end
// need no memory access at all; the rest walks the sealed index.
__device__ __forceinline__ int bm_escape_count(const IndexDev &ix, const int32_t *__restrict__ e_sorted, const BmGeom &g, int qs, int qe)
{
    if (qs < qe && (qe <= g.cmin || qs >= g.cmax)) return 0;
    return count_one_global(ix, e_sorted, qs, qe);
}

// #{a[i] < key}, a sorted, by ALL 64 lanes of a wave together (uniform arguments): 64 probes per step narrow the range 65-fold, so
// 10 M keys are three steps and a last look at <= 64 neighbours -- four round trips to HBM where a lane's own binary search
// is twenty-four.  Two arrays of one length side by side (their loads travel together).
__device__ __forceinline__ void bm_wave_rank2_lt(const int32_t *__restrict__ a, int keyA, const int32_t *__restrict__ b, int keyB, int n, int &rankA, int &rankB)
{
    const int lane = lane_id();
    int loA = 0, hiA = n, loB = 0, hiB = n;  // everything below lo is < key, everything from hi on is >= key
    while (hiA - loA > 64 || hiB - loB > 64) {
        const int stepA = (hiA - loA + 64) / 65, stepB = (hiB - loB + 64) / 65;
        const long long pA = (long long)loA + (long long)(lane + 1) * stepA - 1, pB = (long long)loB + (long long)(lane + 1) * stepB - 1;
        const bool inA = hiA - loA > 64 && pA < hiA, inB = hiB - loB > 64 && pB < hiB;
        const int vA = inA ? a[pA] : INT_MAX, vB = inB ? b[pB] : INT_MAX;
        const int kA = __popcll(__ballot(inA && vA < keyA)), kB = __popcll(__ballot(inB && vB < keyB));  // (sorted: the first k probes)
        if (hiA - loA > 64) {
            const long long probe_k = (long long)loA + (long long)(kA + 1) * stepA - 1;  // the first probe that is not below the key, if it lies in the range
            if (probe_k < hiA) hiA = (int)probe_k;
            loA += kA * stepA;
        }
        if (hiB - loB > 64) {
            const long long probe_k = (long long)loB + (long long)(kB + 1) * stepB - 1;
            if (probe_k < hiB) hiB = (int)probe_k;
            loB += kB * stepB;
        }
    }
    const bool ltA = loA + lane < hiA && a[loA + lane] < keyA, ltB = loB + lane < hiB && b[loB + lane] < keyB;
    rankA = loA + __popcll(__ballot(ltA)), rankB = loB + __popcll(__ballot(ltB));
}

// A total-only batch on cell images (the search keeps the totals itself, nothing is put back into query order): the queries the
// images cannot answer -- improper, off the grid, longer than a record holds -- are answered here from the sealed index, tile by
// tile; a tile the tile sort found no escape record in (tesc[tile] == 0: every tile of an ordinary batch) costs one load.
// A handful of escapes (a uniform batch: the queries that start below the first target) are answered one by one by the whole
// wave (bm_wave_rank2_lt); a wave that meets many, or an improper query (a window scan), lets every lane answer its own.
// (Tried in round 6: the same work at the end of the tile sort itself, on the queries still in registers -- the registers cost the
// kernel 280 -> 378 us; out of line and re-reading the tile -- the call's spills 404 us.)
__global__ __launch_bounds__(1024) void bm_escape_totals_kernel(const BmSeg *__restrict__ segs, const unsigned short *__restrict__ tile_seg,
                                                                const unsigned *__restrict__ tesc, int64_t ntp, int tile_log2,
                                                                unsigned long long *__restrict__ total_slots /* [segments][PT_SLOTS] */,
                                                                const unsigned *__restrict__ gate)
{
    __shared__ long long red[1024 / 64];
    if (gate && *gate == 0) return;
    const int64_t tile = blockIdx.x;  // (a workgroup per tile: the rare tile with an escape is read at full width, 16 bytes per lane and load)
    const int seg = tile_seg[tile];
    const BmSeg &sg = segs[seg];
    const int64_t ltile = tile - sg.tile0;
    if (ltile >= sg.ntiles || tesc[tile] == 0u) return;  // (uniform)
    const BmGeom g = sg.g;
    const IndexDev ix = sg.ix;
    const int32_t *__restrict__ e_sorted = sg.e_sorted;
    const int64_t q0 = ltile << tile_log2;
    const int64_t left = sg.nq - q0;
    const int n = (int)(left < ((int64_t)1 << tile_log2) ? left : (int64_t)1 << tile_log2);
    long long acc = 0;
    // every lane of the wave comes here together (`live` says whether the lane holds a query): ballots and shuffles inside
    auto one = [&](bool live, int s, int e) {
        const bool esc = live && bm_record_of(s, e, g) == BM_REC_ESC;
        const unsigned long long m = __ballot(esc);
        if (m == 0ull) return;
        if (__popcll(m) > 8 || __any(esc && s >= e)) {  // many, or a zero-length / reversed query: lane by lane
            if (esc) acc += (long long)bm_escape_count(ix, e_sorted, g, s, e);
            return;
        }
        for (unsigned long long rest = m; rest; rest &= rest - 1ull) {
            const int src = __ffsll((long long)rest) - 1;
            const int S = __shfl(s, src, 64), E = __shfl(e, src, 64);
            if (E <= g.cmin || S >= g.cmax) continue;  // (bm_escape_count: nothing out there)
            int rS, rE;
            bm_wave_rank2_lt(ix.s_ord, E, e_sorted, S + 1, ix.n, rS, rE);  // (S < E: S is not INT_MAX)
            if (lane_id() == 0) acc += (long long)(rS - rE);
        }
    };
    const int n4 = n >> 2;  // (the query arrays are 16-byte aligned and a tile starts on a multiple of its size)
    const int4 *__restrict__ s4 = reinterpret_cast<const int4 *>(sg.qs + q0), *__restrict__ e4 = reinterpret_cast<const int4 *>(sg.qe + q0);
    for (int k0 = 0; k0 < n4; k0 += 8 * 1024) {  // (a tile of 32768 queries: one round, all sixteen loads of a lane in flight together)
        int4 vs[8], ve[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = k0 + u * 1024 + (int)threadIdx.x;
            vs[u] = s4[k < n4 ? k : 0], ve[u] = e4[k < n4 ? k : 0];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool live = k0 + u * 1024 + (int)threadIdx.x < n4;
            one(live, vs[u].x, ve[u].x), one(live, vs[u].y, ve[u].y), one(live, vs[u].z, ve[u].z), one(live, vs[u].w, ve[u].w);
        }
    }
    for (int k0 = 4 * n4; k0 < n; k0 += 1024) {  // (uniform trip count: the lanes past the end come along "not live")
        const int k = k0 + (int)threadIdx.x;
        const bool live = k < n;
        one(live, live ? sg.qs[q0 + k] : 0, live ? sg.qe[q0 + k] : 0);
    }
    block_accumulate_i64(acc, red, total_slots + (int64_t)seg * PT_SLOTS + (tile & (PT_SLOTS - 1)));
}

// FIND: also leave loff[tile-sorted position] = exclusive prefix of the counts inside the tile, in tile-sorted order
// (bit 31 = escape record, which contributes nothing): where the record's hits go in the tile's scratch region
// (count_slices.hpp, sl_fill_pipe_kernel).
// FIND = 2 (find_exchange.hpp): additionally, in QUERY order, svq[query] = that offset of the query's record (so that the hit
// copy streams it instead of gathering it through the slot), and the sums of the counts of every BM_PART_Q consecutive
// queries (`parts`) and of the tile (`tile_tot`): the CSR offsets are then one scan over the TILES away -- the copy kernel
// finishes them inside each part -- instead of a three-kernel scan over all queries.
// FIND = 3 (find_exchange.hpp, the fill straight into the CSR list): loff[tile-sorted position] and svq[query] are instead the
// exclusive prefix of the counts in QUERY order inside the tile -- tile base + that = the query's CSR offset, so the fill
// writes every record's hits where they belong and no copy follows.  Escape queries take part in the prefix (their hits are
// written by fx_offsets_kernel); bit 31 marks them as before.
constexpr int BM_PART_Q = 1024;
template <int THREADS, int ITEMS, int FIND = 0>
__global__ __launch_bounds__(THREADS) void bm_unpermute_kernel(const unsigned *__restrict__ cnt /* tile-sorted: the records array after the search */,
                                                               const unsigned short *__restrict__ slots, const BmSeg *__restrict__ segs,
                                                               const unsigned short *__restrict__ tile_seg,
                                                               unsigned long long *__restrict__ total_slots /* [segments][PT_SLOTS], may be NULL */,
                                                               const unsigned *__restrict__ gate, unsigned *__restrict__ loff = nullptr,
                                                               unsigned *__restrict__ svq = nullptr /* FIND 2: [ntp][TILE] */,
                                                               unsigned long long *__restrict__ parts = nullptr /* FIND 2: [ntp][TILE / BM_PART_Q] */,
                                                               unsigned long long *__restrict__ tile_tot = nullptr /* FIND 2: [ntp] */)
{
    constexpr int TILE = THREADS * ITEMS;
    constexpr int PARTS = TILE / BM_PART_Q;
    static_assert(FIND < 2 || (THREADS == 1024 && TILE % BM_PART_Q == 0), "parts of 1024 queries = 256 threads' four-query groups");
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    if (gate && *gate == 0) return;
    unsigned *vals = reinterpret_cast<unsigned *>(dyn);  // [TILE]
    __shared__ long long red[THREADS / 64];
    __shared__ unsigned long long s_part[FIND >= 2 ? PARTS : 1];
    if (FIND >= 2 && threadIdx.x < PARTS) s_part[threadIdx.x] = 0ull;
    const int64_t tile = blockIdx.x;
    const int seg_id = tile_seg[tile];
    const BmSeg &sg = segs[seg_id];
    const int64_t ltile = tile - sg.tile0;
    if (ltile >= sg.ntiles) return;  // padding up to the next plan group
    const IndexDev ix = sg.ix;
    const BmGeom g = sg.g;
    const int32_t *__restrict__ e_sorted = sg.e_sorted;
    const int32_t BX_GLOBAL *__restrict__ qs_arr = as_global(sg.qs) + ltile * TILE, *__restrict__ qe_arr = as_global(sg.qe) + ltile * TILE;  // escapes only
    int32_t BX_GLOBAL *__restrict__ out = as_global(sg.counts) + ltile * TILE;  // (as_global: common.hpp)
    cnt += tile * TILE, slots += tile * TILE;  // scratch is laid out by the batch's tile numbering
    const int64_t nq = sg.nq - ltile * TILE;
    const int64_t base = 0;
    const int n = (int)(nq < TILE ? nq : TILE);
    {
        const int4 *src = reinterpret_cast<const int4 *>(cnt + base);
        if (n == TILE) {
            int4 v[ITEMS / 4];
#pragma unroll
            for (int j = 0; j < ITEMS / 4; j++) v[j] = src[j * THREADS + threadIdx.x];
#pragma unroll
            for (int j = 0; j < ITEMS / 4; j++) reinterpret_cast<int4 *>(vals)[j * THREADS + threadIdx.x] = v[j];
        } else {
            const int n4 = (n + 3) >> 2;  // (the scratch is padded to whole tiles)
            for (int i = threadIdx.x; i < n4; i += THREADS) reinterpret_cast<int4 *>(vals)[i] = src[i];
        }
    }
    __syncthreads();
    long long acc = 0;
    if (n == TILE) {
        const uint2 *l4 = reinterpret_cast<const uint2 *>(slots + base);
        int32_t BX_GLOBAL *o4 = out + base;
        uint2 sl[ITEMS / 4];
#pragma unroll
        for (int j = 0; j < ITEMS / 4; j++) sl[j] = l4[j * THREADS + threadIdx.x];
#pragma unroll
        for (int j = 0; j < ITEMS / 4; j++) {
            unsigned c[4] = {vals[sl[j].x & 0xffffu], vals[sl[j].x >> 16], vals[sl[j].y & 0xffffu], vals[sl[j].y >> 16]};
            if (c[0] == BM_REC_ESC || c[1] == BM_REC_ESC || c[2] == BM_REC_ESC || c[3] == BM_REC_ESC) {
                const int64_t k0 = base + 4 * (int64_t)(j * THREADS + threadIdx.x);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (c[u] == BM_REC_ESC) c[u] = (unsigned)bm_escape_count(ix, e_sorted, g, qs_arr[k0 + u], qe_arr[k0 + u]);
            }
            if (sg.counts) store_int4(o4 + 4 * (j * THREADS + (int)threadIdx.x), (int)c[0], (int)c[1], (int)c[2], (int)c[3]);  // (NULL: the caller wants the total only)
            acc += (long long)c[0] + c[1] + c[2] + c[3];
            if (FIND >= 2) {  // the four queries 4 (j THREADS + t) ..: part 4 j + t / 256 -- one part per wave and j
                unsigned long long ws = (unsigned long long)c[0] + c[1] + c[2] + c[3];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) ws += __shfl_down(ws, off, 64);
                if (lane_id() == 0 && ws) atomicAdd(&s_part[j * (THREADS / 256) + (int)(threadIdx.x >> 8)], ws);
            }
        }
    } else {
        for (int k = threadIdx.x; k < n; k += THREADS) {
            unsigned c = vals[slots[base + k]];
            if (c == BM_REC_ESC) c = (unsigned)bm_escape_count(ix, e_sorted, g, qs_arr[base + k], qe_arr[base + k]);
            if (sg.counts) out[base + k] = (int)c;
            acc += c;
            if (FIND >= 2 && c) atomicAdd(&s_part[k / BM_PART_Q], (unsigned long long)c);
        }
    }
    constexpr int NW = THREADS / 64, PER_WAVE = TILE / NW, CH = FIND ? PER_WAVE / 64 : 1;
    if (FIND == 3) {
        static_assert(FIND != 3 || PER_WAVE % BM_PART_Q == 0, "a wave's share of the tile is whole parts");
        __syncthreads();  // every count picked (LDS still holds the tile-sorted counts), every part sum complete, `out` written
        const int w = threadIdx.x >> 6, lane = lane_id();
        unsigned carry = 0;  // the hits of the queries before this wave's share
        for (int i = 0; i < w * (PER_WAVE / BM_PART_Q); i++) carry += (unsigned)s_part[i];
        unsigned *sv_out = svq + tile * TILE;
        for (int c = 0; c < PER_WAVE / 64; c++) {
            const int idx = w * PER_WAVE + c * 64 + lane;
            unsigned x = 0u, slot = 0u;
            if (idx < n) x = (unsigned)out[base + idx], slot = slots[base + idx];  // (the counts this workgroup has just stored: escapes recomputed)
            const unsigned inc = wave_inclusive_scan(x, OpSum());
            if (idx < n) {
                const unsigned v = (carry + inc - x) | (vals[slot] == BM_REC_ESC ? 0x80000000u : 0u);
                sv_out[idx] = v;
                vals[slot] = v;  // (the slot is this query's alone)
            }
            carry += (unsigned)__shfl((int)inc, 63, 64);
        }
        __syncthreads();
        if (threadIdx.x < PARTS) parts[tile * PARTS + threadIdx.x] = s_part[threadIdx.x];
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (int i = 0; i < PARTS; i++) t += s_part[i];
            tile_tot[tile] = t;
        }
        int4 *lo4 = reinterpret_cast<int4 *>(loff + tile * TILE);
        const int n4 = (n + 3) >> 2;  // (the scratch is padded to whole tiles)
        for (int i = threadIdx.x; i < n4; i += THREADS) lo4[i] = reinterpret_cast<const int4 *>(vals)[i];
    } else if (FIND) {
        // (after the counts were picked: the offsets take their place in LDS below, and `e` is not live across the loop above)
        __syncthreads();
        // every wave scans its contiguous share of the tile, 64 positions at a time
        __shared__ unsigned s_wtot[NW];
        const int w = threadIdx.x >> 6, lane = lane_id();
        unsigned e[CH], carry = 0;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int idx = w * PER_WAVE + c * 64 + lane;
            const unsigned v = idx < n ? vals[idx] : 0u;
            const bool esc = v == BM_REC_ESC;
            const unsigned x = esc ? 0u : v;
            const unsigned inc = wave_inclusive_scan(x, OpSum());
            e[c] = (carry + inc - x) | (esc ? 0x80000000u : 0u);
            carry += (unsigned)__shfl((int)inc, 63, 64);
        }
        if (lane == 0) s_wtot[w] = carry;
        __syncthreads();
        unsigned wbase = 0;
        for (int i = 0; i < w; i++) wbase += s_wtot[i];
        unsigned *lo_out = loff + tile * TILE;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int idx = w * PER_WAVE + c * 64 + lane;
            const unsigned v = ((e[c] & 0x7FFFFFFFu) + wbase) | (e[c] & 0x80000000u);
            lo_out[idx] = v;
            if (FIND == 2) vals[idx] = v;  // (this lane read the count of the same position above)
        }
    }
    if (FIND == 2) {
        __syncthreads();  // the tile-sorted offsets have taken the counts' place in LDS, every part sum is complete
        if (threadIdx.x < PARTS) parts[tile * PARTS + threadIdx.x] = s_part[threadIdx.x];
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (int i = 0; i < PARTS; i++) t += s_part[i];
            tile_tot[tile] = t;
        }
        // ... and are picked through the slots like the counts were
        unsigned *sv_out = svq + tile * TILE;
        if (n == TILE) {
            const uint2 *l4 = reinterpret_cast<const uint2 *>(slots + base);  // (a second read of the tile's 64 KB of slots: L2 hits)
#pragma unroll
            for (int j = 0; j < ITEMS / 4; j++) {
                const uint2 sl = l4[j * THREADS + threadIdx.x];
                reinterpret_cast<uint4 *>(sv_out)[j * THREADS + threadIdx.x] =
                    make_uint4(vals[sl.x & 0xffffu], vals[sl.x >> 16], vals[sl.y & 0xffffu], vals[sl.y >> 16]);
            }
        } else {
            for (int k = threadIdx.x; k < n; k += THREADS) sv_out[k] = vals[slots[base + k]];
        }
    }
    if (total_slots) block_accumulate_i64(acc, red, total_slots + (int64_t)seg_id * PT_SLOTS + (blockIdx.x & (PT_SLOTS - 1)));
}

// the segments' partial totals, folded into the caller's int64 per segment (accumulated, like bxmi_ivl_count_dev's total)
__global__ void bm_fold_totals_kernel(const unsigned long long *__restrict__ slots /* [segments][PT_SLOTS] */,
                                      unsigned long long *const *__restrict__ totals /* [segments] */)
{
    unsigned long long v = threadIdx.x < PT_SLOTS ? slots[(int64_t)blockIdx.x * PT_SLOTS + threadIdx.x] : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (threadIdx.x == 0 && v && totals[blockIdx.x]) atomicAdd(totals[blockIdx.x], v);
}

}  // namespace bxmi
