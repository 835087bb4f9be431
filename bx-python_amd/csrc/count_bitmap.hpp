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
//   3. bm_search_kernel      one workgroup per item.  The bucket's slice of the index sits in LDS as a BITMAP: one
//                            8-byte cell per 32 coordinates = {bitmap of occupied coordinates, rank of the cell's first
//                            key : 20, one duplicate descriptor : 12}.  A rank is ONE ds_read_b64, a mask, a popcount
//                            and an add -- no search at all.  The workgroup walks the tiles, 8 lanes per (tile, bucket)
//                            run (16 per run of a bucket PAIR: bm_search_pipe_kernel, the default), and leaves the
//                            32-bit counts where it found the records;
//   4. bm_unpermute_kernel   per tile: counts pulled through the 16-bit slots back into query order, escapes recomputed.
// HBM bytes per query: 8 (queries) + 4 + 4 (records out and in) + 2 + 2 (slots) + 4 + 4 (counts, in place) + 4 (result) = 32, plus
// the images (151 MB per pass) and the tables (75 MB).
// count_slices.hpp holds a second search stage on the same exchange (sorted key slices instead of images: sparse
// indexes, wide spans) and the two kernels that turn the pass into find().
//
// A cell holds exact multiplicities only when at most one of its 32 coordinates carries duplicates (<= 126 extra copies);
// other cells are "hard": their rank is finished by a short binary search in the sorted array between the cell's and the
// next cell's base.  bm_image_kernel counts the hard cells while it builds the images (once per sealed index) and the
// host keeps the first-generation path for indexes where they are not rare (heavily duplicated coordinates), for spans
// wider than 2^28 (a bucket's image must fit half a CU's LDS) and for indexes with reversed targets.
#pragma once

namespace bxmi {

constexpr int BM_NB = PT_NB;             // coordinate buckets (the grid of the first-generation path: same geometry)
constexpr int BM_MARGIN = 32768;         // the starts' cells reach this far past the bucket: every record's qe is covered
constexpr unsigned BM_LEN_ESC = 0x7FFFu;  // length field of an escape record (17-bit offsets: the image format)
constexpr unsigned BM_REC_ESC = 0xFFFFFFFFu;
constexpr int BM_MAX_SHIFT = 17;         // bucket width <= 131072 coordinates: (4098 + 5121) cells = 72 KiB of LDS
constexpr int BM_MIN_SHIFT = 5;          // at least one whole cell per bucket
constexpr int BM_GROUP_TILES = 64;       // tiles per plan group (granularity of work-item boundaries)
constexpr int BM_CHUNK = 65536;          // queries per search work item (soft: a single group is never split)
constexpr int BM_SEARCH_THREADS = 1024;
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

struct BmBucket {
    int32_t eLo;  // #{ends   < bucket's first coordinate}
    int32_t sLo;  // #{starts < bucket's first coordinate}
};

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
    const uint2 *images;
    const unsigned char *dimages;  // dense stage: the index's unit images (count_dense.hpp)
    const unsigned char *pimages;  // flat walk on cell images: the index's unit images (count_dense.hpp, bp_*)
    const BmBucket *bmeta;
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
        if (at + 17 <= nq) {
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

// The first launch of a batch also zeroes what the later kernels accumulate into (the segments' partial totals with the
// order flag behind them, the plan's item count): two memsets less on the stream.
__global__ __launch_bounds__(256) void bm_params_kernel(BmSegChunk c, int first, BmSeg *__restrict__ segs, unsigned long long **__restrict__ totals,
                                                        unsigned short *__restrict__ tile_seg, unsigned long long *__restrict__ zero_u64, int n_zero,
                                                        int *__restrict__ n_items, unsigned *__restrict__ probe = nullptr)
{
    if (first == 0 && blockIdx.x == 0) {
        for (int i = threadIdx.x; i < n_zero; i += 256) zero_u64[i] = 0ull;
        if (threadIdx.x == 0) *n_items = 0;
        if (probe) {
            // No order check in this pass (bm_count_segments stopped launching it after shuffled batches): a PROBE instead.
            // *probe (zeroed above) = 1: descent seen.
            if (bm_probe_descent(c.seg[0].qs, c.seg[0].nq) && threadIdx.x == 0) *probe = 1u;  // (behind its barrier: after the zeroing above)
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
// images: built once per sealed index
// ---------------------------------------------------------------------------
// One workgroup per bucket, one array at a time.  Key r of the slice sets bit (rel & 31) of cell (rel >> 5) when it is
// the first of its coordinate and bumps the cell's duplicate bookkeeping otherwise; `first` of a non-empty cell is the
// slice rank of its first key, empty cells inherit the next non-empty cell's (a suffix minimum), so that
// base[c] = #{slice keys below cell c} for every c, the sentinel included.
__global__ __launch_bounds__(1024) void bm_image_kernel(const int32_t *__restrict__ s_ord, const int32_t *__restrict__ e_sorted, int n,
                                                        BmGeom g, uint2 *__restrict__ images, BmBucket *__restrict__ bmeta,
                                                        unsigned *__restrict__ stats /* [0] hard cells, [1] slices too long */)
{
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    __shared__ int s_r[2];
    __shared__ int scan_tmp[16];
    const int b = blockIdx.x;
    const long long W = 1ll << g.shift;
    const long long lo = (long long)g.cmin + (long long)b * W;
    int r0s[2];
    for (int arr = 0; arr < 2; arr++) {
        const int32_t *__restrict__ A = arr == 0 ? e_sorted : s_ord;
        const int nc = arr == 0 ? g.nce : g.ncs;
        const long long span = arr == 0 ? W + 1 : W + BM_MARGIN;  // keys with rel in [0, span) belong to this image
        unsigned *bm = reinterpret_cast<unsigned *>(dyn);
        int *first = dyn + nc, *dcnt = dyn + 2 * nc, *dmin = dyn + 3 * nc, *dmax = dyn + 4 * nc;
        if (threadIdx.x < 2) s_r[threadIdx.x] = bm_rank_lt64(A, n, threadIdx.x == 0 ? lo : lo + span);
        __syncthreads();
        const int r0 = s_r[0], r1 = s_r[1], ns = r1 - r0;
        r0s[arr] = r0;
        for (int c = threadIdx.x; c < nc; c += 1024) {
            bm[c] = 0;
            first[c] = ns;
            dcnt[c] = 0;
            dmin[c] = 32;
            dmax[c] = -1;
        }
        __syncthreads();
        for (int r = r0 + (int)threadIdx.x; r < r1; r += 1024) {
            const int k = A[r];
            const unsigned rel = (unsigned)((long long)k - lo);
            const int c = (int)(rel >> 5), p = (int)(rel & 31);
            if (r == r0 || A[r - 1] != k) {
                atomicOr(&bm[c], 1u << p);
                atomicMin(&first[c], r - r0);
            } else {
                atomicAdd(&dcnt[c], 1);
                atomicMin(&dmin[c], p);
                atomicMax(&dmax[c], p);
            }
        }
        __syncthreads();
        // suffix minimum of `first`: thread t owns chunk 1023 - t, so an exclusive scan in thread order covers the higher chunks
        const int K = (nc + 1023) >> 10;
        const int chunk = 1023 - (int)threadIdx.x;
        const int c_lo = chunk * K, c_hi = c_lo + K < nc ? c_lo + K : nc;
        int run = INT_MAX;
        for (int c = c_hi - 1; c >= c_lo; c--) {
            run = first[c] < run ? first[c] : run;
            first[c] = run;
        }
        int tot;
        const int above = block_exclusive_scan(run, OpMin(), INT_MAX, scan_tmp, &tot);
        unsigned hard = 0;
        uint2 *out = images + (size_t)b * g.stride + (arr == 0 ? 0 : g.nce);
        for (int c = c_lo; c < c_hi; c++) {
            const int base = first[c] < above ? first[c] : above;
            unsigned meta = (unsigned)base & 0xFFFFFu;
            if (dcnt[c] > 0) {
                if (dmin[c] == dmax[c] && dcnt[c] < BM_HARD)
                    meta |= ((unsigned)dmin[c] << 20) | ((unsigned)dcnt[c] << 25);
                else {
                    meta |= (unsigned)BM_HARD << 25;
                    hard++;
                }
            }
            out[c] = make_uint2(bm[c], meta);
        }
        if (hard) atomicAdd(&stats[0], hard);
        if (threadIdx.x == 0 && ns >= (1 << 20)) atomicAdd(&stats[1], 1u);
        __syncthreads();
    }
    if (threadIdx.x == 0) bmeta[b] = BmBucket{r0s[0], r0s[1]};
}

// ---------------------------------------------------------------------------
// pass 0: is the batch already sorted by start?
// ---------------------------------------------------------------------------
// BED files usually arrive sorted, and a sorted batch needs no exchange at all (ivl_local_count_kernel answers it as it
// lies).  Every kernel of this pass reads the flag this one leaves: 1 = a descent was seen, go on; 0 = sorted, stand
// down.  A workgroup that sees the flag already raised leaves at once, so a shuffled batch costs a few microseconds
// (every workgroup finds a descent in its first 4096 starts) and a sorted one a single read of the starts.
// The flag is read and written with ORDINARY accesses on purpose: raising it with device-scope (write-through) stores
// from 2048 workgroups serialises them at the memory side (measured: 76 us for a shuffled batch); ordinary stores of the
// same value meet in each XCD's L2 and are written back at the kernel boundary, which is all the next kernel needs.
__global__ __launch_bounds__(256) void bm_sorted_check_kernel(const int32_t *__restrict__ qs, int64_t nq, unsigned *__restrict__ unsorted)
{
    constexpr int CH = 256 * 16;
    for (int64_t c = blockIdx.x; c * CH < nq; c += gridDim.x) {
        if (__hip_atomic_load(unsorted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return;
        const int64_t base = c * CH + 16 * (int64_t)threadIdx.x;
        bool descent = false;
        if (base + 17 <= nq) {
            const int4 *p = reinterpret_cast<const int4 *>(qs + base);
            const int4 a = p[0], b = p[1], d = p[2], e = p[3];
            const int nxt = qs[base + 16];
            descent = a.x > a.y || a.y > a.z || a.z > a.w || a.w > b.x || b.x > b.y || b.y > b.z || b.z > b.w || b.w > d.x || d.x > d.y ||
                      d.y > d.z || d.z > d.w || d.w > e.x || e.x > e.y || e.y > e.z || e.z > e.w || e.w > nxt;
        } else {
            for (int64_t i = base; i + 1 < nq && i < base + 16; i++) descent |= qs[i] > qs[i + 1];
        }
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

template <int THREADS, int ITEMS, bool PAD = false>
__global__ __launch_bounds__(THREADS) void bm_tile_sort_kernel(const BmSeg *__restrict__ segs, const unsigned short *__restrict__ tile_seg,
                                                               unsigned *__restrict__ recs /* [ntiles][TILE (+ BM_PAD_ROOM)], tile-sorted */,
                                                               unsigned short *__restrict__ slots /* [nq] slot of every query in its tile */,
                                                               unsigned short *__restrict__ tbl /* [ntiles][BM_NB] first slot of every bucket */,
                                                               const unsigned *__restrict__ gate /* NULL, or 0 = sorted batch: stand down */,
                                                               unsigned *__restrict__ tend = nullptr /* PAD: [ntiles] slots used */)
{
    constexpr int TILE = THREADS * ITEMS;
    if (gate && *gate == 0) return;
    constexpr int BPT = BM_NB / THREADS;  // buckets per thread in the scan
    static_assert(BM_NB % THREADS == 0 && (BPT == 2 || BPT == 4), "2 or 4 buckets per thread");
    constexpr int STAGED = PAD ? TILE + 3 * THREADS : TILE;  // (at most one unit per thread: three pad slots each)
    constexpr int STRIDE = PAD ? TILE + BM_PAD_ROOM : TILE;
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    unsigned *staged = reinterpret_cast<unsigned *>(dyn);                          // [STAGED] records in sorted order
    unsigned *cnt = staged + STAGED;                                               // [BM_NB]
    unsigned short *toff = reinterpret_cast<unsigned short *>(cnt + BM_NB);        // [BM_NB]
    unsigned *scan_tmp = reinterpret_cast<unsigned *>(toff + BM_NB);               // [16]
    const int64_t tile = blockIdx.x;
    const BmSeg &sg = segs[tile_seg[tile]];
    const int64_t ltile = tile - sg.tile0;
    if (ltile >= sg.ntiles) return;  // padding up to the next plan group
    const BmGeom g = sg.g;
    const int32_t *__restrict__ qs = sg.qs + ltile * TILE, *__restrict__ qe = sg.qe + ltile * TILE;  // this tile's queries
    const int64_t nq = sg.nq - ltile * TILE;
    recs += tile * STRIDE, slots += tile * TILE;  // scratch is laid out by the batch's tile numbering
    const int64_t base = 0;
    const int n = (int)(nq < TILE ? nq : TILE);
    int n_out = n;  // slots of the sorted tile (PAD: the units' gaps included)
    for (int i = threadIdx.x; i < BM_NB; i += THREADS) cnt[i] = 0;
    __syncthreads();
    unsigned br[ITEMS];  // bucket << 16 | rank inside the (tile, bucket) run
    int4 vs[ITEMS / 4], ve[ITEMS / 4];
    if (n == TILE) {
        const int4 *s4 = reinterpret_cast<const int4 *>(qs + base), *e4 = reinterpret_cast<const int4 *>(qe + base);
#pragma unroll
        for (int j = 0; j < ITEMS / 4; j++) vs[j] = s4[j * THREADS + threadIdx.x];
#pragma unroll
        for (int j = 0; j < ITEMS / 4; j++) ve[j] = e4[j * THREADS + threadIdx.x];
#pragma unroll
        for (int j = 0; j < ITEMS / 4; j++) {
            const unsigned bx = bm_bucket_of(vs[j].x, g), by = bm_bucket_of(vs[j].y, g), bz = bm_bucket_of(vs[j].z, g), bw = bm_bucket_of(vs[j].w, g);
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
                const unsigned b = bm_bucket_of(qs[base + k], g);
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
        unsigned short *row = tbl + tile * BM_NB + BPT * threadIdx.x;
        if (BPT == 4)
            *reinterpret_cast<uint2 *>(row) = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[BPT - 1] << 16));
        else
            *reinterpret_cast<unsigned *>(row) = (unsigned)o[0] | ((unsigned)o[1] << 16);
    }
    __syncthreads();
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
            staged[s0] = bm_record_of(vs[j].x, ve[j].x, g);
            staged[s1] = bm_record_of(vs[j].y, ve[j].y, g);
            staged[s2] = bm_record_of(vs[j].z, ve[j].z, g);
            staged[s3] = bm_record_of(vs[j].w, ve[j].w, g);
            l4[j * THREADS + threadIdx.x] = make_uint2(s0 | (s1 << 16), s2 | (s3 << 16));
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const int k = j * THREADS + threadIdx.x;
            if (k < n) {
                const unsigned s = toff[br[j] >> 16] + (br[j] & 0xffffu);
                staged[s] = bm_record_of(qs[base + k], qe[base + k], g);
                slots[base + k] = (unsigned short)s;
            }
        }
    }
    __syncthreads();
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
__global__ __launch_bounds__(256) void bm_transpose_kernel(const unsigned short *__restrict__ tbl, const BmSeg *__restrict__ segs,
                                                           const unsigned short *__restrict__ tile_seg, int tile_log2,
                                                           unsigned *__restrict__ runT /* [BM_NB][ntp] */, int64_t ntp,
                                                           unsigned *__restrict__ grpcnt /* [ngroups][BM_NB] */, const unsigned *__restrict__ gate)
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
        const unsigned short *row = tbl + tile * BM_NB + b0 + 16 * q;
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
        if (q == 3) t[r][64] = (unsigned short)(b0 + 64 < BM_NB ? (live ? row[16] : 0) : ntile);
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
        if (q == 0) grpcnt[(int64_t)grp * BM_NB + b0 + c] = sum;
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

// #{keys of the slice below coordinate lo + rel}, from one cell; `odd` collects "this cell is hard".
__device__ __forceinline__ int bm_cell_rank(lds_cell_p cells, unsigned rel, bool &odd)
{
    const unsigned long long c = cells[rel >> 5];
    const unsigned bits = (unsigned)c, meta = (unsigned)(c >> 32);
    const unsigned mask = (1u << (rel & 31u)) - 1u;  // the coordinates of the cell below rel
    const unsigned extra = meta >> 25;
    odd |= extra == (unsigned)BM_HARD;
    // the duplicated coordinate counts `extra` more times when it lies below rel: bit dpos of the mask says so
    return (int)(meta & 0xFFFFFu) + __popc(bits & mask) + (int)(extra * ((mask >> ((meta >> 20) & 31u)) & 1u));
}

__device__ __forceinline__ int bm_hard_rank(lds_cell_p cells, unsigned rel, const int32_t *__restrict__ a, int slice_lo, long long lo)
{
    const unsigned c = rel >> 5;
    const int r0 = slice_lo + (int)((unsigned)(cells[c] >> 32) & 0xFFFFFu), r1 = slice_lo + (int)((unsigned)(cells[c + 1] >> 32) & 0xFFFFFu);
    const long long key = lo + (long long)rel;
    if (key > INT_MAX) return r1 - slice_lo;  // every int32 key is below it
    return global_rank_lt(a, r0, r1, (int)key) - slice_lo;
}

// The count of one record (BM_REC_ESC = "ask the index again").  The common case is two LDS reads and ~30 VALU
// instructions; a hard cell or an escape record is noticed by one flag and redone off the fast path.
__device__ __forceinline__ unsigned bm_count_record(lds_cell_p cE, lds_cell_p cS, int eLo, int sLo, long long lo, unsigned rec,
                                                    const int32_t *__restrict__ s_ord, const int32_t *__restrict__ e_sorted)
{
    const bool esc = (rec >> 17) == BM_LEN_ESC;
    rec = esc ? (1u << 17) : rec;  // keep the lookups of an escape record inside the image
    const unsigned off = rec & 0x1FFFFu, len = rec >> 17;
    const unsigned relE = off + 1u, relS = off + len;
    bool odd = esc;
    const int rE = bm_cell_rank(cE, relE, odd);
    const int rS = bm_cell_rank(cS, relS, odd);
    unsigned c = (unsigned)((sLo - eLo) + (rS - rE));
    if (odd) {
        if (esc) {
            c = BM_REC_ESC;
        } else {
            bool e_hard = false, s_hard = false;
            int hE = bm_cell_rank(cE, relE, e_hard), hS = bm_cell_rank(cS, relS, s_hard);
            if (e_hard) hE = bm_hard_rank(cE, relE, e_sorted, eLo, lo);
            if (s_hard) hS = bm_hard_rank(cS, relS, s_ord, sLo, lo);
            c = (unsigned)((sLo - eLo) + (hS - hE));
        }
    }
    return c;
}

// One workgroup per work item = one bucket (PAIR: two neighbouring buckets, both images in LDS, one workgroup per CU)
// and a range of tiles.  L lanes (8, PAIR: 16) take one (tile, bucket) run -- the records of the item's bucket(s) in
// that tile, contiguous in the tile-sorted array, ~8 (16) of them on shuffled input: the first L records in one pass per
// run, what is left of the group's U runs flattened into shared passes, runs longer than 4 L left to the whole workgroup
// at the end.  The count of a record is written over the record (the line is in L2 from the read: a separate array of
// counts measured 2.8x write amplification, partial lines going out as 64-byte pieces).
// EXP (diagnostics, ivl.bm_exp): 0 = the real thing; 1 = no count stores, 2 = no record loads, 3 = neither (results are
// wrong then: the timing matrix of tools/bm_perf.py uses them to price the pieces of this kernel).
template <bool PAIR, int U /* tile runs in flight per lane group */, int EXP = 0>
__global__ __launch_bounds__(BM_SEARCH_THREADS) void bm_search_kernel(const BmSeg *__restrict__ segs, const int4 *__restrict__ items,
                                                                      const int *__restrict__ n_items, const unsigned *__restrict__ runT, int64_t ntp,
                                                                      unsigned *__restrict__ recs /* records in, counts out */, int tile_log2,
                                                                      const unsigned *__restrict__ gate)
{
    constexpr int L = PAIR ? 16 : 8;
    if (gate && *gate == 0) return;
    constexpr int NG = BM_SEARCH_THREADS / L;
    constexpr unsigned LONG_RUN = 4 * L;  // longer runs go to the cooperative finish
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    __shared__ uint2 s_long[BM_LONG_CAP];  // {first record, length} of the long runs met during the walk
    __shared__ int s_nlong;
    // Work item of this workgroup, XCD-aware: workgroup w runs on XCD w % 8 (observed dispatch order, speed only); giving
    // every XCD a contiguous range of items (= of buckets) lets the runs of neighbouring buckets, which share 128-byte
    // lines of the tile-sorted array, meet in one L2.
    const int nit = *n_items;
    const int per_xcd = (nit + 7) >> 3;
    const int slot = (int)(blockIdx.x >> 3);
    const int it = (int)(blockIdx.x & 7) * per_xcd + slot;
    if (slot >= per_xcd || it >= nit) return;
    const int4 item = items[it];
    const int b = item.x & 0xffff, t0 = item.y, t1 = item.z;
    const BmSeg &sg = segs[item.x >> 16];
    const BmGeom g = sg.g;
    const uint2 *__restrict__ images = sg.images;
    const BmBucket *__restrict__ bmeta = sg.bmeta;
    const int32_t *__restrict__ s_ord = sg.ix.s_ord, *__restrict__ e_sorted = sg.e_sorted;
    const unsigned *__restrict__ runs0 = runT + (int64_t)b * ntp;
    const unsigned *__restrict__ runs1 = runs0 + ntp;  // PAIR only
    const int gid = threadIdx.x / L, sub = threadIdx.x % L;
    unsigned run[U], run2[U];
#pragma unroll
    for (int u = 0; u < U; u++) {  // the first round's runs travel with the image
        const int t = t0 + u * NG + gid;
        run[u] = t < t1 ? runs0[t] : 0u;
        run2[u] = PAIR && t < t1 ? runs1[t] : 0u;
    }
    {
        // the image(s): every load of a lane issued before its first LDS store
        const int4 *src = reinterpret_cast<const int4 *>(images + (size_t)b * g.stride);
        const int n4 = (PAIR ? 2 : 1) * (g.stride >> 1);
        constexpr int SWEEPS = 5;
        for (int i0 = 0; i0 < n4; i0 += SWEEPS * BM_SEARCH_THREADS) {
            int4 v[SWEEPS];
#pragma unroll
            for (int k = 0; k < SWEEPS; k++) {
                const int i = i0 + k * BM_SEARCH_THREADS + (int)threadIdx.x;
                v[k] = src[i < n4 ? i : n4 - 1];  // (a valid address: no branch around the load)
            }
#pragma unroll
            for (int k = 0; k < SWEEPS; k++) {
                const int i = i0 + k * BM_SEARCH_THREADS + (int)threadIdx.x;
                if (i < n4) reinterpret_cast<int4 *>(dyn)[i] = v[k];
            }
        }
    }
    // the item's bucket b and, PAIR, its neighbour b + 1: cell arrays in LDS, ranks and coordinate of the first position
    const lds_cell_p cE0 = (lds_cell_p) reinterpret_cast<unsigned long long *>(dyn), cS0 = cE0 + g.nce;
    const int cells1 = PAIR ? g.stride : 0;  // the neighbour's image follows at this many cells
    const BmBucket bk0 = bmeta[b], bk1 = bmeta[PAIR ? b + 1 : b];
    const long long lo0 = (long long)g.cmin + ((long long)b << g.shift), lo1 = lo0 + (PAIR ? (long long)1 << g.shift : 0ll);
    if (threadIdx.x == 0) s_nlong = 0;
    __syncthreads();
    unsigned sink = 0;
    // one record: position p of a run that starts at record `first` and whose first `len0` records belong to bucket b
    auto answer = [&](unsigned first, unsigned p, unsigned len0, unsigned rec) {
        const bool second = PAIR && p >= len0;
        const int shift_cells = second ? cells1 : 0;
        const unsigned c = (EXP & 4) ? rec + (unsigned)shift_cells  // diagnostics: the memory traffic alone
                                     : bm_count_record(cE0 + shift_cells, cS0 + shift_cells, second ? bk1.eLo : bk0.eLo, second ? bk1.sLo : bk0.sLo,
                                                       second ? lo1 : lo0, rec, s_ord, e_sorted);
        if (EXP & 1)
            sink += c;
        else
            recs[(size_t)first + p] = c;
    };
    for (int tb = t0; tb < t1; tb += NG * U) {
        unsigned first[U], len0[U], len[U], rec[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = tb + u * NG + gid;
            first[u] = ((unsigned)t << tile_log2) + (run[u] & 0xffffu);  // (record indices stay below 2^32: nq < 2^31)
            len0[u] = run[u] >> 16;
            len[u] = len0[u] + (PAIR ? run2[u] >> 16 : 0u);
            if (EXP & 2)
                rec[u] = run[u] & 0xfff1ffffu;  // anything valid: short, never an escape
            else
                rec[u] = (unsigned)sub < len[u] ? recs[(size_t)first[u] + sub] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {  // the next round's runs, in flight while this round computes
            const int t = tb + NG * U + u * NG + gid;
            run[u] = t < t1 ? runs0[t] : 0u;
            run2[u] = PAIR && t < t1 ? runs1[t] : 0u;
        }
        // what the first pass leaves over, flattened across the group's U runs: c[u] = leftovers of runs < u
        unsigned cum[U + 1];
        cum[0] = 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            unsigned rem = len[u] > (unsigned)L ? len[u] - (unsigned)L : 0u;
            if (len[u] > LONG_RUN) {
                // rare on shuffled input, the rule on sorted or clumped input: a sorted batch is a handful of runs of
                // thousands of records per bucket, and the group that met one would otherwise work alone
                bool listed = false;
                if (sub == 0) {
                    const int k = atomicAdd(&s_nlong, 1);
                    if (k < BM_LONG_CAP) {
                        s_long[k] = make_uint2(first[u], len[u] | (len0[u] << 16));
                        listed = true;
                    }
                }
                listed = __shfl(listed, (int)(threadIdx.x & 63) - sub, 64);
                if (listed) rem = 0;  // (a full list: the group works the run off itself, exactness never depends on it)
            }
            cum[u + 1] = cum[u] + rem;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if ((unsigned)sub < len[u]) answer(first[u], (unsigned)sub, len0[u], rec[u]);
        for (unsigned base = 0; __any(base < cum[U]); base += L) {
            const unsigned i = base + (unsigned)sub;
            unsigned f = first[0], l0 = len0[0], lo_c = 0;
#pragma unroll
            for (int u = 1; u < U; u++) {
                const bool ge = i >= cum[u];
                f = ge ? first[u] : f;
                l0 = ge ? len0[u] : l0;
                lo_c = ge ? cum[u] : lo_c;
            }
            if (i < cum[U]) {
                const unsigned p = (unsigned)L + (i - lo_c);
                answer(f, p, l0, (EXP & 2) ? (f & 0xfff1ffffu) : recs[(size_t)f + p]);
            }
        }
    }
    __syncthreads();
    {
        const int nl = s_nlong < BM_LONG_CAP ? s_nlong : BM_LONG_CAP;
        for (int k = 0; k < nl; k++) {
            const uint2 e = s_long[k];
            const unsigned ll = e.y & 0xffffu, l0 = e.y >> 16;  // (a run is at most a tile: 2^15 records)
            for (unsigned p = (unsigned)L + threadIdx.x; p < ll; p += BM_SEARCH_THREADS) answer(e.x, p, l0, recs[(size_t)e.x + p]);
        }
    }
    if ((EXP & 1) && sink == 0x12345678u) recs[0] = 1;  // keeps the work of the store-less variant alive
}

// The same walk, software-pipelined: a wave that loads, then computes, then stores leaves the memory pipe idle while
// it computes and its SIMD idle while it waits -- with every wave slot of the CU taken (the images fill the LDS) the
// kernel time was the SUM of the two (measured: compute 120 us + record loads 140 us + count stores 240 us).  Here
// the records of round r + 1 are requested before round r is computed; all loads of a round, the first leftover pass
// included, are issued together and without branches so that the waits can count instead of draining.
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

template <bool PAIR, int U, bool NT /* image loads non-temporal: they stream through L2 once */>
__global__ __launch_bounds__(BM_SEARCH_THREADS) void bm_search_pipe_kernel(const BmSeg *__restrict__ segs, const int4 *__restrict__ items,
                                                                           const int *__restrict__ n_items, const unsigned *__restrict__ runT,
                                                                           int64_t ntp, unsigned *__restrict__ recs /* records in, counts out */,
                                                                           int tile_log2, const unsigned *__restrict__ gate)
{
    constexpr int L = PAIR ? 16 : 8;
    if (gate && *gate == 0) return;
    constexpr int NG = BM_SEARCH_THREADS / L;
    constexpr unsigned LONG_RUN = 4 * L;
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    __shared__ uint2 s_long[BM_LONG_CAP];
    __shared__ int s_nlong;
    const int nit = *n_items;
    const int per_xcd = (nit + 7) >> 3;
    const int slot = (int)(blockIdx.x >> 3);
    const int it = (int)(blockIdx.x & 7) * per_xcd + slot;
    if (slot >= per_xcd || it >= nit) return;
    const int4 item = items[it];
    const int b = item.x & 0xffff, t0 = item.y, t1 = item.z;
    const BmSeg &sg = segs[item.x >> 16];
    const BmGeom g = sg.g;
    const uint2 *__restrict__ images = sg.images;
    const BmBucket *__restrict__ bmeta = sg.bmeta;
    const int32_t *__restrict__ s_ord = sg.ix.s_ord, *__restrict__ e_sorted = sg.e_sorted;
    const unsigned *__restrict__ runs0 = runT + (int64_t)b * ntp;
    const unsigned *__restrict__ runs1 = runs0 + ntp;  // PAIR only
    const int gid = threadIdx.x / L, sub = threadIdx.x % L;
    unsigned run[U], run2[U];
    auto load_runs = [&](int tb) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = tb + u * NG + gid;
            const int tc = t < t1 ? t : t0;  // a valid address: no branch around the load
            const unsigned a = runs0[tc], c = PAIR ? runs1[tc] : 0u;
            run[u] = t < t1 ? a : 0u;
            run2[u] = t < t1 ? c : 0u;
        }
    };
    load_runs(t0);
    {
        const int4 *src = reinterpret_cast<const int4 *>(images + (size_t)b * g.stride);
        const int n4 = (PAIR ? 2 : 1) * (g.stride >> 1);
        constexpr int SWEEPS = 5;
        for (int i0 = 0; i0 < n4; i0 += SWEEPS * BM_SEARCH_THREADS) {
            int4 v[SWEEPS];
#pragma unroll
            for (int k = 0; k < SWEEPS; k++) {
                const int i = i0 + k * BM_SEARCH_THREADS + (int)threadIdx.x;
                if (NT) {
                    const bm_v4i w = __builtin_nontemporal_load(reinterpret_cast<const bm_v4i *>(src) + (i < n4 ? i : n4 - 1));
                    v[k] = make_int4(w.x, w.y, w.z, w.w);
                } else {
                    v[k] = src[i < n4 ? i : n4 - 1];
                }
            }
#pragma unroll
            for (int k = 0; k < SWEEPS; k++) {
                const int i = i0 + k * BM_SEARCH_THREADS + (int)threadIdx.x;
                if (i < n4) reinterpret_cast<int4 *>(dyn)[i] = v[k];
            }
        }
    }
    const lds_cell_p cE0 = (lds_cell_p) reinterpret_cast<unsigned long long *>(dyn), cS0 = cE0 + g.nce;
    const int cells1 = PAIR ? g.stride : 0;
    const BmBucket bk0 = bmeta[b], bk1 = bmeta[PAIR ? b + 1 : b];
    const long long lo0 = (long long)g.cmin + ((long long)b << g.shift), lo1 = lo0 + (PAIR ? (long long)1 << g.shift : 0ll);
    if (threadIdx.x == 0) s_nlong = 0;
    __syncthreads();
    auto answer = [&](unsigned at, bool second, unsigned rec) {
        const int shift_cells = second ? cells1 : 0;
        recs[(size_t)at] = bm_count_record(cE0 + shift_cells, cS0 + shift_cells, second ? bk1.eLo : bk0.eLo, second ? bk1.sLo : bk0.sLo,
                                           second ? lo1 : lo0, rec, s_ord, e_sorted);
    };
    // round `tb`: addresses from the runs in run[] / run2[], every load of the round issued
    auto prep = [&](BmRound<U> &R, int tb) {
        unsigned cum = 0, lf_at = ~0u, listed_mask = 0;
        bool lf_second = false;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = tb + u * NG + gid;
            const unsigned first = ((unsigned)t << tile_log2) + (run[u] & 0xffffu);
            const unsigned len0 = run[u] >> 16, len = len0 + (PAIR ? run2[u] >> 16 : 0u);
            R.first[u] = first;
            R.lens[u] = len0 | (len << 16);
            R.rec[u] = recs[(size_t)((unsigned)sub < len ? first + (unsigned)sub : 0u)];
            unsigned rem = len > (unsigned)L ? len - (unsigned)L : 0u;
            if (len > LONG_RUN) {  // left to the whole workgroup (see bm_search_kernel)
                bool listed = false;
                if (sub == 0) {
                    const int k = atomicAdd(&s_nlong, 1);
                    if (k < BM_LONG_CAP) {
                        s_long[k] = make_uint2(first, len | (len0 << 16));
                        listed = true;
                    }
                }
                listed = __shfl(listed, (int)(threadIdx.x & 63) - sub, 64);
                if (listed) {
                    rem = 0;
                    listed_mask |= 1u << u;
                }
            }
            const unsigned i = (unsigned)sub - cum;  // position among this run's leftovers, if it is this lane's turn
            if ((unsigned)sub >= cum && i < rem) {
                lf_at = first + (unsigned)L + i;
                lf_second = PAIR && (unsigned)L + i >= len0;
            }
            cum += rem;
        }
        R.lf_total = cum;
        R.listed = listed_mask;
        R.lf_at = lf_at;
        R.lf_second = lf_second;
        R.lf_rec = recs[(size_t)(lf_at != ~0u ? lf_at : 0u)];
    };
    auto finish = [&](BmRound<U> &R) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned len0 = R.lens[u] & 0xffffu, len = R.lens[u] >> 16;
            if ((unsigned)sub < len) answer(R.first[u] + (unsigned)sub, PAIR && (unsigned)sub >= len0, R.rec[u]);
        }
        if (R.lf_at != ~0u) answer(R.lf_at, R.lf_second, R.lf_rec);
        for (unsigned base = L; __any(base < R.lf_total); base += L) {  // further leftover passes: rare
            const unsigned i = base + (unsigned)sub;
            unsigned cum = 0, at = ~0u;
            bool second = false;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const unsigned len0 = R.lens[u] & 0xffffu, len = R.lens[u] >> 16;
                const unsigned rem = len > (unsigned)L && !((R.listed >> u) & 1u) ? len - (unsigned)L : 0u;
                const unsigned j = i - cum;
                if (i >= cum && j < rem) {
                    at = R.first[u] + (unsigned)L + j;
                    second = PAIR && (unsigned)L + j >= len0;
                }
                cum += rem;
            }
            if (at != ~0u) answer(at, second, recs[(size_t)at]);
        }
    };
    BmRound<U> A, B;
    prep(A, t0);
    load_runs(t0 + NG * U);
    for (int tb = t0; tb < t1; tb += 2 * NG * U) {
        prep(B, tb + NG * U);
        load_runs(tb + 2 * NG * U);
        finish(A);
        prep(A, tb + 2 * NG * U);
        load_runs(tb + 3 * NG * U);
        finish(B);
    }
    __syncthreads();
    {
        const int nl = s_nlong < BM_LONG_CAP ? s_nlong : BM_LONG_CAP;
        for (int k = 0; k < nl; k++) {
            const uint2 e = s_long[k];
            const unsigned ll = e.y & 0xffffu, l0 = e.y >> 16;
            for (unsigned p = (unsigned)L + threadIdx.x; p < ll; p += BM_SEARCH_THREADS)
                answer(e.x + p, PAIR && p >= l0, recs[(size_t)e.x + p]);
        }
    }
}

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

// FIND: also leave loff[tile-sorted position] = exclusive prefix of the counts inside the tile, in tile-sorted order
// (bit 31 = escape record, which contributes nothing): where the record's hits go in the tile's scratch region
// (count_slices.hpp, sl_fill_pipe_kernel).
template <int THREADS, int ITEMS, bool FIND = false>
__global__ __launch_bounds__(THREADS) void bm_unpermute_kernel(const unsigned *__restrict__ cnt /* tile-sorted: the records array after the search */,
                                                               const unsigned short *__restrict__ slots, const BmSeg *__restrict__ segs,
                                                               const unsigned short *__restrict__ tile_seg,
                                                               unsigned long long *__restrict__ total_slots /* [segments][PT_SLOTS], may be NULL */,
                                                               const unsigned *__restrict__ gate, unsigned *__restrict__ loff = nullptr)
{
    constexpr int TILE = THREADS * ITEMS;
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    if (gate && *gate == 0) return;
    unsigned *vals = reinterpret_cast<unsigned *>(dyn);  // [TILE]
    __shared__ long long red[THREADS / 64];
    const int64_t tile = blockIdx.x;
    const int seg_id = tile_seg[tile];
    const BmSeg &sg = segs[seg_id];
    const int64_t ltile = tile - sg.tile0;
    if (ltile >= sg.ntiles) return;  // padding up to the next plan group
    const IndexDev ix = sg.ix;
    const BmGeom g = sg.g;
    const int32_t *__restrict__ e_sorted = sg.e_sorted;
    const int32_t *__restrict__ qs_arr = sg.qs + ltile * TILE, *__restrict__ qe_arr = sg.qe + ltile * TILE;  // escapes only
    int32_t *__restrict__ out = sg.counts + ltile * TILE;
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
    if (FIND) {
        // every wave scans its contiguous share of the tile, 64 positions at a time
        constexpr int NW = THREADS / 64, PER_WAVE = TILE / NW, CH = PER_WAVE / 64;
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
            lo_out[idx] = ((e[c] & 0x7FFFFFFFu) + wbase) | (e[c] & 0x80000000u);
        }
    }
    long long acc = 0;
    if (n == TILE) {
        const uint2 *l4 = reinterpret_cast<const uint2 *>(slots + base);
        int4 *o4 = reinterpret_cast<int4 *>(out + base);
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
            o4[j * THREADS + threadIdx.x] = make_int4((int)c[0], (int)c[1], (int)c[2], (int)c[3]);
            acc += (long long)c[0] + c[1] + c[2] + c[3];
        }
    } else {
        for (int k = threadIdx.x; k < n; k += THREADS) {
            unsigned c = vals[slots[base + k]];
            if (c == BM_REC_ESC) c = (unsigned)bm_escape_count(ix, e_sorted, g, qs_arr[base + k], qe_arr[base + k]);
            out[base + k] = (int)c;
            acc += c;
        }
    }
    if (total_slots) block_accumulate_i64(acc, red, total_slots + (int64_t)seg_id * PT_SLOTS + (blockIdx.x & (PT_SLOTS - 1)));
}

// the segments' partial totals, folded into the caller's int64 per segment (accumulated, like bxmi_ivl_count_dev's total)
__global__ void bm_fold_totals_kernel(const unsigned long long *__restrict__ slots /* [segments][PT_SLOTS] */,
                                      unsigned long long *const *__restrict__ totals /* [segments] */,
                                      const unsigned *__restrict__ probe = nullptr, unsigned long long *__restrict__ order_host = nullptr,
                                      unsigned long long seq = 0)
{
    // (what the probe of bm_params_kernel saw, into host memory for the next calls: see bm_count_segments)
    if (order_host && blockIdx.x == 0 && threadIdx.x == 0) *order_host = (seq << 1) | (*probe != 0 ? 1ull : 0ull);
    if (!totals) return;
    unsigned long long v = threadIdx.x < PT_SLOTS ? slots[(int64_t)blockIdx.x * PT_SLOTS + threadIdx.x] : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (threadIdx.x == 0 && v && totals[blockIdx.x]) atomicAdd(totals[blockIdx.x], v);
}

}  // namespace bxmi
