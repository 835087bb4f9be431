// count_parts.hpp -- the first-generation bucketed count pass (part_* kernels: histogram, column scans, LDS-ordered scatter, tree / cell searches, gather) and the kernels for batches already sorted by start (ivl_local_*).  Selected by ivl.bitmap = 0 and by indexes no stage of the exchange serves.
// Included by intervals.hip (one translation unit; the kernels share its constants and device helpers).
#pragma once

namespace bxmi {

// ---------------------------------------------------------------------------
// partitioned count path (large batches)
// ---------------------------------------------------------------------------
// The direct kernel above is instruction-issue bound (rocprof: ~27 wave
// instructions per query, SIMDs 100 % busy) and pulls ~350 B/query through the
// fabric because random queries touch random leaves.  For big batches we make
// the accesses local instead:
//   1. bucket the queries by coordinate (2048 buckets over the targets' span):
//      histogram -> scan -> scatter (LDS atomics give the in-tile ranks);
//   2. one workgroup per (bucket, chunk): the bucket's slice of the sorted
//      ends/starts (a few thousand keys) is staged in LDS and every lane does
//      two plain binary searches there -- ~3 wave instructions per query, and
//      the targets are read from HBM once, coalesced;
//   3. counts come back in bucket order and are gathered into query order.
// Everything stays exact: slices are chosen so that ranks outside them are
// known, and anything that falls outside (very long / reversed queries) takes
// a per-lane global search.  rocprofv3 numbers for each step: DESIGN.md 3.1.
constexpr int PT_NB_LOG2 = 11;
constexpr int PT_NB = 1 << PT_NB_LOG2;      // coordinate buckets
constexpr int PT_THREADS = 1024;
#ifndef BXMI_PT_ITEMS
#define BXMI_PT_ITEMS 16
#endif
constexpr int PT_ITEMS = BXMI_PT_ITEMS;
constexpr int PT_TILE = PT_THREADS * PT_ITEMS;  // 16384 queries per partition tile (staged whole in LDS)
#ifndef BXMI_PT_CHUNK
#define BXMI_PT_CHUNK 65536
#endif
constexpr int PT_CHUNK = BXMI_PT_CHUNK;     // queries per search workgroup
constexpr int PT_LDS_INTS = 19456;          // 76 KiB of slices per workgroup -> two workgroups per CU
constexpr int PT_SLOTS = 64;                // spread the total over 64 counters (one atomic per workgroup)
constexpr int PT_ILP = 4;                   // queries in flight per lane in the search kernel
constexpr int LANE_WINDOW = 24;             // find(): windows up to this long are scanned by their own lane, longer ones by the whole wave

struct PartGeom {
    int32_t cmin;   // smallest coordinate of the bucket grid
    int32_t shift;  // bucket width = 1 << shift
};

struct SliceBound {
    int32_t eLo, eHi;    // staged slice of the sorted ends    [eLo, eHi)
    int32_t sLo, sHi;    // staged slice of the sorted starts  [sLo, sHi)
    int32_t qeLo, qeHi;  // rank_lt(starts, qe) may use the slice iff qeLo <= qe <= qeHi
    int32_t kE, kS;      // the slices are staged as perfect search trees of 2^k - 1 keys ...
    int32_t strideE, strideS;  // ... over every stride-th key (stride 1 = all of them: the tree alone gives the rank)
    int32_t pLo, pHi, kP, strideP;  // same for the prefix-max array (window start of find): ranks of pm <= qs
};

__device__ __forceinline__ int part_bucket(int qs, PartGeom g)
{
    if (qs < g.cmin) return 0;
    unsigned b = ((unsigned)qs - (unsigned)g.cmin) >> g.shift;
    return b < (unsigned)(PT_NB - 1) ? (int)b : PT_NB - 1;
}

// Workgroup -> tile, XCD-aware.  Workgroup w runs on XCD w % 8 (observed dispatch order; used for
// speed only).  Giving each XCD a CONTIGUOUS range of tiles means the four (tile, bucket) runs that
// share one 128-byte line of the bucketed arrays are written / read by the same XCD close in time,
// so its L2 merges them: measured 1.7x write and 3x read amplification without this.
__device__ __forceinline__ int64_t part_tile_of_block(int64_t ntiles)
{
    const int64_t per_xcd = (ntiles + 7) >> 3;
    return (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
}

// Histogram pass.  The LDS atomics that count a tile's buckets also hand every query its rank inside its (tile,
// bucket) run, and after one block scan the workgroup knows where each bucket starts inside the tile -- so the
// query's slot in the tile's sorted order (`lpos`, 16 bits) is written right here and the scatter needs no atomics
// of its own (LDS atomics run at ~1 lane/clk/CU: 0.16 ms per 100M, paid once instead of twice).
__global__ __launch_bounds__(PT_THREADS) void part_hist_kernel(const int32_t *__restrict__ qs, int64_t nq, PartGeom g,
                                                               unsigned *__restrict__ table /* [ntiles][PT_NB] */, int64_t ntiles,
                                                               unsigned short *__restrict__ lpos, unsigned *__restrict__ unsorted /* may be NULL */)
{
    __shared__ unsigned cnt[PT_NB];
    __shared__ unsigned short toff[PT_NB];
    __shared__ unsigned scan_tmp[16];
    const int64_t tile = part_tile_of_block(ntiles);
    if (tile >= ntiles) return;
    for (int i = threadIdx.x; i < PT_NB; i += PT_THREADS) cnt[i] = 0;
    __syncthreads();
    const int64_t base = tile * PT_TILE;
    const int n = (int)(nq - base < PT_TILE ? nq - base : PT_TILE);
    // While the starts stream by, notice whether they are already non-decreasing (a sorted BED file): such a batch
    // needs no bucketing at all and is answered by ivl_local_count_kernel instead (see there).
    bool descent = false;
    unsigned br[PT_ITEMS];  // bucket << 16 | rank inside the (tile, bucket) run
    if (n == PT_TILE) {
        // full tile: 4 x 16-byte loads in flight per lane before the first atomic
        const int4 *q4 = reinterpret_cast<const int4 *>(qs + base);
        int4 v[PT_ITEMS / 4];
        int nxt[PT_ITEMS / 4];
#pragma unroll
        for (int j = 0; j < PT_ITEMS / 4; j++) {
            v[j] = q4[j * PT_THREADS + threadIdx.x];
            int64_t k = base + 4 * (int64_t)(j * PT_THREADS + threadIdx.x) + 4;
            nxt[j] = unsorted && k < nq ? qs[k] : INT_MAX;
        }
#pragma unroll
        for (int j = 0; j < PT_ITEMS / 4; j++) {
            descent |= v[j].x > v[j].y || v[j].y > v[j].z || v[j].z > v[j].w || v[j].w > nxt[j];
            const unsigned bx = part_bucket(v[j].x, g), by = part_bucket(v[j].y, g), bz = part_bucket(v[j].z, g), bw = part_bucket(v[j].w, g);
            // Sorted input puts the wave's 256 consecutive queries in one bucket, and 256 same-address LDS atomics
            // serialize (measured 4.4x on a sorted batch): one lane adds for the whole wave then.
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
                br[4 * j + 0] = (bx << 16) | atomicAdd(&cnt[bx], 1u);
                br[4 * j + 1] = (by << 16) | atomicAdd(&cnt[by], 1u);
                br[4 * j + 2] = (bz << 16) | atomicAdd(&cnt[bz], 1u);
                br[4 * j + 3] = (bw << 16) | atomicAdd(&cnt[bw], 1u);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < PT_ITEMS; j++) {
            const int k = j * PT_THREADS + threadIdx.x;
            if (k < n) {
                int a = qs[base + k];
                descent |= base + k + 1 < nq && a > qs[base + k + 1];
                unsigned b = (unsigned)part_bucket(a, g);
                br[j] = (b << 16) | atomicAdd(&cnt[b], 1u);
            }
        }
    }
    if (unsorted && __ballot(descent) && lane_id() == 0 && *unsorted == 0) *unsorted = 1;
    __syncthreads();
    {
        unsigned a = cnt[2 * threadIdx.x], b = cnt[2 * threadIdx.x + 1];
        unsigned tot;
        unsigned exc = block_exclusive_scan(a + b, OpSum(), 0u, scan_tmp, &tot);
        toff[2 * threadIdx.x] = (unsigned short)exc;
        toff[2 * threadIdx.x + 1] = (unsigned short)(exc + a);
        *reinterpret_cast<uint2 *>(table + tile * PT_NB + 2 * threadIdx.x) = make_uint2(a, b);
    }
    __syncthreads();
    if (n == PT_TILE) {
        uint2 *l4 = reinterpret_cast<uint2 *>(lpos + base);  // four 16-bit slots per 8-byte store
#pragma unroll
        for (int j = 0; j < PT_ITEMS / 4; j++) {
            unsigned s0 = toff[br[4 * j + 0] >> 16] + (br[4 * j + 0] & 0xffffu), s1 = toff[br[4 * j + 1] >> 16] + (br[4 * j + 1] & 0xffffu);
            unsigned s2 = toff[br[4 * j + 2] >> 16] + (br[4 * j + 2] & 0xffffu), s3 = toff[br[4 * j + 3] >> 16] + (br[4 * j + 3] & 0xffffu);
            l4[j * PT_THREADS + threadIdx.x] = make_uint2(s0 | (s1 << 16), s2 | (s3 << 16));
        }
    } else {
#pragma unroll
        for (int j = 0; j < PT_ITEMS; j++) {
            const int k = j * PT_THREADS + threadIdx.x;
            if (k < n) lpos[base + k] = (unsigned short)(toff[br[j] >> 16] + (br[j] & 0xffffu));
        }
    }
}

// The table is tile-major ([tile][bucket], every workgroup reads/writes its own 8 KiB row coalesced).
// Destination of (tile t, bucket b) = sum of all counts of buckets < b, plus counts of bucket b in
// tiles < t: a scan DOWN the columns after a scan ACROSS the column totals, in three small kernels.
__global__ __launch_bounds__(PT_THREADS) void part_colsum_kernel(const unsigned *__restrict__ table, int64_t ntiles, int rows_per_block,
                                                                 unsigned *__restrict__ partial /* [nblocks][PT_NB] */,
                                                                 const unsigned *__restrict__ gate)
{
    if (gate && *gate == 0) return;  // sorted batch: the bucketed path is skipped
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < ntiles ? r0 + rows_per_block : ntiles;
    unsigned s0 = 0, s1 = 0;
    int64_t r = r0;
    for (; r + 16 <= r1; r += 16) {  // 32 loads in flight: the kernel is a chain of round trips otherwise
        unsigned v0[16], v1[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            v0[i] = table[(r + i) * PT_NB + threadIdx.x];
            v1[i] = table[(r + i) * PT_NB + PT_THREADS + threadIdx.x];
        }
#pragma unroll
        for (int i = 0; i < 16; i++) s0 += v0[i], s1 += v1[i];
    }
    for (; r < r1; r++) {
        s0 += table[r * PT_NB + threadIdx.x];
        s1 += table[r * PT_NB + PT_THREADS + threadIdx.x];
    }
    partial[(int64_t)blockIdx.x * PT_NB + threadIdx.x] = s0;
    partial[(int64_t)blockIdx.x * PT_NB + PT_THREADS + threadIdx.x] = s1;
}

__global__ __launch_bounds__(PT_THREADS) void part_colbase_kernel(unsigned *__restrict__ partial, int nblocks, int64_t nq,
                                                                  int32_t *__restrict__ wg_first /* [PT_NB + 1] */,
                                                                  const unsigned *__restrict__ gate)
{
    __shared__ unsigned scan_tmp[16];
    if (gate && *gate == 0) return;
    __shared__ int scan_tmp_i[16];
    // thread t owns the adjacent columns 2t and 2t+1 (so that one block scan orders all 2048 buckets)
    const int c0 = 2 * threadIdx.x, c1 = c0 + 1;
    unsigned t0 = 0, t1 = 0;
    {
        int r = 0;
        for (; r + 16 <= nblocks; r += 16) {
            uint2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = *reinterpret_cast<const uint2 *>(partial + (int64_t)(r + i) * PT_NB + c0);
#pragma unroll
            for (int i = 0; i < 16; i++) t0 += v[i].x, t1 += v[i].y;
        }
        for (; r < nblocks; r++) {
            uint2 v = *reinterpret_cast<const uint2 *>(partial + (int64_t)r * PT_NB + c0);
            t0 += v.x;
            t1 += v.y;
        }
    }
    unsigned tot;
    unsigned base0 = block_exclusive_scan(t0 + t1, OpSum(), 0u, scan_tmp, &tot);
    unsigned base1 = base0 + t0;
    // search-workgroup plan: bucket b gets ceil(n_b / PT_CHUNK) workgroups
    int ch0 = (int)((t0 + PT_CHUNK - 1) / PT_CHUNK), ch1 = (int)((t1 + PT_CHUNK - 1) / PT_CHUNK);
    int chtot;
    int w0 = block_exclusive_scan(ch0 + ch1, OpSum(), 0, scan_tmp_i, &chtot);
    wg_first[c0] = w0;
    wg_first[c1] = w0 + ch0;
    if (threadIdx.x == 0) wg_first[PT_NB] = chtot;
    unsigned run0 = base0, run1 = base1;
    {
        int r = 0;
        for (; r + 16 <= nblocks; r += 16) {
            uint2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = *reinterpret_cast<const uint2 *>(partial + (int64_t)(r + i) * PT_NB + c0);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                *reinterpret_cast<uint2 *>(partial + (int64_t)(r + i) * PT_NB + c0) = make_uint2(run0, run1);
                run0 += v[i].x;
                run1 += v[i].y;
            }
        }
        for (; r < nblocks; r++) {
            uint2 *cell = reinterpret_cast<uint2 *>(partial + (int64_t)r * PT_NB + c0);
            uint2 v = *cell;
            *cell = make_uint2(run0, run1);
            run0 += v.x;
            run1 += v.y;
        }
    }
}

__global__ __launch_bounds__(PT_THREADS) void part_colscan_kernel(unsigned *__restrict__ table, int64_t ntiles, int rows_per_block,
                                                                  const unsigned *__restrict__ partial,
                                                                  const unsigned *__restrict__ gate)
{
    if (gate && *gate == 0) return;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < ntiles ? r0 + rows_per_block : ntiles;
    unsigned run0 = partial[(int64_t)blockIdx.x * PT_NB + threadIdx.x];
    unsigned run1 = partial[(int64_t)blockIdx.x * PT_NB + PT_THREADS + threadIdx.x];
    int64_t r = r0;
    for (; r + 16 <= r1; r += 16) {
        unsigned v0[16], v1[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            v0[i] = table[(r + i) * PT_NB + threadIdx.x];
            v1[i] = table[(r + i) * PT_NB + PT_THREADS + threadIdx.x];
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            table[(r + i) * PT_NB + threadIdx.x] = run0;
            table[(r + i) * PT_NB + PT_THREADS + threadIdx.x] = run1;
            run0 += v0[i];
            run1 += v1[i];
        }
    }
    for (; r < r1; r++) {
        unsigned v0 = table[r * PT_NB + threadIdx.x], v1 = table[r * PT_NB + PT_THREADS + threadIdx.x];
        table[r * PT_NB + threadIdx.x] = run0;
        table[r * PT_NB + PT_THREADS + threadIdx.x] = run1;
        run0 += v0;
        run1 += v1;
    }
}

// One workgroup moves one tile of 16384 queries into bucket order.  A scattered 4-byte store
// costs a whole L2 request, so the tile is ordered INSIDE LDS first ((qs,qe) pairs written to the
// slot of the tile's sorted order that the histogram pass recorded in `lpos`) and then streamed
// out: consecutive lanes store to consecutive addresses, one request per (tile, bucket) run.
__global__ __launch_bounds__(PT_THREADS) void part_scatter_kernel(const int32_t *__restrict__ qs, const int32_t *__restrict__ qe,
                                                                  int64_t nq, PartGeom g,
                                                                  const unsigned *__restrict__ tile_table /* [ntiles][PT_NB] */,
                                                                  int64_t ntiles, int2 *__restrict__ pairs_out /* (qs, qe) in bucket order */,
                                                                  const unsigned short *__restrict__ lpos,
                                                                  const unsigned *__restrict__ gate)
{
    // LDS: the whole tile of (qs, qe) pairs (128 KiB) + one 2048-entry table (8 KiB).  The tile's pairs and slots live
    // in registers (102 VGPRs), so ONE workgroup runs per CU whatever the LDS footprint; staging half a tile at a time
    // (80 KiB) measured 4 % slower, and keeping only the slots in registers and re-reading the pairs (64 VGPRs, two
    // workgroups per CU) measured 0.61 ms against 0.46 ms -- more tiles in flight spread the runs that share a
    // 128-byte line further apart in time.  A persistent grid (one workgroup per CU looping over its tiles, next tile's
    // loads issued before the current one is streamed out) measured +23 % on the whole pass: the workgroups march in
    // step and the load and store bursts stop overlapping.
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    int2 *staged = reinterpret_cast<int2 *>(dyn);                    // [PT_TILE] (qs, qe) in bucket order
    unsigned *delta = reinterpret_cast<unsigned *>(dyn + 2 * PT_TILE);   // [PT_NB] global base of the (tile, bucket) run - its offset in the tile
    unsigned *scan_tmp = reinterpret_cast<unsigned *>(dyn);           // the staging area is idle during the scan
    const int64_t tile = part_tile_of_block(ntiles);
    if (tile >= ntiles) return;
    const unsigned go = gate ? *gate : 1u;  // 0 = sorted batch, answered elsewhere
    const int64_t base = tile * PT_TILE;
    const int n = (int)(nq - base < PT_TILE ? nq - base : PT_TILE);
    int s[PT_ITEMS], e[PT_ITEMS];
    unsigned slot[PT_ITEMS];
    if (go == 0) return;
#pragma unroll
    for (int j = 0; j < PT_ITEMS; j++) {
        int k = j * PT_THREADS + threadIdx.x;
        if (k < n) {
            s[j] = qs[base + k];
            e[j] = qe[base + k];
            slot[j] = lpos[base + k];
        }
    }
    {
        // Tile counts = distance to the next entry of the (linear, bucket-major) exclusive scan: the next
        // tile's entry for the same bucket, or -- for the last tile -- tile 0's entry of the next bucket.
        const bool last_tile = tile + 1 == ntiles;
        const unsigned *row = tile_table + tile * PT_NB;
        const unsigned *next = last_tile ? tile_table : row + PT_NB;
        unsigned c[2], lo[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            int b = 2 * threadIdx.x + u;
            lo[u] = row[b];
            unsigned hi = !last_tile ? next[b] : (b + 1 < PT_NB ? next[b + 1] : (unsigned)nq);
            c[u] = hi - lo[u];
        }
        unsigned tot;
        unsigned exc = block_exclusive_scan(c[0] + c[1], OpSum(), 0u, scan_tmp, &tot);
        delta[2 * threadIdx.x] = lo[0] - exc;
        delta[2 * threadIdx.x + 1] = lo[1] - (exc + c[0]);
    }
    __syncthreads();  // table ready, scan scratch free
#pragma unroll
    for (int j = 0; j < PT_ITEMS; j++) {
        int k = j * PT_THREADS + threadIdx.x;
        if (k < n) staged[slot[j]] = make_int2(s[j], e[j]);
    }
    __syncthreads();
#pragma unroll 4
    for (int p = threadIdx.x; p < n; p += PT_THREADS) {
        int2 v = staged[p];
        unsigned d = delta[part_bucket(v.x, g)] + (unsigned)p;  // global base of the run + offset inside it
        pairs_out[d] = v;  // one 8-byte store: a (tile, bucket) run is 64 contiguous bytes
    }
}

// Finish a sampled-tree rank inside one group of `stride` keys: short groups are counted with independent loads
// (one round trip, usually one line); long ones fall back to a binary search.
__device__ __forceinline__ int group_rank_lt(const int32_t *__restrict__ a, int lo, int hi, int key);

__device__ __forceinline__ int global_rank_lt(const int32_t *__restrict__ a, int lo, int hi, int key)
{
    while (lo < hi) {
        int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if (a[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// Counts travel back to query order as 16 bits when they fit: 0xFFFF says "ask again" and the gather recomputes
// that query from the index (exact; only pile-ups of >= 65535 overlapping targets ever take it).  Halves the bytes of
// the counts' round trip (0.4 GB of the pass at 100M queries).
constexpr unsigned COUNT_ESCAPE = 0xFFFFu;
__device__ __forceinline__ void store_count(int32_t *p, int64_t i, int c) { p[i] = c; }
__device__ __forceinline__ void store_count(unsigned short *p, int64_t i, int c)
{
    p[i] = (unsigned short)((unsigned)c < COUNT_ESCAPE ? (unsigned)c : COUNT_ESCAPE);
}

// One query straight from the sealed index (global binary searches): the escape path of the 16-bit counts.
__device__ __forceinline__ int count_one_global(const IndexDev &ix, const int32_t *__restrict__ e_sorted, int qs, int qe);

// Which bucket / which queries does this search workgroup own?  (shared prologue of the count and window kernels)
__device__ __forceinline__ bool part_chunk_of_block(const int32_t *__restrict__ wg_first, const unsigned *__restrict__ table,
                                                    int64_t nq, int *s_bucket, int &b, int64_t &q_begin, int64_t &q_end)
{
    if (threadIdx.x == 0) *s_bucket = -1;
    __syncthreads();
    const int w = (int)blockIdx.x;
#pragma unroll
    for (int u = 0; u < PT_NB / PT_THREADS; u++) {
        int c = u * PT_THREADS + threadIdx.x;
        if (wg_first[c] <= w && w < wg_first[c + 1]) *s_bucket = c;
    }
    __syncthreads();
    b = __builtin_amdgcn_readfirstlane(*s_bucket);  // uniform: everything derived from it lives in SGPRs
    if (b < 0) return false;
    const int64_t q_lo = table[b];
    const int64_t q_hi = b + 1 < PT_NB ? (int64_t)table[b + 1] : nq;
    q_begin = q_lo + (int64_t)(w - wg_first[b]) * PT_CHUNK;
    q_end = q_begin + PT_CHUNK < q_hi ? q_begin + PT_CHUNK : q_hi;
    return true;
}

// Stage `m = n / stride` samples of a sorted slice as a perfect Eytzinger tree of 2^k slots (slot 0 unused).
template <int THREADS>
__device__ __forceinline__ void part_stage_tree(int32_t *tree, int k, const int32_t *__restrict__ src, int n, int stride)
{
    const int m = n / stride;
    for (int r = threadIdx.x; r < m; r += THREADS) {
        int tpos = r + 1, z = __ffs(tpos) - 1;  // in-order number and height of the node holding sample r
        tree[(tpos >> (z + 1)) + (1 << (k - 1 - z))] = src[(r + 1) * stride - 1];
    }
}

__device__ __forceinline__ int group_rank_lt(const int32_t *__restrict__ a, int lo, int hi, int key)
{
    if (hi - lo > 8) return global_rank_lt(a, lo, hi, key);
    int c = lo;
#pragma unroll
    for (int u = 0; u < 8; u++) c += (lo + u < hi) && a[lo + u < hi ? lo + u : lo] < key;
    return c;
}

__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// ---- the same search with direct addressing instead of trees (default) ----
// Inside one bucket the keys are close to uniform, so most of a binary search is wasted: the bucket's coordinate
// range is cut into 4096 (+512 for the starts, whose keys reach W/8 past the bucket) equal cells, `cs[c]` = number of
// slice keys whose cell is below c (16 bits), and a rank is  cs[cell(key)] + (a 2-4 step search among the cell's
// own keys)  -- the step count is the bit length of the fullest cell, found while staging, so dense or clumped
// buckets just take more steps and stay exact.  cell() is monotone (clamped), hence keys in lower cells are smaller
// and keys in higher cells larger than the probe whatever the clamping does.  ~20 lane-instructions per rank
// instead of ~52 for the 13-level tree.
constexpr int PC_CELLS_LOG2 = 12;
constexpr int PC_NC = (1 << PC_CELLS_LOG2) + (1 << (PC_CELLS_LOG2 - 3));  // 4608
constexpr int PC_CS_INTS = (PC_NC + 2) / 2;                                // one 16-bit table, in ints
#ifndef BXMI_PC_ILP
#define BXMI_PC_ILP 4
#endif
constexpr int PC_ILP = BXMI_PC_ILP;  // queries in flight per lane
constexpr int PC_PAD = 64;                                                 // INT_MAX fence after each slice: searches of <= 6 steps need no bound check
constexpr int PC_KEYS = (PT_LDS_INTS - 2 * PC_CS_INTS - 2 * PC_PAD) / 2;  // keys (or samples) per staged slice: 7 359

struct CellMap {
    int lo, hi;  // coordinates of the first cell's first and the last cell's last position
    int cshift;  // cell width = 1 << cshift
};
__device__ __forceinline__ int cell_of(int x, CellMap m)
{
    x = x < m.lo ? m.lo : x;  // (a v_med3_i32)
    x = x > m.hi ? m.hi : x;
    return (int)(((unsigned)x - (unsigned)m.lo) >> m.cshift);
}
__device__ __forceinline__ CellMap cell_map_of(int b, PartGeom g)
{
    CellMap cm;
    long long lo = (long long)g.cmin + ((long long)b << g.shift);
    cm.lo = lo > INT_MAX ? INT_MAX : (int)lo;
    cm.cshift = g.shift > PC_CELLS_LOG2 ? g.shift - PC_CELLS_LOG2 : 0;
    long long hi = (long long)cm.lo + ((long long)PC_NC << cm.cshift) - 1;
    cm.hi = hi > INT_MAX ? INT_MAX : (int)hi;
    return cm;
}
typedef __attribute__((address_space(3))) const int32_t *lds_i32p;
typedef __attribute__((address_space(3))) const unsigned short *lds_u16p;

// Stage m = n / stride samples of a sorted slice linearly (arr[m] = INT_MAX fence) and build its cell table.
// Returns the number of search steps: the bit length of the fullest cell.
__device__ __forceinline__ int cells_stage(int32_t *arr, unsigned short *cs, const int32_t *__restrict__ src, int n, int stride, CellMap cm,
                                           int *s_red /* [16] */)
{
    const int m = n / stride;
    for (int r = threadIdx.x; r < m; r += PT_THREADS) arr[r] = src[(r + 1) * stride - 1];
    if (threadIdx.x < PC_PAD) arr[m + threadIdx.x] = INT_MAX;
    __syncthreads();
    // element r opens every cell in (cell(arr[r-1]), cell(arr[r])]; the virtual element m closes the table
    for (int r = threadIdx.x; r <= m; r += PT_THREADS) {
        const int cp = r == 0 ? -1 : cell_of(arr[r - 1], cm);
        const int cr = r == m ? PC_NC - 1 : cell_of(arr[r], cm);
        for (int c = cp + 1; c <= cr; c++) cs[c] = (unsigned short)r;
    }
    __syncthreads();
    int pop = 0;
    for (int c = threadIdx.x; c < PC_NC; c += PT_THREADS) {
        int p = (c + 1 < PC_NC ? (int)cs[c + 1] : m) - (int)cs[c];
        pop = p > pop ? p : pop;
    }
    pop = wave_max_i32(pop);
    if (lane_id() == 0) s_red[threadIdx.x >> 6] = pop;
    __syncthreads();
    pop = 0;
#pragma unroll
    for (int i = 0; i < PT_THREADS / 64; i++) pop = s_red[i] > pop ? s_red[i] : pop;
    __syncthreads();
    return 32 - __clz(pop);  // 0 for an empty slice
}

// What a search workgroup needs of its bucket, ready to be copied into LDS: [csE][csS][arrE + fence][arrS + fence].
// It depends only on the sealed index, so it is built once per bucket (part_cells_image_kernel, on the first large batch) and the search
// kernel starts with one streaming copy instead of two gathers, two table builds and eight barriers (measured ~25 us
// per workgroup, a quarter of the kernel).
struct CellsMeta {
    int mE, mS;            // staged keys (or samples) of the ends / starts slice
    int strideE, strideS;  // 1 = every key
    int stepsE, stepsS;    // search steps inside a cell
    int used_ints;         // ints of the image in use
    int pad;
};

__global__ __launch_bounds__(PT_THREADS) void part_cells_image_kernel(IndexDev ix, const int32_t *__restrict__ e_sorted,
                                                                      const SliceBound *__restrict__ bounds, PartGeom g,
                                                                      int32_t *__restrict__ images /* [PT_NB][PT_LDS_INTS] */,
                                                                      CellsMeta *__restrict__ meta)
{
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    __shared__ int s_red[PT_THREADS / 64];
    const int b = blockIdx.x;
    const SliceBound sb = bounds[b];
    const int nE = sb.eHi - sb.eLo, nS = sb.sHi - sb.sLo;
    // strides chosen for the tree kernel may be finer than this layout holds: widen if needed
    const int strideE = nE / sb.strideE > PC_KEYS ? nE / PC_KEYS + 1 : sb.strideE;
    const int strideS = nS / sb.strideS > PC_KEYS ? nS / PC_KEYS + 1 : sb.strideS;
    const int mE = nE / strideE, mS = nS / strideS;
    const CellMap cm = cell_map_of(b, g);
    unsigned short *csE = reinterpret_cast<unsigned short *>(lds), *csS = reinterpret_cast<unsigned short *>(lds + PC_CS_INTS);
    int32_t *arrE = lds + 2 * PC_CS_INTS, *arrS = arrE + mE + PC_PAD;
    const int stepsE = cells_stage(arrE, csE, e_sorted + sb.eLo, nE, strideE, cm, s_red);
    const int stepsS = cells_stage(arrS, csS, ix.s_ord + sb.sLo, nS, strideS, cm, s_red);
    const int used = ((2 * PC_CS_INTS + mE + mS + 2 * PC_PAD) + 3) & ~3;
    __syncthreads();
    int4 *dst = reinterpret_cast<int4 *>(images + (int64_t)b * PT_LDS_INTS);
    for (int i = threadIdx.x; i < used / 4; i += PT_THREADS) dst[i] = reinterpret_cast<const int4 *>(lds)[i];
    if (threadIdx.x == 0) meta[b] = CellsMeta{mE, mS, strideE, strideS, stepsE, stepsS, used, 0};
}

template <typename CT>
__global__ __launch_bounds__(PT_THREADS) void part_count_cells_kernel(IndexDev ix, const int32_t *__restrict__ e_sorted,
                                                                      const SliceBound *__restrict__ bounds,
                                                                      const int32_t *__restrict__ images, const CellsMeta *__restrict__ meta,
                                                                      const int32_t *__restrict__ wg_first,
                                                                      const unsigned *__restrict__ table /* row 0 = bucket offsets */,
                                                                      const int2 *__restrict__ pairs /* (qs, qe), bucket order */, int64_t nq,
                                                                      PartGeom g,
                                                                      CT *__restrict__ counts /* bucket order, may be NULL */,
                                                                      unsigned long long *__restrict__ total_slots,
                                                                      const unsigned *__restrict__ gate)
{
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    __shared__ int s_bucket;
    __shared__ long long red[PT_THREADS / 64];
    int b;
    int64_t q_begin, q_end;
    const unsigned go = gate ? *gate : 1u;  // 0 = sorted batch, answered by ivl_local_count_kernel
    if (!part_chunk_of_block(wg_first, table, nq, &s_bucket, b, q_begin, q_end) || go == 0) return;
    const SliceBound sb = bounds[b];
    const CellsMeta cmeta = meta[b];
    const int nE = sb.eHi - sb.eLo, nS = sb.sHi - sb.sLo;
    const int strideE = cmeta.strideE, strideS = cmeta.strideS, mE = cmeta.mE, mS = cmeta.mS;
    const int stepsE = cmeta.stepsE, stepsS = cmeta.stepsS;
    const CellMap cm = cell_map_of(b, g);
    {
        const int4 *src = reinterpret_cast<const int4 *>(images + (int64_t)b * PT_LDS_INTS);
        for (int i = threadIdx.x; i < cmeta.used_ints / 4; i += PT_THREADS) reinterpret_cast<int4 *>(lds)[i] = src[i];
    }
    __syncthreads();
    unsigned short *csE = reinterpret_cast<unsigned short *>(lds), *csS = reinterpret_cast<unsigned short *>(lds + PC_CS_INTS);
    int32_t *arrE = lds + 2 * PC_CS_INTS, *arrS = arrE + mE + PC_PAD;
    // positions are LDS pointers to "the last key known to be below the probe" (one add + one read per step)
    const lds_i32p aE = (lds_i32p)arrE, aS = (lds_i32p)arrS;
    const lds_u16p cE = (lds_u16p)csE, cS = (lds_u16p)csS;
    const bool fenced = stepsE <= 6 && stepsS <= 6;  // every probe stays inside the INT_MAX fence
    // The common case -- an ordinary query (qs < qe, qe inside the staged slice) against unsampled slices -- is kept
    // lean: 32-bit offsets from the chunk's base, count = (pS - pE) + constant, one test per round for "anything unusual".
    const unsigned nch = (unsigned)(q_end - q_begin);
    const int2 *__restrict__ qb = pairs + q_begin;
    CT *__restrict__ cb = counts ? counts + q_begin : nullptr;
    const bool unsampled = strideS == 1 && strideE == 1;
    const int cconst = (sb.sLo - sb.eLo) - (int)(aS - aE);
    const unsigned qe_span = (unsigned)sb.qeHi - (unsigned)sb.qeLo;
    long long acc = 0;
    for (unsigned u0 = threadIdx.x; u0 < nch; u0 += PT_THREADS * PC_ILP) {
        int qs[PC_ILP], qe[PC_ILP];
        lds_i32p pS[PC_ILP], pE[PC_ILP];
#pragma unroll
        for (int j = 0; j < PC_ILP; j++) {
            unsigned u = u0 + (unsigned)j * PT_THREADS;
            u = u < nch ? u : nch - 1;  // a valid address: no branch around the loads
            const int2 v = qb[u];
            qs[j] = v.x;
            qe[j] = v.y;
        }
#pragma unroll
        for (int j = 0; j < PC_ILP; j++) {
            pS[j] = aS + cS[cell_of(qe[j], cm)] - 1;
            pE[j] = aE + cE[cell_of(qs[j], cm)] - 1;
        }
        // only keys of the probe's own cell can still qualify, everything in later cells is larger, the fence stops the walk
        if (fenced) {
            for (int st = stepsS - 1; st >= 0; st--) {
#pragma unroll
                for (int j = 0; j < PC_ILP; j++) {
                    const lds_i32p t = pS[j] + (1 << st);
                    pS[j] = *t < qe[j] ? t : pS[j];
                }
            }
            for (int st = stepsE - 1; st >= 0; st--) {
#pragma unroll
                for (int j = 0; j < PC_ILP; j++) {
                    const lds_i32p t = pE[j] + (1 << st);
                    pE[j] = *t <= qs[j] ? t : pE[j];  // (qs == INT_MAX passes the fence: handled below)
                }
            }
        } else {
            const lds_i32p endS = aS + mS, endE = aE + mE;
            for (int st = stepsS - 1; st >= 0; st--) {
#pragma unroll
                for (int j = 0; j < PC_ILP; j++) {
                    lds_i32p t = pS[j] + (1 << st);
                    t = t < endS ? t : endS;
                    pS[j] = *t < qe[j] ? t : pS[j];
                }
            }
            for (int st = stepsE - 1; st >= 0; st--) {
#pragma unroll
                for (int j = 0; j < PC_ILP; j++) {
                    lds_i32p t = pE[j] + (1 << st);
                    t = t < endE ? t : endE;
                    pE[j] = *t <= qs[j] ? t : pE[j];
                }
            }
        }
        int c[PC_ILP];
        bool odd = !unsampled;
#pragma unroll
        for (int j = 0; j < PC_ILP; j++) {
            c[j] = (int)(pS[j] - pE[j]) + cconst;  // (sLo + #starts < qe) - (eLo + #ends <= qs)
            odd |= !(qs[j] < qe[j]) | ((unsigned)qe[j] - (unsigned)sb.qeLo > qe_span);
        }
        if (odd) {
#pragma unroll
            for (int j = 0; j < PC_ILP; j++) {
                const bool in_slice = (unsigned)qe[j] - (unsigned)sb.qeLo <= qe_span;
                if (unsampled && qs[j] < qe[j] && in_slice) continue;
                // sampled slices: finish each rank inside its group; qe outside the slice: global search;
                // zero-length / reversed query: exact predicate over the candidate window
                int rS = ((int)(pS[j] - aS) + 1) * strideS, rE = ((int)(pE[j] - aE) + 1) * strideE;
                if (strideS > 1) rS = group_rank_lt(ix.s_ord + sb.sLo, rS, rS + strideS < nS ? rS + strideS : nS, qe[j]);
                if (strideE > 1 && qs[j] != INT_MAX) rE = group_rank_lt(e_sorted + sb.eLo, rE, rE + strideE < nE ? rE + strideE : nE, qs[j] + 1);
                const int s_rank = in_slice ? sb.sLo + rS : global_rank_lt(ix.s_ord, 0, ix.n, qe[j]);
                if (qs[j] < qe[j]) {
                    c[j] = s_rank - (sb.eLo + rE);  // (qs < qe rules out qs == INT_MAX)
                } else {
                    int lo = first_pm_gt(ix.pm, ix.n, qs[j]);
                    int cc = 0;
                    for (int k = lo; k < s_rank; k++) cc += ix.e_ord[k] > qs[j];
                    c[j] = cc;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PC_ILP; j++) {
            const unsigned u = u0 + (unsigned)j * PT_THREADS;
            if (u < nch) {
                if (cb) store_count(cb, (int64_t)u, c[j]);
                acc += c[j];
            }
        }
    }
    if (total_slots) block_accumulate_i64(acc, red, total_slots + (blockIdx.x & (PT_SLOTS - 1)));
}

// ---- sorted batches: no bucketing at all ----
// When the query starts are already non-decreasing (the usual BED file), 16384 consecutive queries touch one short
// stretch of the sorted ends / starts.  One workgroup takes such a chunk as it lies: min/max of its keys (block
// reduction), the four slice boundaries (8-lane walks of the index's 32-ary trees by the first wave), the slices
// staged as LDS search trees exactly as in part_count_kernel, counts stored straight back in query order: 8 B read
// and 4 B written per query, no scratch.  Nothing in here relies on the order for correctness -- an unsorted chunk
// would just get long (sampled) slices and be slow -- the flag computed by part_hist_kernel only decides which of
// the two paths does the work.
#ifndef LC_THREADS_V
#define LC_THREADS_V 512
#endif
constexpr int LC_THREADS = LC_THREADS_V;
constexpr int LC_ITEMS = 8;
constexpr int LC_CHUNK = LC_THREADS * LC_ITEMS;  // 4096 consecutive queries per workgroup
#ifndef LC_WALK_BOTH
#define LC_WALK_BOTH 1  // the slice bounds of a chunk: both trees walked side by side (0: one after the other)
#endif
#ifndef LC_TREE_LOG2
#define LC_TREE_LOG2 12
#endif
constexpr int LC_TREE_KEYS = (1 << LC_TREE_LOG2) - 1;  // (round 5's layout: two Eytzinger trees of 4096 slots = 32 KiB of LDS, four workgroups per CU)
#ifndef LC_EXP
#define LC_EXP 0  // diagnostics (compile time, wrong results): bit 0 = no walks inside the cells, 1 = no interior table fill, 2 = no pop pass, 3 = no head / tail fill
#endif
// Round 6: the chunk's two slices lie in LDS as they are (sorted, an INT_MAX fence behind each) under a table of LC_NC equal
// coordinate cells -- cs[c] = keys in the cells below c -- so a rank is one table read and a walk of as many steps as the fullest
// cell's bit length (three or four at one key per cell) where the tree took twelve dependent reads: the searches were 300 of the
// kernel's 470 us on configs[4] (LC_EXP = 1).  The idea is this file's own direct addressing (cells_stage), built per chunk.
#ifndef LC_KEYS_V
#define LC_KEYS_V 4352
#endif
#ifndef LC_NC_LOG2_V
#define LC_NC_LOG2_V 10
#endif
#ifndef LC_WAVES_V
#define LC_WAVES_V 8
#endif
constexpr int LC_KEYS = LC_KEYS_V;                 // keys (or samples) of one staged slice, at most
constexpr int LC_NC_LOG2 = LC_NC_LOG2_V;
constexpr int LC_NC = (1 << LC_NC_LOG2) + (1 << (LC_NC_LOG2 - 3));  // cells per table (a map's width is rounded up to a power of two: the tail is room)
constexpr int LC_CS_INTS = (LC_NC + 2) / 2;        // one 16-bit table, in ints
constexpr int LC_LDS_INTS = 2 * (LC_KEYS + PC_PAD) + 2 * LC_CS_INTS;  // 39.4 KiB: four workgroups per CU (4608 keys + 2304 cells, three per CU: 398 us against 332)

// cells_stage for a chunk's TWO slices at once (LC_THREADS threads, LC_NC cells each): m = n / stride samples of each linearly with an
// INT_MAX fence behind them -- all loads of both slices in flight together -- then cs[c] = samples in the cells below c; the bit
// lengths of the two fullest cells (= the steps of a walk inside a cell) come back in stepsE / stepsS.  Four barriers
// (slice by slice, as cells_stage does it for a bucket image, the chunk paid ten and two round trips to HBM: 26 us per chunk).
__device__ __forceinline__ void lc_cells_stage2(int32_t *arrE, unsigned short *csE, const int32_t *__restrict__ srcE, int nE, int strideE, CellMap cmE,
                                                int32_t *arrS, unsigned short *csS, const int32_t *__restrict__ srcS, int nS, int strideS, CellMap cmS,
                                                int *s_red /* [2][LC_THREADS / 64] */, int &stepsE, int &stepsS)
{
    const int mE = nE / strideE, mS = nS / strideS;
    for (int r = threadIdx.x; r < mE; r += LC_THREADS) arrE[r] = srcE[(r + 1) * strideE - 1];
    for (int r = threadIdx.x; r < mS; r += LC_THREADS) arrS[r] = srcS[(r + 1) * strideS - 1];
    if (threadIdx.x < PC_PAD) arrE[mE + threadIdx.x] = INT_MAX, arrS[mS + threadIdx.x] = INT_MAX;
    __syncthreads();
    // element r opens every cell in (cell(arr[r-1]), cell(arr[r])]; the cells up to the first key's and behind the last key's --
    // a map's width is a power of two: up to half the table -- are filled by everybody (one thread walking them alone was
    // 10 us of every chunk)
    {
        const int firstE = mE ? cell_of(arrE[0], cmE) : LC_NC - 1, lastE = mE ? cell_of(arrE[mE - 1], cmE) : LC_NC - 1;
        const int firstS = mS ? cell_of(arrS[0], cmS) : LC_NC - 1, lastS = mS ? cell_of(arrS[mS - 1], cmS) : LC_NC - 1;
        for (int c = threadIdx.x; c < ((LC_EXP & 8) ? 0 : LC_NC); c += LC_THREADS) {
            if (c <= firstE) csE[c] = 0;
            else if (c > lastE) csE[c] = (unsigned short)mE;
            if (c <= firstS) csS[c] = 0;
            else if (c > lastS) csS[c] = (unsigned short)mS;
        }
    }
    // (measured and not kept: a thread's LDS reads of this pass issued together, the element before r from the neighbouring lane --
    // the registers spill and the walks pay: 335 -> 452 us for the kernel)
    for (int r = 1 + (int)threadIdx.x; r < ((LC_EXP & 2) ? 0 : mE); r += LC_THREADS) {
        const int cp = cell_of(arrE[r - 1], cmE), cr = cell_of(arrE[r], cmE);
        for (int c = cp + 1; c <= cr; c++) csE[c] = (unsigned short)r;
    }
    for (int r = 1 + (int)threadIdx.x; r < ((LC_EXP & 2) ? 0 : mS); r += LC_THREADS) {
        const int cp = cell_of(arrS[r - 1], cmS), cr = cell_of(arrS[r], cmS);
        for (int c = cp + 1; c <= cr; c++) csS[c] = (unsigned short)r;
    }
    __syncthreads();
    if (LC_EXP & 4) {
        stepsE = stepsS = 4;
        return;
    }
    int popE = 0, popS = 0;
    for (int c = threadIdx.x; c < LC_NC; c += LC_THREADS) {
        const int pE = (c + 1 < LC_NC ? (int)csE[c + 1] : mE) - (int)csE[c], pS = (c + 1 < LC_NC ? (int)csS[c + 1] : mS) - (int)csS[c];
        popE = pE > popE ? pE : popE, popS = pS > popS ? pS : popS;
    }
    popE = wave_max_i32(popE), popS = wave_max_i32(popS);
    if (lane_id() == 0) s_red[threadIdx.x >> 6] = popE, s_red[LC_THREADS / 64 + (threadIdx.x >> 6)] = popS;
    __syncthreads();
    popE = popS = 0;
#pragma unroll
    for (int i = 0; i < LC_THREADS / 64; i++) {
        popE = s_red[i] > popE ? s_red[i] : popE;
        popS = s_red[LC_THREADS / 64 + i] > popS ? s_red[LC_THREADS / 64 + i] : popS;
    }
    stepsE = 32 - __clz(popE), stepsS = 32 - __clz(popS);  // 0 for an empty slice
}

// One chunk of LC_CHUNK consecutive queries from `base` on: the workgroup's queries are k = j * LC_THREADS + thread, and
// emit(j, k, live, count, #{start < qe}, qs) is called once per (thread, j) with j a compile-time constant after unrolling.
template <typename Emit>
__device__ __forceinline__ void lc_chunk_counts(const TreeDev &S, const TreeDev &E, const IndexDev &ix, const int32_t *__restrict__ e_sorted,
                                                const int32_t *__restrict__ qs_arr, const int32_t *__restrict__ qe_arr, int64_t base, int n, int32_t *lds,
                                                int (*s_mm)[LC_THREADS / 64], int *s_slice, Emit emit)
{
    int qs[LC_ITEMS], qe[LC_ITEMS];
    int mn = INT_MAX, mx = INT_MIN, emx = INT_MIN;
#pragma unroll
    for (int j = 0; j < LC_ITEMS; j++) {
        int k = j * LC_THREADS + threadIdx.x;
        bool live = k < n;
        qs[j] = live ? qs_arr[base + k] : 0;
        qe[j] = live ? qe_arr[base + k] : 0;
        if (live) {
            mn = qs[j] < mn ? qs[j] : mn;
            mx = qs[j] > mx ? qs[j] : mx;
            emx = qe[j] > emx ? qe[j] : emx;
        }
    }
    mn = wave_min_i32(mn), mx = wave_max_i32(mx), emx = wave_max_i32(emx);
    if (lane_id() == 0) s_mm[0][threadIdx.x >> 6] = mn, s_mm[1][threadIdx.x >> 6] = mx, s_mm[2][threadIdx.x >> 6] = emx;
    __syncthreads();
    if (threadIdx.x < 64) {
        int a = INT_MAX, b = INT_MIN, c = INT_MIN;
#pragma unroll
        for (int i = 0; i < LC_THREADS / 64; i++) {
            a = s_mm[0][i] < a ? s_mm[0][i] : a;
            b = s_mm[1][i] > b ? s_mm[1][i] : b;
            c = s_mm[2][i] > c ? s_mm[2][i] : c;
        }
        // ends: #{end <= qs} for qs in [a, b] lies in [#{end <= a}, #{end <= b}].  starts: the keys qe of ordinary
        // queries lie in [a, max qe]; a lone far-away qe must not blow the slice up, so the key range is capped at
        // a few chunk spans and whatever falls outside takes a global search.
        long long cap = (long long)b + 4 * ((long long)b - (long long)a) + 65536;
        if (cap > INT_MAX) cap = INT_MAX;
        int s_hi_key = (long long)c < cap ? c : (int)cap;
        if (s_hi_key < a) s_hi_key = a;
        const int sub = threadIdx.x & 7, upper = (threadIdx.x >> 3) & 1;
        const int qs_key = upper ? b : a;
#if LC_WALK_BOTH
        {   // (both trees stand on n keys: the same depth) groups 0 / 1: the ends' bounds, 2 / 3: the starts', side by side
            const bool starts = ((threadIdx.x >> 4) & 1) != 0;
            const int key = starts ? (upper ? s_hi_key : a) : (qs_key == INT_MAX ? INT_MAX : qs_key + 1);
            int r = tree_rank_lt_either<true>(E, S, starts, key, sub);
            if (!starts && qs_key == INT_MAX) r = ix.n;  // every end is <= INT_MAX
            if (sub == 0 && threadIdx.x < 32) {
                s_slice[(starts ? 2 : 0) + upper] = r;
                if (starts) s_slice[4 + upper] = upper ? s_hi_key : a;
                if (!starts && upper) s_slice[6] = b;  // the chunk's largest qs: the ends' slice reaches that far
            }
        }
#else
        int keyE[1] = {qs_key == INT_MAX ? INT_MAX : qs_key + 1};
        int keyS[1] = {upper ? s_hi_key : a};
        int rE[1], rS[1];
        tree_rank_lt<true, 1>(E, lds, keyE, rE, sub);
        tree_rank_lt<true, 1>(S, lds, keyS, rS, sub);
        if (qs_key == INT_MAX) rE[0] = ix.n;  // every end is <= INT_MAX
        if (sub == 0 && threadIdx.x < 16) {
            s_slice[0 + upper] = rE[0];
            s_slice[2 + upper] = rS[0];
            s_slice[4 + upper] = upper ? s_hi_key : a;
            if (upper) s_slice[6] = b;
        }
#endif
    }
    __syncthreads();
    const int eLo = s_slice[0], eHi = s_slice[1], sLo = s_slice[2], sHi = s_slice[3], qeLo = s_slice[4], qeHi = s_slice[5], qsHi = s_slice[6];
    const int nE = eHi - eLo, nS = sHi - sLo;
    const int strideE = nE / LC_KEYS + 1, strideS = nS / LC_KEYS + 1;
    const int mE = nE / strideE, mS = nS / strideS;
    // the ends of the slice lie in (qeLo, qsHi], the starts in [qeLo, qeHi): one map each, cells of a power of two
    auto map_of = [](int lo, int hi) {
        CellMap cm;
        cm.lo = lo;
        const unsigned span = (unsigned)hi - (unsigned)lo;
        cm.cshift = span >> LC_NC_LOG2 ? 32 - __clz(span >> LC_NC_LOG2) : 0;
        const long long top = (long long)lo + ((long long)LC_NC << cm.cshift) - 1;
        cm.hi = top > INT_MAX ? INT_MAX : (int)top;
        return cm;
    };
    const CellMap cmE = map_of(qeLo, qsHi), cmS = map_of(qeLo, qeHi);
    unsigned short *csE = reinterpret_cast<unsigned short *>(lds), *csS = reinterpret_cast<unsigned short *>(lds + LC_CS_INTS);
    int32_t *arrE = lds + 2 * LC_CS_INTS, *arrS = arrE + mE + PC_PAD;
    int stepsE, stepsS;
    lc_cells_stage2(arrE, csE, e_sorted + eLo, nE, strideE, cmE, arrS, csS, ix.s_ord + sLo, nS, strideS, cmS, &s_mm[0][0], stepsE, stepsS);
    const lds_i32p aE = (lds_i32p)arrE, aS = (lds_i32p)arrS;
    const lds_u16p cE = (lds_u16p)csE, cS = (lds_u16p)csS;
    const bool fenced = stepsE <= 6 && stepsS <= 6;  // every probe stays inside the INT_MAX fence
    const bool unsampled = strideS == 1 && strideE == 1;
    const int cconst = (sLo - eLo) - (int)(aS - aE);
    const unsigned qe_span = (unsigned)qeHi - (unsigned)qeLo;
#pragma unroll
    for (int j0 = 0; j0 < LC_ITEMS; j0 += PT_ILP) {
        lds_i32p pS[PT_ILP], pE[PT_ILP];
#pragma unroll
        for (int j = 0; j < PT_ILP; j++) {
            pS[j] = aS + cS[cell_of(qe[j0 + j], cmS)] - 1;  // the last key known to be below the probe
            pE[j] = aE + cE[cell_of(qs[j0 + j], cmE)] - 1;
        }
        if ((LC_EXP & 1) == 0) {
            // only keys of the probe's own cell can still qualify, everything in later cells is larger, the fence stops the walk
            if (fenced) {
                for (int st = stepsS - 1; st >= 0; st--) {
#pragma unroll
                    for (int j = 0; j < PT_ILP; j++) {
                        const lds_i32p t = pS[j] + (1 << st);
                        pS[j] = *t < qe[j0 + j] ? t : pS[j];
                    }
                }
                for (int st = stepsE - 1; st >= 0; st--) {
#pragma unroll
                    for (int j = 0; j < PT_ILP; j++) {
                        const lds_i32p t = pE[j] + (1 << st);
                        pE[j] = *t <= qs[j0 + j] ? t : pE[j];  // (qs == INT_MAX passes the fence: handled below)
                    }
                }
            } else {
                const lds_i32p endS = aS + mS, endE = aE + mE;
                for (int st = stepsS - 1; st >= 0; st--) {
#pragma unroll
                    for (int j = 0; j < PT_ILP; j++) {
                        lds_i32p t = pS[j] + (1 << st);
                        t = t < endS ? t : endS;
                        pS[j] = *t < qe[j0 + j] ? t : pS[j];
                    }
                }
                for (int st = stepsE - 1; st >= 0; st--) {
#pragma unroll
                    for (int j = 0; j < PT_ILP; j++) {
                        lds_i32p t = pE[j] + (1 << st);
                        t = t < endE ? t : endE;
                        pE[j] = *t <= qs[j0 + j] ? t : pE[j];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PT_ILP; j++) {
            const int k = (j0 + j) * LC_THREADS + threadIdx.x;
            const bool live = k < n;
            const int s = qs[j0 + j], e = qe[j0 + j];
            int rS = (int)(pS[j] - aS) + 1, rE = (int)(pE[j] - aE) + 1;  // staged keys (samples) below the probe
            int c = (int)(pS[j] - pE[j]) + cconst;                        // (sLo + #starts < qe) - (eLo + #ends <= qs)
            int s_rank = sLo + rS;
            const bool in_slice = (unsigned)e - (unsigned)qeLo <= qe_span;
            if (live && !(unsampled && s < e && in_slice)) {
                // sampled slices: each rank finished inside its group; qe outside the slice: a global search;
                // a zero-length / reversed query: the exact predicate over the candidate window
                rS *= strideS, rE *= strideE;
                if (strideS > 1) rS = group_rank_lt(ix.s_ord + sLo, rS, rS + strideS < nS ? rS + strideS : nS, e);
                if (strideE > 1 && s != INT_MAX) rE = group_rank_lt(e_sorted + eLo, rE, rE + strideE < nE ? rE + strideE : nE, s + 1);
                s_rank = in_slice ? sLo + rS : global_rank_lt(ix.s_ord, 0, ix.n, e);
                if (s < e) {
                    c = s_rank - (eLo + rE);  // (s < e rules out s == INT_MAX)
                } else {
                    const int lo = first_pm_gt(ix.pm, ix.n, s);
                    c = 0;
                    for (int t = lo; t < s_rank; t++) c += ix.e_ord[t] > s;
                }
            }
            if (!live) c = 0, s_rank = 0;
            if (LC_EXP & 1) c = 0, s_rank = sLo;  // (diagnostics: nothing found, every rank inside the index)
            emit(j0 + j, k, live, c, s_rank, s);
        }
    }
}

// (eight waves per SIMD = four workgroups per CU: the kernel lives on the chunks it keeps in flight; 62 registers)
__global__ __launch_bounds__(LC_THREADS) __attribute__((amdgpu_waves_per_eu(LC_WAVES_V))) void ivl_local_count_kernel(TreeDev S, TreeDev E, IndexDev ix, const int32_t *__restrict__ e_sorted,
                                                                     const int32_t *__restrict__ qs_arr,
                                                                     const int32_t *__restrict__ qe_arr, int64_t nq,
                                                                     int32_t *__restrict__ counts /* may be NULL */,
                                                                     unsigned long long *__restrict__ total_slots,
                                                                     const unsigned *__restrict__ gate,
                                                                     int32_t *__restrict__ his = nullptr /* find(): #{start < qe} of every query */,
                                                                     unsigned long long *__restrict__ order_host = nullptr, unsigned long long seq = 0,
                                                                     unsigned long long *__restrict__ chunk_tot = nullptr /* find(): the sum of every chunk's counts */)
{
    __shared__ __attribute__((aligned(16))) int32_t lds[LC_LDS_INTS];
    __shared__ int s_mm[3][LC_THREADS / 64];
    __shared__ int s_slice[8];  // eLo, eHi, sLo, sHi, qeLo, qeHi, the largest qs
    __shared__ long long red[LC_THREADS / 64];
    __shared__ long long red2[LC_THREADS / 64];
    // what the order check found, into host memory: the host picks the shape of THIS kernel for later batches by it
    // (bm_count_segments; pass number << 1 | 1 = not sorted)
    if (order_host && blockIdx.x == 0 && threadIdx.x == 0) *order_host = (seq << 1) | (gate && *gate != 0 ? 1ull : 0ull);
    if (gate && *gate != 0) return;  // unsorted batch: the bucketed path answers it
    long long acc = 0;
    {
        const int64_t chunk = blockIdx.x;
        const int64_t base = chunk * LC_CHUNK;
        const int n = (int)(nq - base < LC_CHUNK ? nq - base : LC_CHUNK);
        long long cacc = 0;
        lc_chunk_counts(S, E, ix, e_sorted, qs_arr, qe_arr, base, n, lds, s_mm, s_slice, [&](int, int k, bool live, int c, int s_rank, int) {
            if (!live) return;
            if (counts) counts[base + k] = c;
            if (his) his[base + k] = s_rank;
            cacc += c;
        });
        acc += cacc;
        if (chunk_tot) {  // (find(): the CSR offsets are then one scan over the CHUNKS away, ivl_find_local)
            const long long w = wave_sum_i64(cacc);
            if (lane_id() == 0) red2[threadIdx.x >> 6] = w;
            __syncthreads();
            if (threadIdx.x == 0) {
                long long t = 0;
                for (int i = 0; i < LC_THREADS / 64; i++) t += red2[i];
                chunk_tot[chunk] = (unsigned long long)t;
            }
        }
    }
    if (total_slots) block_accumulate_i64(acc, red, total_slots + (blockIdx.x & (PT_SLOTS - 1)));
}

// Counts come back in bucket order.  One workgroup per partition tile pulls the tile's runs
// (one per bucket, contiguous in the bucketed array) into LDS in the tile's sorted order, then
// every query picks its count through the 16-bit slot remembered by the scatter: all global
// traffic is coalesced, the random access happens in LDS.
template <typename CT /* int32_t, or unsigned short with COUNT_ESCAPE */>
__global__ __launch_bounds__(PT_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void part_gather_kernel(const CT *__restrict__ bucketed,
                                                                 const unsigned short *__restrict__ lpos,
                                                                 const unsigned *__restrict__ tile_table /* [ntiles][PT_NB] */,
                                                                 int64_t ntiles, int64_t nq, int32_t *__restrict__ out,
                                                                 const unsigned *__restrict__ gate, IndexDev ix,
                                                                 const int32_t *__restrict__ e_sorted, const int32_t *__restrict__ qs_arr,
                                                                 const int32_t *__restrict__ qe_arr /* the four: escape path only */)
{
    __shared__ CT vals[PT_TILE];
    __shared__ unsigned short toff[PT_NB + 2];
    __shared__ unsigned gbase[PT_NB];
    __shared__ unsigned scan_tmp[16];
    const int64_t tile = part_tile_of_block(ntiles);
    if (tile >= ntiles || (gate && *gate == 0)) return;
    const int64_t base = tile * PT_TILE;
    const int n = (int)(nq - base < PT_TILE ? nq - base : PT_TILE);
    {
        // Tile counts = distance to the next entry of the (linear, bucket-major) exclusive scan: the next
        // tile's entry for the same bucket, or -- for the last tile -- tile 0's entry of the next bucket.
        const bool last_tile = tile + 1 == ntiles;
        const unsigned *row = tile_table + tile * PT_NB;
        const unsigned *next = last_tile ? tile_table : row + PT_NB;
        unsigned c[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            int b = 2 * threadIdx.x + u;
            unsigned lo = row[b];
            unsigned hi = !last_tile ? next[b] : (b + 1 < PT_NB ? next[b + 1] : (unsigned)nq);
            gbase[b] = lo;
            c[u] = hi - lo;
        }
        unsigned tot;
        unsigned exc = block_exclusive_scan(c[0] + c[1], OpSum(), 0u, scan_tmp, &tot);
        toff[2 * threadIdx.x] = (unsigned short)exc;
        toff[2 * threadIdx.x + 1] = (unsigned short)(exc + c[0]);
        if (threadIdx.x == 0) toff[PT_NB] = (unsigned short)tot;  // tot == n <= 16384
    }
    __syncthreads();
    // 8 lanes per bucket run (runs average 8 queries).  A lane's 16 runs are handled eight at a time with all loads
    // of a round issued before the first LDS write: the loop "per run: load, store" is one dependent round trip per
    // run (measured 29 us per tile, nearly all of it latency).
    const unsigned sub = threadIdx.x & 7;
    constexpr int RUNS = PT_NB / (PT_THREADS / 8);  // 16 runs per lane
#pragma unroll
    for (int round = 0; round < 2; round++) {  // elements sub and sub + 8 of all 16 runs: two round trips in all
        const unsigned r = sub + 8u * round;
        CT v[RUNS];
        unsigned short at[RUNS];
        unsigned live = 0;
#pragma unroll
        for (int i = 0; i < RUNS; i++) {
            const int b = (int)(threadIdx.x >> 3) + i * (PT_THREADS / 8);
            const unsigned o = toff[b], len = (b + 1 < PT_NB ? toff[b + 1] : (unsigned)n) - o;
            const bool ok = r < len;
            live |= (unsigned)ok << i;
            at[i] = (unsigned short)(o + r);
            v[i] = ok ? bucketed[gbase[b] + r] : (CT)0;
        }
#pragma unroll
        for (int i = 0; i < RUNS; i++)
            if (live >> i & 1) vals[at[i]] = v[i];
    }
#pragma unroll 1
    for (int i = 0; i < RUNS; i++) {  // runs longer than 16 (rare; the whole tile for a sorted batch): the whole wave copies them
        const int b = (int)(threadIdx.x >> 3) + i * (PT_THREADS / 8);
        const unsigned o = toff[b], len = (b + 1 < PT_NB ? toff[b + 1] : (unsigned)n) - o, gb = gbase[b];
        unsigned long long m = __ballot(sub == 0 && len > 16);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const unsigned oo = __shfl(o, src, 64), ll = __shfl(len, src, 64), gg = __shfl(gb, src, 64);
            for (unsigned q = 16 + lane_id(); q < ll; q += 64) vals[oo + q] = bucketed[gg + q];
        }
    }
    __syncthreads();
    constexpr bool ESC = sizeof(CT) == 2;
    if (n == PT_TILE) {
        // a lane takes 4 consecutive queries: 8-byte loads of the slots, 16-byte stores of the counts, all loads first
        const uint2 *l4 = reinterpret_cast<const uint2 *>(lpos + base);
        int4 *o4 = reinterpret_cast<int4 *>(out + base);
        uint2 sl[PT_ITEMS / 4];
#pragma unroll
        for (int j = 0; j < PT_ITEMS / 4; j++) sl[j] = l4[j * PT_THREADS + threadIdx.x];
#pragma unroll
        for (int j = 0; j < PT_ITEMS / 4; j++) {
            int c[4] = {(int)vals[sl[j].x & 0xffffu], (int)vals[sl[j].x >> 16], (int)vals[sl[j].y & 0xffffu], (int)vals[sl[j].y >> 16]};
            if (ESC && ((unsigned)c[0] == COUNT_ESCAPE || (unsigned)c[1] == COUNT_ESCAPE || (unsigned)c[2] == COUNT_ESCAPE ||
                        (unsigned)c[3] == COUNT_ESCAPE)) {
                const int64_t k0 = base + 4 * (int64_t)(j * PT_THREADS + threadIdx.x);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if ((unsigned)c[u] == COUNT_ESCAPE) c[u] = count_one_global(ix, e_sorted, qs_arr[k0 + u], qe_arr[k0 + u]);
            }
            o4[j * PT_THREADS + threadIdx.x] = make_int4(c[0], c[1], c[2], c[3]);
        }
    } else {
        for (int k = threadIdx.x; k < n; k += PT_THREADS) {
            int c = (int)vals[lpos[base + k]];
            if (ESC && (unsigned)c == COUNT_ESCAPE) c = count_one_global(ix, e_sorted, qs_arr[base + k], qe_arr[base + k]);
            out[base + k] = c;
        }
    }
}

__device__ __forceinline__ int count_one_global(const IndexDev &ix, const int32_t *__restrict__ e_sorted, int qs, int qe)
{
    if (qs < qe) {  // (qs < qe rules out qs == INT_MAX)
        // #{start < qe} - #{end <= qs}: both searches step together (the arrays have one length), two independent loads per step --
        // a lone escape is a chain of dependent loads from HBM, and a total-only pass waits for the slowest one (bm_escape_totals_kernel)
        const int n = ix.n, keyE = qs + 1;
        int pS = -1, pE = -1;  // the last element below the key
        for (int step = n > 0 ? 1 << (31 - __clz(n)) : 0; step > 0; step >>= 1) {
            const int nS = pS + step, nE = pE + step;
            const int vS = nS < n ? ix.s_ord[nS] : INT_MAX, vE = nE < n ? e_sorted[nE] : INT_MAX;
            if (nS < n && vS < qe) pS = nS;
            if (nE < n && vE < keyE) pE = nE;
        }
        return pS - pE;
    }
    const int s_rank = global_rank_lt(ix.s_ord, 0, ix.n, qe);
    const int lo = first_pm_gt(ix.pm, ix.n, qs);  // zero-length / reversed query: exact predicate over the candidate window
    int c = 0;
    for (int k = lo; k < s_rank; k++) c += ix.e_ord[k] > qs;
    return c;
}

}  // namespace bxmi
