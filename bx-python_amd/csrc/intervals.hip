// intervals.hip -- the interval index behind IntervalTree/Intersecter.find().
//
// Reference algorithm (lib/bx/intervals/intersection.pyx): a randomised treap
// of Python nodes, find() = pointer-chasing DFS (:180-189) that reports, in
// in-order, every interval with  end > qs  and  start < qe.
//
// MI355X design (not a treap):
//   * seal(): radix-sort the intervals into the treap's in-order, which is the
//     sort by (start, end<=start first, -i / +i) (intersection.pyx:112-116);
//     keep SoA int32 arrays in that order: s_ord, e_ord, idx, plus the prefix
//     max of e_ord (pm) and the separately sorted ends (e_sorted).
//   * every rank query goes through a static 32-ary search tree: a node is
//     32 int32 keys = one 128-byte line; 8 lanes cooperate on a node (one
//     coalesced 16-byte load each, compare 4 keys, 3 DPP adds).  The top
//     levels are staged in LDS, the lower ones come from L2 / HBM.  A 10M
//     index is 5 levels deep: 3 LDS visits + 2 line fetches per rank.
//   * count = rank_lt(starts, qe) - rank_le(ends, qs) for proper queries on
//     proper targets; anything else (zero-length / reversed query, reversed
//     target) takes the exact window scan [first pm>qs, rank_lt(starts,qe)).
//   * find = the same window, compacted with wave ballots into CSR order.
//   * batches of >= 4 Mi queries leave the trees: the queries are bucketed by coordinate (histogram
//     + LDS-ordered scatter, "part_*" kernels), every bucket is searched against its slice of the
//     sorted arrays held in LDS by direct addressing, and the counts are gathered back; a batch
//     whose starts are already sorted skips the bucketing ("ivl_local_*" kernels).
//   * clusters (ClusterTree) fall out of s_ord and pm: a boundary wherever start - d > pm[i-1].
// Integer compares and popcounts only: HBM / LDS bound, no MFMA.
//
// Map of the file: build kernels and tree search helpers; direct count kernel; partitioned count
// path (histogram, column scans, scatter, tree and cell searches, sorted-batch kernel, gather);
// partitioned / sorted find (window, permute, fill kernels); cluster kernels; host structs and the
// extern "C" entry points (bxmi_ivl_*), DESIGN.md 3 has the measurements.
#include <climits>
#include <vector>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "primitives.hpp"

namespace bxmi {

constexpr int MAXLEV = 7;          // 32^7 > 2^31
constexpr int FAN = 32;            // keys per node (128 B)
constexpr int LDS_TREE_INTS = 18688;  // per tree: 73 KiB, two trees + scratch < 160 KiB
constexpr int CNT_THREADS = 1024;  // one workgroup per CU, 16 waves
constexpr int CNT_Q = 4;           // queries in flight per 8-lane group
constexpr int FIND_THREADS = 512;
constexpr int FIND_Q = 2;

struct TreeDev {
    const int32_t *lev[MAXLEV];  // lev[0] = leaves (the sorted array, padded with INT_MAX)
    int32_t lds_off[MAXLEV];     // offset (ints) of the level inside this tree's LDS region
    int32_t lev_ints[MAXLEV];    // ints in the level (32 * nodes)
    int32_t nlev;
    int32_t lds_from;            // levels >= lds_from live in LDS
    int32_t lds_ints;            // total ints staged
};

struct IndexDev {
    const int32_t *s_ord, *e_ord, *idx, *pm;
    int32_t n;
    int32_t has_reversed;
};

// ---------------------------------------------------------------------------
// build kernels
// ---------------------------------------------------------------------------
// 64-bit sort key: biased start in the high word, then the tie rule of
// intersection.pyx:112-116 -- on equal starts, intervals with end <= start go
// LEFT (so they come first, newest first), the others go right (oldest first).
__global__ void ivl_make_keys_kernel(const int32_t *__restrict__ start, const int32_t *__restrict__ end, int64_t n,
                                     unsigned long long *__restrict__ keys, uint32_t *__restrict__ end_keys,
                                     unsigned *__restrict__ n_reversed)
{
    unsigned rev = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int32_t s = start[i], e = end[i];
        uint32_t sub = (e <= s) ? (0x7fffffffu - (uint32_t)i) : (0x80000000u | (uint32_t)i);
        keys[i] = ((unsigned long long)((uint32_t)s ^ 0x80000000u) << 32) | sub;
        end_keys[i] = (uint32_t)e ^ 0x80000000u;
        rev += (e < s);
    }
    unsigned long long m = __ballot(rev != 0);
    if (m && lane_id() == (int)__ffsll((long long)m) - 1) {
        // one atomic per wave is plenty: we only need "zero or not"
        atomicAdd(n_reversed, 1u);
    }
}

__global__ void ivl_unpack_kernel(const unsigned long long *__restrict__ keys, const int32_t *__restrict__ end,
                                  int64_t n, int64_t n_pad, int32_t *__restrict__ s_ord, int32_t *__restrict__ e_ord,
                                  int32_t *__restrict__ idx)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_pad; k += (int64_t)gridDim.x * blockDim.x) {
        if (k < n) {
            unsigned long long key = keys[k];
            uint32_t sub = (uint32_t)key;
            int32_t i = (sub & 0x80000000u) ? (int32_t)(sub & 0x7fffffffu) : (int32_t)(0x7fffffffu - sub);
            s_ord[k] = (int32_t)((uint32_t)(key >> 32) ^ 0x80000000u);
            idx[k] = i;
            e_ord[k] = end[i];
        } else {  // padding up to a whole node
            s_ord[k] = INT_MAX;
            e_ord[k] = INT_MIN;
            idx[k] = -1;
        }
    }
}

__global__ void ivl_unbias_kernel(const uint32_t *__restrict__ in, int64_t n, int64_t n_pad, int32_t *__restrict__ out)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_pad; k += (int64_t)gridDim.x * blockDim.x)
        out[k] = k < n ? (int32_t)(in[k] ^ 0x80000000u) : INT_MAX;
}

__global__ void ivl_pad_kernel(int32_t *__restrict__ a, int64_t n, int64_t n_pad, int32_t v)
{
    int64_t k = n + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_pad) a[k] = v;
}

// Level l+1 of a search tree: entry j = last key of child node j, except that
// the LAST child (and every padding slot) gets INT_MAX, a fence no key is
// greater than.  With the fence, "number of entries < key" is always a valid
// child index and no clamping is needed on the way down.
__global__ void ivl_tree_level_kernel(const int32_t *__restrict__ below, int64_t nodes_below,
                                      int32_t *__restrict__ out, int64_t out_ints)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < out_ints; j += (int64_t)gridDim.x * blockDim.x)
        out[j] = (j < nodes_below - 1) ? below[j * FAN + (FAN - 1)] : INT_MAX;
}

// ---------------------------------------------------------------------------
// device-side search
// ---------------------------------------------------------------------------
template <bool DPP>
__device__ __forceinline__ int node_count_lt(int4 v, int key)
{
    int c = (v.x < key) + (v.y < key) + (v.z < key) + (v.w < key);
    return DPP ? group8_sum_dpp(c) : group8_sum_shfl(c);
}

__device__ __forceinline__ void stage_tree(const TreeDev &t, int32_t *lds)
{
    for (int l = t.nlev - 1; l >= t.lds_from; --l) {
        const int4 *src = reinterpret_cast<const int4 *>(t.lev[l]);
        int4 *dst = reinterpret_cast<int4 *>(lds + t.lds_off[l]);
        int n4 = t.lev_ints[l] >> 2;
        for (int i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = src[i];
    }
}

// rank_lt for NQ independent keys at once (ILP): returns #{a[i] < key}.
// Two separate loops so the staged levels compile to ds_read_b128 and the lower
// ones to global_load_dwordx4 (one merged loop makes hipcc fall back to flat_load
// with a full vmcnt+lgkmcnt drain per level).
template <bool DPP, int NQ>
__device__ __forceinline__ void tree_rank_lt(const TreeDev &t, const int32_t *lds, const int (&key)[NQ], int (&rank)[NQ],
                                             int sub)
{
#pragma unroll
    for (int j = 0; j < NQ; j++) rank[j] = 0;
    int l = t.nlev - 1;
    for (; l >= t.lds_from; --l) {
        const int4 *b = reinterpret_cast<const int4 *>(lds + t.lds_off[l]) + sub;
        int4 v[NQ];
#pragma unroll
        for (int j = 0; j < NQ; j++) v[j] = b[rank[j] * (FAN / 4)];
#pragma unroll
        for (int j = 0; j < NQ; j++) rank[j] = rank[j] * FAN + node_count_lt<DPP>(v[j], key[j]);
    }
    for (; l >= 0; --l) {
        const int4 *b = reinterpret_cast<const int4 *>(t.lev[l]) + sub;
        int4 v[NQ];
#pragma unroll
        for (int j = 0; j < NQ; j++) v[j] = b[(int64_t)rank[j] * (FAN / 4)];
#pragma unroll
        for (int j = 0; j < NQ; j++) rank[j] = rank[j] * FAN + node_count_lt<DPP>(v[j], key[j]);
    }
}

// One key per 8-lane group, each group walking ONE of two trees of the same depth (global levels only): the walks of the
// sorted-batch kernels' slice bounds -- two keys in each of two trees -- are four groups of one wave side by side, instead of
// one tree after the other (a chunk's set-up is a chain of dependent loads: twelve of them became six).
template <bool DPP>
__device__ __forceinline__ int tree_rank_lt_either(const TreeDev &a, const TreeDev &b, bool use_b, int key, int sub)
{
    int rank = 0;
    for (int l = a.nlev - 1; l >= 0; --l) {  // (a.nlev == b.nlev: the caller's business)
        // (both pointers as scalars first: a select between the two STRUCTS' members sends the structs to scratch memory)
        const int32_t *pa = a.lev[l], *pb = b.lev[l];
        const int4 *lev = reinterpret_cast<const int4 *>(use_b ? pb : pa) + sub;
        const int4 v = lev[(int64_t)rank * (FAN / 4)];
        rank = rank * FAN + node_count_lt<DPP>(v, key);
    }
    return rank;
}

// Plain lower bound on the monotone prefix-max array: first k with pm[k] > qs.
__device__ __forceinline__ int first_pm_gt(const int32_t *__restrict__ pm, int n, int qs)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if (pm[mid] > qs)
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo;
}

// #{k in [lo,hi) : e_ord[k] > qs}, 8 lanes x int4 per 32-element step, aligned.
template <bool DPP>
__device__ __forceinline__ int window_count(const int32_t *__restrict__ e_ord, int lo, int hi, int qs, int sub)
{
    int c = 0;
    for (int k0 = lo & ~(FAN - 1); k0 < hi; k0 += FAN) {
        int kb = k0 + sub * 4;
        int4 v = *reinterpret_cast<const int4 *>(e_ord + kb);
        c += (kb + 0 >= lo && kb + 0 < hi && v.x > qs);
        c += (kb + 1 >= lo && kb + 1 < hi && v.y > qs);
        c += (kb + 2 >= lo && kb + 2 < hi && v.z > qs);
        c += (kb + 3 >= lo && kb + 3 < hi && v.w > qs);
    }
    return DPP ? group8_sum_dpp(c) : group8_sum_shfl(c);
}

// ---------------------------------------------------------------------------
// count kernel (the headline path: 100M queries x 10M targets)
// ---------------------------------------------------------------------------
template <bool DPP>
__global__ __launch_bounds__(CNT_THREADS) void ivl_count_kernel(TreeDev S, TreeDev E, IndexDev ix,
                                                               const int32_t *__restrict__ qs_arr,
                                                               const int32_t *__restrict__ qe_arr, int64_t nq,
                                                               int32_t *__restrict__ counts,
                                                               unsigned long long *__restrict__ total)
{
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    int32_t *ldsS = lds, *ldsE = lds + S.lds_ints;
    long long *red = reinterpret_cast<long long *>(lds + S.lds_ints + E.lds_ints);  // 16 slots after the staged levels
    stage_tree(S, ldsS);
    stage_tree(E, ldsE);
    __syncthreads();

    const int sub = threadIdx.x & 7;
    const int64_t group = (int64_t)blockIdx.x * (CNT_THREADS / 8) + (threadIdx.x >> 3);
    const int64_t ngroups = (int64_t)gridDim.x * (CNT_THREADS / 8);
    long long acc = 0;

    for (int64_t q0 = group * CNT_Q; q0 < nq; q0 += ngroups * CNT_Q) {
        int qs[CNT_Q], qe[CNT_Q];
        if (q0 + CNT_Q <= nq) {
            int4 a = *reinterpret_cast<const int4 *>(qs_arr + q0);
            int4 b = *reinterpret_cast<const int4 *>(qe_arr + q0);
            qs[0] = a.x, qs[1] = a.y, qs[2] = a.z, qs[3] = a.w;
            qe[0] = b.x, qe[1] = b.y, qe[2] = b.z, qe[3] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < CNT_Q; j++) {
                bool ok = q0 + j < nq;
                qs[j] = ok ? qs_arr[q0 + j] : 0;
                qe[j] = ok ? qe_arr[q0 + j] : 0;  // (0,0): zero-length, handled below, result discarded
            }
        }
        // rank_lt(starts, qe) and rank_le(ends, qs) = rank_lt(ends, qs+1)
        int kE[CNT_Q], rS[CNT_Q], rE[CNT_Q];
#pragma unroll
        for (int j = 0; j < CNT_Q; j++) kE[j] = qs[j] == INT_MAX ? INT_MAX : qs[j] + 1;
        tree_rank_lt<DPP, CNT_Q>(S, ldsS, qe, rS, sub);
        tree_rank_lt<DPP, CNT_Q>(E, ldsE, kE, rE, sub);

        int cnt[CNT_Q];
#pragma unroll
        for (int j = 0; j < CNT_Q; j++) {
            bool regular = (qs[j] < qe[j]) && !ix.has_reversed;
            if (regular) {
                cnt[j] = rS[j] - (qs[j] == INT_MAX ? ix.n : rE[j]);
            } else {
                // exact predicate over the candidate window (uniform inside the 8-lane group)
                int lo = first_pm_gt(ix.pm, ix.n, qs[j]);
                cnt[j] = lo < rS[j] ? window_count<DPP>(ix.e_ord, lo, rS[j], qs[j], sub) : 0;
            }
        }
        if (sub == 0) {
            if (q0 + CNT_Q <= nq) {
                if (counts) *reinterpret_cast<int4 *>(counts + q0) = make_int4(cnt[0], cnt[1], cnt[2], cnt[3]);
                acc += (long long)cnt[0] + cnt[1] + cnt[2] + cnt[3];
            } else {
#pragma unroll
                for (int j = 0; j < CNT_Q; j++)
                    if (q0 + j < nq) {
                        if (counts) counts[q0 + j] = cnt[j];
                        acc += cnt[j];
                    }
            }
        }
    }
    if (total) block_accumulate_i64(acc, red, total);
}


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
constexpr int LC_TREE_KEYS = (1 << LC_TREE_LOG2) - 1;  // two trees of 4096 slots = 32 KiB of LDS: four workgroups per CU

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
        }
#endif
    }
    __syncthreads();
    const int eLo = s_slice[0], eHi = s_slice[1], sLo = s_slice[2], sHi = s_slice[3], qeLo = s_slice[4], qeHi = s_slice[5];
    const int nE = eHi - eLo, nS = sHi - sLo;
    const int strideE = nE / LC_TREE_KEYS + 1, strideS = nS / LC_TREE_KEYS + 1;
    int kE = 0, kS = 0;
    while ((1 << kE) - 1 < nE / strideE) kE++;
    while ((1 << kS) - 1 < nS / strideS) kS++;
    int32_t *treeE = lds, *treeS = lds + (1 << kE);
    {
        const int total = (1 << kE) + (1 << kS);
        for (int i = threadIdx.x; i < total; i += LC_THREADS) lds[i] = INT_MAX;
        __syncthreads();
        part_stage_tree<LC_THREADS>(treeE, kE, e_sorted + eLo, nE, strideE);
        part_stage_tree<LC_THREADS>(treeS, kS, ix.s_ord + sLo, nS, strideS);
    }
    __syncthreads();
#pragma unroll
    for (int j0 = 0; j0 < LC_ITEMS; j0 += PT_ILP) {
        int rS[PT_ILP], rE[PT_ILP];
#pragma unroll
        for (int j = 0; j < PT_ILP; j++) rS[j] = rE[j] = 1;
        for (int it = 0; it < kS; it++) {
#pragma unroll
            for (int j = 0; j < PT_ILP; j++) rS[j] = 2 * rS[j] + (treeS[rS[j]] < qe[j0 + j]);
        }
        for (int it = 0; it < kE; it++) {
#pragma unroll
            for (int j = 0; j < PT_ILP; j++) rE[j] = 2 * rE[j] + (treeE[rE[j]] <= qs[j0 + j] && qs[j0 + j] != INT_MAX);
        }
#pragma unroll
        for (int j = 0; j < PT_ILP; j++) {
            rS[j] = (rS[j] - (1 << kS)) * strideS;
            rE[j] = (rE[j] - (1 << kE)) * strideE;
        }
        if (strideS > 1) {
#pragma unroll
            for (int j = 0; j < PT_ILP; j++) {
                int hi = rS[j] + strideS < nS ? rS[j] + strideS : nS;
                rS[j] = group_rank_lt(ix.s_ord + sLo, rS[j], hi, qe[j0 + j]);
            }
        }
        if (strideE > 1) {
#pragma unroll
            for (int j = 0; j < PT_ILP; j++) {
                int hi = rE[j] + strideE < nE ? rE[j] + strideE : nE;
                rE[j] = qs[j0 + j] == INT_MAX ? 0 : group_rank_lt(e_sorted + eLo, rE[j], hi, qs[j0 + j] + 1);
            }
        }
#pragma unroll
        for (int j = 0; j < PT_ILP; j++) {
            const int k = (j0 + j) * LC_THREADS + threadIdx.x;
            const bool live = k < n;
            int c = 0, s_rank = 0;
            const int s = qs[j0 + j], e = qe[j0 + j];
            if (live) {
                const bool in_slice = e >= qeLo && e <= qeHi;
                s_rank = in_slice ? sLo + rS[j] : global_rank_lt(ix.s_ord, 0, ix.n, e);
                if (s < e) {
                    const int e_rank = s == INT_MAX ? ix.n : eLo + rE[j];
                    c = s_rank - e_rank;
                } else {  // zero-length / reversed query: exact predicate over the candidate window
                    int lo = first_pm_gt(ix.pm, ix.n, s);
                    for (int t = lo; t < s_rank; t++) c += ix.e_ord[t] > s;
                }
            }
            emit(j0 + j, k, live, c, s_rank, s);
        }
    }
}

// (eight waves per SIMD = four workgroups per CU: the kernel lives on the chunks it keeps in flight -- said out loud, the compiler
// took 70 registers for a build that needed 64)
__global__ __launch_bounds__(LC_THREADS) __attribute__((amdgpu_waves_per_eu(8))) void ivl_local_count_kernel(TreeDev S, TreeDev E, IndexDev ix, const int32_t *__restrict__ e_sorted,
                                                                     const int32_t *__restrict__ qs_arr,
                                                                     const int32_t *__restrict__ qe_arr, int64_t nq,
                                                                     int32_t *__restrict__ counts /* may be NULL */,
                                                                     unsigned long long *__restrict__ total_slots,
                                                                     const unsigned *__restrict__ gate,
                                                                     int32_t *__restrict__ his = nullptr /* find(): #{start < qe} of every query */,
                                                                     unsigned long long *__restrict__ order_host = nullptr, unsigned long long seq = 0,
                                                                     unsigned long long *__restrict__ chunk_tot = nullptr /* find(): the sum of every chunk's counts */)
{
    __shared__ __attribute__((aligned(16))) int32_t lds[2 * (LC_TREE_KEYS + 1)];
    __shared__ int s_mm[3][LC_THREADS / 64];
    __shared__ int s_slice[6];  // eLo, eHi, sLo, sHi, qeLo, qeHi
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
    const int s_rank = global_rank_lt(ix.s_ord, 0, ix.n, qe);
    if (qs < qe) return s_rank - global_rank_lt(e_sorted, 0, ix.n, qs + 1);  // (qs < qe rules out qs == INT_MAX)
    const int lo = first_pm_gt(ix.pm, ix.n, qs);  // zero-length / reversed query: exact predicate over the candidate window
    int c = 0;
    for (int k = lo; k < s_rank; k++) c += ix.e_ord[k] > qs;
    return c;
}

}  // namespace bxmi
#include "count_bitmap.hpp"
#include "count_slices.hpp"
#include "find_exchange.hpp"
#include "count_dense.hpp"
namespace bxmi {

// ---- partitioned find: window + count per query in bucket order, offsets carried to bucket order ----
// For every query of [q_begin, q_end): hi = #{start < qe}, lo = #{prefix-max <= qs} and the number of hits in the
// window [lo, hi) of the tree-ordered arrays.  Ranks come from LDS trees one lane per query; the window is then
// scanned by 8 lanes per query with 16-byte loads (a per-lane serial scan would issue 8 scattered requests per query).
struct WindowSlices {
    int sLo, nS, kS, strideS;  // staged slice of the starts (tree order)
    int pLo, nP, kP, strideP;  // staged slice of the prefix-max array
    int qeLo, qeHi;            // rank_lt(starts, qe) may use the slice iff qeLo <= qe <= qeHi
};

template <int THREADS>
__device__ __forceinline__ void window_stage(const IndexDev &ix, const WindowSlices &w, int32_t *lds, int32_t *&treeP, int32_t *&treeS)
{
    treeP = lds, treeS = lds + (1 << w.kP);
    const int total = (1 << w.kP) + (1 << w.kS);
    for (int i = threadIdx.x; i < total; i += THREADS) lds[i] = INT_MAX;
    __syncthreads();
    part_stage_tree<THREADS>(treeP, w.kP, ix.pm + w.pLo, w.nP, w.strideP);
    part_stage_tree<THREADS>(treeS, w.kS, ix.s_ord + w.sLo, w.nS, w.strideS);
    __syncthreads();
}

template <int THREADS, bool PAIRS /* qs_arr is an array of (qs, qe) pairs, qe_arr unused */,
          bool PER_LANE /* neighbouring queries have neighbouring windows (sorted batch): one lane scans one window */>
__device__ __forceinline__ void window_queries(const IndexDev &ix, const WindowSlices &w, const int32_t *treeP, const int32_t *treeS,
                                               int64_t q_begin, int64_t q_end, const int32_t *__restrict__ qs_arr,
                                               const int32_t *__restrict__ qe_arr, int32_t *__restrict__ win_lo,
                                               int32_t *__restrict__ win_hi, int32_t *__restrict__ counts)
{
    for (int64_t i0 = q_begin + threadIdx.x; i0 - threadIdx.x < q_end; i0 += THREADS) {
        const bool live = i0 < q_end;
        int qs = 0, qe = 0;
        if (PAIRS) {
            const int2 v = live ? reinterpret_cast<const int2 *>(qs_arr)[i0] : make_int2(0, 0);
            qs = v.x, qe = v.y;
        } else if (live) {
            qs = qs_arr[i0], qe = qe_arr[i0];
        }
        int rS = 1, rP = 1;
        for (int it = 0; it < w.kS; it++) rS = 2 * rS + (treeS[rS] < qe);
        for (int it = 0; it < w.kP; it++) rP = 2 * rP + (treeP[rP] <= qs && qs != INT_MAX);
        rS = (rS - (1 << w.kS)) * w.strideS;
        rP = (rP - (1 << w.kP)) * w.strideP;
        if (w.strideS > 1) rS = group_rank_lt(ix.s_ord + w.sLo, rS, rS + w.strideS < w.nS ? rS + w.strideS : w.nS, qe);
        if (w.strideP > 1 && qs != INT_MAX)
            rP = group_rank_lt(ix.pm + w.pLo, rP, rP + w.strideP < w.nP ? rP + w.strideP : w.nP, qs + 1);
        const bool in_slice = qe >= w.qeLo && qe <= w.qeHi;
        int hi = in_slice ? w.sLo + rS : global_rank_lt(ix.s_ord, 0, ix.n, qe);
        int lo = qs == INT_MAX ? ix.n : w.pLo + rP;
        if (!live) lo = hi = 0;
        int mine = 0;
        if (PER_LANE) {
            // window scan, one lane per query (see part_fill_lane_kernel); a long window is counted by the whole wave
            const bool wide = hi - lo > LANE_WINDOW;
            if (!wide) {
                for (int k = lo; k < hi; k++) mine += ix.e_ord[k] > qs;
            }
            unsigned long long wm = __ballot(wide);
            while (wm) {
                const int src = __ffsll((long long)wm) - 1;
                wm &= wm - 1;
                const int L = __shfl(lo, src, 64), H = __shfl(hi, src, 64), S = __shfl(qs, src, 64);
                int c = 0;
                for (int k = L + lane_id(); k < H; k += 64) c += ix.e_ord[k] > S;
    #pragma unroll
                for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
                if (lane_id() == src) mine = c;
            }
        } else {
            const int sub = threadIdx.x & 7, gbase = lane_id() & ~7;
            // cooperative window scan: the 8 lanes of a group take their 8 queries one after the other; the first
            // 32-candidate step of all 8 windows is loaded up front (one dependent round trip instead of eight)
            int wl[8], wh[8], wk[8];
            int4 v[8];
    #pragma unroll
            for (int r = 0; r < 8; r++) {
                wl[r] = __shfl(lo, gbase + r, 64);
                wh[r] = __shfl(hi, gbase + r, 64);
                wk[r] = __shfl(qs, gbase + r, 64);
                v[r] = wl[r] < wh[r] ? *reinterpret_cast<const int4 *>(ix.e_ord + (wl[r] & ~(FAN - 1)) + sub * 4) : make_int4(0, 0, 0, 0);
            }
    #pragma unroll
            for (int r = 0; r < 8; r++) {
                int c = 0;
                if (wl[r] < wh[r]) {
                    const int k0 = wl[r] & ~(FAN - 1), kb = k0 + sub * 4;
                    c += (kb + 0 >= wl[r] && kb + 0 < wh[r] && v[r].x > wk[r]);
                    c += (kb + 1 >= wl[r] && kb + 1 < wh[r] && v[r].y > wk[r]);
                    c += (kb + 2 >= wl[r] && kb + 2 < wh[r] && v[r].z > wk[r]);
                    c += (kb + 3 >= wl[r] && kb + 3 < wh[r] && v[r].w > wk[r]);
                    c = group8_sum_dpp(c);
                    if (k0 + FAN < wh[r]) c += window_count<true>(ix.e_ord, k0 + FAN, wh[r], wk[r], sub);  // long window: the rest
                }
                if (sub == r) mine = c;
            }
        }
        if (live) {
            win_lo[i0] = lo;
            win_hi[i0] = hi;
            counts[i0] = mine;
        }
    }
}

__global__ __launch_bounds__(PT_THREADS) void part_window_kernel(IndexDev ix, const SliceBound *__restrict__ bounds,
                                                                 const int32_t *__restrict__ wg_first,
                                                                 const unsigned *__restrict__ table,
                                                                 const int2 *__restrict__ pairs /* (qs, qe), bucket order */, int64_t nq,
                                                                 int32_t *__restrict__ win_lo, int32_t *__restrict__ win_hi,
                                                                 int32_t *__restrict__ counts)
{
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    __shared__ int s_bucket;
    int b;
    int64_t q_begin, q_end;
    if (!part_chunk_of_block(wg_first, table, nq, &s_bucket, b, q_begin, q_end)) return;
    const SliceBound sb = bounds[b];
    const WindowSlices w = {sb.sLo, sb.sHi - sb.sLo, sb.kS, sb.strideS, sb.pLo, sb.pHi - sb.pLo, sb.kP, sb.strideP, sb.qeLo, sb.qeHi};
    int32_t *treeP, *treeS;
    window_stage<PT_THREADS>(ix, w, lds, treeP, treeS);
    window_queries<PT_THREADS, true, false>(ix, w, treeP, treeS, q_begin, q_end, reinterpret_cast<const int32_t *>(pairs), nullptr, win_lo, win_hi, counts);
}

// Are the starts non-decreasing?  (find path: decided on the host before anything else is launched)
__global__ void ivl_sorted_check_kernel(const int32_t *__restrict__ qs, int64_t nq, unsigned *__restrict__ unsorted)
{
    bool descent = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 1 < nq; i += (int64_t)gridDim.x * blockDim.x)
        descent |= qs[i] > qs[i + 1];
    if (__ballot(descent) && lane_id() == 0 && *unsorted == 0) *unsorted = 1;
}

// Values in query order -> bucket order (the inverse of part_gather_kernel): a workgroup drops its tile's
// values into LDS at the slots the scatter recorded, then streams the tile's runs out, one per bucket.
__global__ __launch_bounds__(PT_THREADS) void part_permute_i64_kernel(const long long *__restrict__ values,
                                                                      const unsigned short *__restrict__ lpos,
                                                                      const unsigned *__restrict__ tile_table, int64_t ntiles,
                                                                      int64_t nq, long long *__restrict__ bucketed)
{
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    long long *vals = reinterpret_cast<long long *>(dyn);                           // [PT_TILE]
    unsigned short *toff = reinterpret_cast<unsigned short *>(vals + PT_TILE);      // [PT_NB + 2]
    unsigned *gbase = reinterpret_cast<unsigned *>(toff + PT_NB + 2);               // [PT_NB]
    __shared__ unsigned scan_tmp[16];
    const int64_t tile = part_tile_of_block(ntiles);
    if (tile >= ntiles) return;
    const int64_t base = tile * PT_TILE;
    const int n = (int)(nq - base < PT_TILE ? nq - base : PT_TILE);
    {
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
    }
    for (int k = threadIdx.x; k < n; k += PT_THREADS) vals[lpos[base + k]] = values[base + k];
    __syncthreads();
    const int sub = threadIdx.x & 7;
    for (int b = threadIdx.x >> 3; b < PT_NB; b += PT_THREADS / 8) {
        unsigned o = toff[b], len = (b + 1 < PT_NB ? toff[b + 1] : (unsigned)n) - o, gb = gbase[b];
        for (unsigned r = sub; r < len; r += 8) bucketed[gb + r] = vals[o + r];
    }
}

// One 32-candidate step of a window: compact the hits of this step behind `base` (CSR order = tree order).
__device__ __forceinline__ int fill_step(int4 v, int4 id, int kb, int lo, int hi, int qs, int64_t base, int32_t *__restrict__ hits,
                                         int gshift, unsigned below)
{
    bool f0 = kb + 0 >= lo && kb + 0 < hi && v.x > qs;
    bool f1 = kb + 1 >= lo && kb + 1 < hi && v.y > qs;
    bool f2 = kb + 2 >= lo && kb + 2 < hi && v.z > qs;
    bool f3 = kb + 3 >= lo && kb + 3 < hi && v.w > qs;
    unsigned b0 = (unsigned)(__ballot(f0) >> gshift) & 0xffu;
    unsigned b1 = (unsigned)(__ballot(f1) >> gshift) & 0xffu;
    unsigned b2 = (unsigned)(__ballot(f2) >> gshift) & 0xffu;
    unsigned b3 = (unsigned)(__ballot(f3) >> gshift) & 0xffu;
    if (f0 | f1 | f2 | f3) {
        int64_t pos = base + __popc(b0 & below) + __popc(b1 & below) + __popc(b2 & below) + __popc(b3 & below);
        if (f0) hits[pos++] = id.x;
        if (f1) hits[pos++] = id.y;
        if (f2) hits[pos++] = id.z;
        if (f3) hits[pos++] = id.w;
    }
    return __popc(b0) + __popc(b1) + __popc(b2) + __popc(b3);
}

// Fill pass in bucket order: the window reads stay inside the bucket's lines (L2) instead of touching two random
// lines per query, and each 8-lane group keeps FILL_Q queries in flight (metadata and the first step of every
// window are loaded before any of them is compacted: the chain load-meta -> load-window -> store is latency bound).
constexpr int FILL_Q = 4;
__global__ __launch_bounds__(FIND_THREADS) void part_fill_kernel(IndexDev ix, const int32_t *__restrict__ qs_arr, int qs_stride /* 2: (qs, qe) pairs */,
                                                                int64_t nq,
                                                                const int32_t *__restrict__ win_lo,
                                                                const int32_t *__restrict__ win_hi,
                                                                const int32_t *__restrict__ cnt,
                                                                const long long *__restrict__ boffs,
                                                                int32_t *__restrict__ hits)
{
    const int lane = lane_id();
    const int sub = lane & 7, gshift = lane & ~7;
    const unsigned below = (1u << sub) - 1u;
    // contiguous block of queries per workgroup, XCD-aware: neighbours in bucket order share lines
    const int64_t per_xcd = ((int64_t)gridDim.x + 7) >> 3;
    const int64_t wg = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int64_t per_wg = (nq + gridDim.x - 1) / gridDim.x;
    const int64_t q0 = wg * per_wg, q1 = q0 + per_wg < nq ? q0 + per_wg : nq;
    for (int64_t qb = q0 + (int64_t)(threadIdx.x >> 3) * FILL_Q; qb < q1; qb += (FIND_THREADS / 8) * FILL_Q) {
        int lo[FILL_Q], hi[FILL_Q], qs[FILL_Q];
        int64_t base[FILL_Q];
#pragma unroll
        for (int j = 0; j < FILL_Q; j++) {
            const int64_t q = qb + j;
            const bool live = q < q1 && cnt[q] != 0;
            lo[j] = live ? win_lo[q] : 0;
            hi[j] = live ? win_hi[q] : 0;
            qs[j] = live ? qs_arr[q * qs_stride] : 0;
            base[j] = live ? boffs[q] : 0;
        }
        int4 ve[FILL_Q], vi[FILL_Q];
#pragma unroll
        for (int j = 0; j < FILL_Q; j++) {
            const int kb = (lo[j] & ~(FAN - 1)) + sub * 4;
            if (lo[j] < hi[j]) {
                ve[j] = *reinterpret_cast<const int4 *>(ix.e_ord + kb);
                vi[j] = *reinterpret_cast<const int4 *>(ix.idx + kb);
            }
        }
#pragma unroll
        for (int j = 0; j < FILL_Q; j++) {
            if (lo[j] >= hi[j]) continue;
            int k0 = lo[j] & ~(FAN - 1);
            base[j] += fill_step(ve[j], vi[j], k0 + sub * 4, lo[j], hi[j], qs[j], base[j], hits, gshift, below);
            for (k0 += FAN; k0 < hi[j]; k0 += FAN) {
                const int kb = k0 + sub * 4;
                int4 v = *reinterpret_cast<const int4 *>(ix.e_ord + kb);
                int4 id = *reinterpret_cast<const int4 *>(ix.idx + kb);
                base[j] += fill_step(v, id, kb, lo[j], hi[j], qs[j], base[j], hits, gshift, below);
            }
        }
    }
}

// The same walk with both of its memory sides made FLAT (round 4).  The kernel above reads the pairs and writes the hits one
// lane at a time: a wave's 64 queries own one contiguous stretch of the CSR list (~320 hits) and one contiguous window of the
// pairs (~100), but every store instruction scatters 64 4-byte pieces over the stretch's ten lines and every load is a lane's own
// dependent step.  Here a wave first copies the window [wbase, kmax) of the pairs into LDS (coalesced 512-byte loads), the
// lanes walk down inside LDS and drop their hits into an LDS image of the wave's stretch, and the stretch goes out as whole
// 256-byte stores, lane i taking positions i, i + 64, ...  A batch whose stretch is longer than FF_HITS (queries on a pile) or
// whose lanes leave the staged window keeps the direct loads / stores for those accesses: exact either way.
// (measured on configs[4] sorted by start, find() end to end: the kernel above 2.42 ms; FF_HITS / FF_PAIRS = 1024 / 256: 1.81 ms,
// 768 / 256: 1.71, 512 / 128: 1.64 -- less LDS per wave, more workgroups per CU)
constexpr int FF_HITS = 512;    // hits of a wave's 64 queries staged in LDS (mean 320 on configs[4])
constexpr int FF_PAIRS = 128;   // pairs below the wave's highest `hi` staged in LDS

// The pairs a wave's 64 queries will walk: [wbase, kmax) = the FF_PAIRS ranks below the highest `hi` of the wave, requested into
// registers ahead of time -- all at once (round 4 issued them one after the other behind a branch each: two dependent round trips
// to HBM per batch, after two more for the queries' numbers; the kernel's time was those four latencies), and by the callers one
// batch EARLY, while the batch before is being walked.
struct FfStage {
    int2 pv[FF_PAIRS / 64];
    int wbase, kmax;
};
__device__ __forceinline__ void ff_load(const int2 *__restrict__ eid /* at index 0 */, int c, int hi, FfStage &S)
{
    const int lane = lane_id();
    S.kmax = wave_max_i32(c ? hi : 0);
    S.wbase = S.kmax > FF_PAIRS ? S.kmax - FF_PAIRS : 0;
    const int last = S.kmax > 0 ? S.kmax - 1 : 0;
#pragma unroll
    for (int j = 0; j < FF_PAIRS / 64; j++) {
        const int kk = S.wbase + 64 * j + lane;
        S.pv[j] = eid[kk < S.kmax ? kk : last];  // (a valid address: no branch around the loads)
    }
}

// One wave, 64 consecutive queries (a lane each: `c` hits to find below rank `hi`, its CSR offset `off`): see the kernel below.
// wp / wh: the wave's LDS images of the pairs and of its stretch of the list; S: what ff_load brought for these queries.
__device__ __forceinline__ void ff_wave_fill(int2 *wp, int32_t *wh, const int2 *__restrict__ eid /* at index 0 */, const FfStage &S, int c, const int hi,
                                             const int qs, const long long off, int32_t *__restrict__ hits)
{
    const int lane = lane_id();
    int k = hi - 1;
    // the wave's stretch of the list: from its first query's offset, as long as the sum of its counts
    const long long base_off = __shfl(off, 0, 64);
    long long total64 = c;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) total64 += __shfl_xor(total64, d, 64);
    if (total64 == 0) return;  // (wave-uniform)
    const bool flat = total64 <= FF_HITS;
    const int total = flat ? (int)total64 : 0;
    const int rel = flat ? (int)(off - base_off) : 0;
    int32_t *__restrict__ dst = hits + off;
    // the window of the pairs: FF_PAIRS below the highest hi of the wave
    const int kmax = S.kmax, wbase = S.wbase;
#pragma unroll
    for (int j = 0; j < FF_PAIRS / 64; j++) {
        const int kk = wbase + 64 * j + lane;
        if (kk < kmax) wp[64 * j + lane] = S.pv[j];
    }
    // (lanes read what OTHER lanes staged: wave-level release + barrier, not just in-order DS issue)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    auto pair_at = [&](int kk) -> int2 { return kk >= wbase ? wp[kk - wbase] : eid[kk]; };
    auto take = [&](const int2 p) {
        if (p.x > qs) {
            --c;
            if (flat)
                wh[rel + c] = p.y;
            else
                dst[c] = p.y;
        }
    };
    for (int step = 0; step < LANE_WINDOW && c > 0 && k >= 0; step++, k--) {
        // (two self-contained arms: where an LDS read and an HBM load meet in one value the compiler waits for ALL outstanding
        // memory operations at every step -- the next batch's numbers and pairs included)
        if (__all(k >= wbase))
            take(wp[k - wbase]);
        else
            take(pair_at(k));
    }
    unsigned long long m = __ballot(c > 0);  // long walks (a few long targets far below hi): the wave takes them one by one
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        int C = __shfl(c, src, 64), K = __shfl(k, src, 64);
        const int S = __shfl(qs, src, 64);
        const int R = __shfl(rel, src, 64);
        int32_t *D = reinterpret_cast<int32_t *>(__shfl((long long)reinterpret_cast<uintptr_t>(dst), src, 64));
        while (C > 0 && K >= 0) {
            const int kk = K - lane;
            int2 p = make_int2(INT_MIN, 0);
            if (kk >= 0) p = pair_at(kk);
            const bool f = kk >= 0 && p.x > S;
            const unsigned long long fm = __ballot(f);
            // hits at higher indices come later in the list: lane 0 (the highest index of the step) takes the last free slot
            const int before = __popcll(fm & ((1ull << lane) - 1ull));
            if (f && before < C) {
                if (flat)
                    wh[R + C - 1 - before] = p.y;
                else
                    D[C - 1 - before] = p.y;
            }
            C -= __popcll(fm);
            K -= 64;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < total; i += 64) hits[base_off + i] = wh[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the next batch overwrites both images)
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(FIND_THREADS) void part_fill_flat_kernel(const int2 *__restrict__ eid /* at index 0 */, const int32_t *__restrict__ qs_arr,
                                                                     int64_t nq, const int32_t *__restrict__ his, const int32_t *__restrict__ cnt,
                                                                     const long long *__restrict__ offs, int32_t *__restrict__ hits)
{
    __shared__ int2 s_pairs[FIND_THREADS / 64][FF_PAIRS];
    __shared__ int32_t s_hits[FIND_THREADS / 64][FF_HITS];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int64_t per_xcd = ((int64_t)gridDim.x + 7) >> 3;
    const int64_t wg = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int64_t per_wg = (nq + gridDim.x - 1) / gridDim.x;
    const int64_t q0 = wg * per_wg, q1 = q0 + per_wg < nq ? q0 + per_wg : nq;
    struct Q {
        int c, hi, qs;
        long long off;
    };
    auto load_q = [&](int64_t qb, Q &x) {  // (independent loads: one round trip)
        const int64_t q = qb + lane;
        const bool live = q < q1;
        const int64_t qa = live ? q : q1 - 1;  // (a dead lane: valid addresses, no hits; its offset is the one behind the stretch's last query)
        x.c = cnt[qa], x.hi = his[qa], x.qs = qs_arr[qa], x.off = offs[qa];
        if (!live) x.c = 0;
    };
    if (q0 + 64 * wave >= q1) return;
    Q cur;
    load_q(q0 + 64 * wave, cur);
    for (int64_t qb = q0 + 64 * wave; qb < q1; qb += FIND_THREADS) {  // (waves are on their own: no workgroup barrier in here)
        FfStage S;
        ff_load(eid, cur.c, cur.hi, S);
        Q nxt = cur;
        if (qb + FIND_THREADS < q1) load_q(qb + FIND_THREADS, nxt);  // the next batch's numbers travel while this one is walked
        ff_wave_fill(s_pairs[wave], s_hits[wave], eid, S, cur.c, cur.hi, cur.qs, cur.off, hits);
        cur = nxt;
    }
}

// CSR offsets of a sorted find(): offsets[q] = chunk_base[chunk of q] + the exclusive prefix of the chunk's counts -- one read of
// the counts, one write of the offsets (the three-kernel scan read the counts twice and took 0.25 ms per 50 M).
// (Tried on top, round 5: the fill making the offsets itself -- a workgroup per run of chunks, a block scan per batch of 512 queries,
// no offsets kernel and no 8 bytes per query read back: 1.75 ms against 1.53 with this kernel + part_fill_flat_kernel, whose waves
// run free of barriers; forced to 8 waves per SIMD it spilled and took 1.86.  Not kept.)
__global__ __launch_bounds__(LC_THREADS) void lf_offsets_kernel(const int32_t *__restrict__ cnt, const long long *__restrict__ chunk_base, int64_t nq,
                                                                long long *__restrict__ offsets)
{
    __shared__ long long lds[16];
    const int64_t base = (int64_t)blockIdx.x * LC_CHUNK + (int64_t)threadIdx.x * LC_ITEMS;
    int c[LC_ITEMS];
    if (base + LC_ITEMS <= nq) {
        const int4 a = *reinterpret_cast<const int4 *>(cnt + base), b = *reinterpret_cast<const int4 *>(cnt + base + 4);
        c[0] = a.x, c[1] = a.y, c[2] = a.z, c[3] = a.w, c[4] = b.x, c[5] = b.y, c[6] = b.z, c[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < LC_ITEMS; j++) c[j] = base + j < nq ? cnt[base + j] : 0;
    }
    long long run = 0;
#pragma unroll
    for (int j = 0; j < LC_ITEMS; j++) run += c[j];
    long long total;
    long long off = chunk_base[blockIdx.x] + block_exclusive_scan(run, OpSum(), 0ll, lds, &total);
    static_assert(LC_ITEMS == 8, "eight consecutive counts per thread");
    if (base + LC_ITEMS <= nq) {
        long long o[LC_ITEMS];
#pragma unroll
        for (int j = 0; j < LC_ITEMS; j++) {
            o[j] = off;
            off += c[j];
        }
#pragma unroll
        for (int j = 0; j < LC_ITEMS; j += 2)
            *reinterpret_cast<longlong2 *>(offsets + base + j) = make_longlong2(o[j], o[j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < LC_ITEMS; j++) {
            if (base + j < nq) offsets[base + j] = off;
            off += c[j];
        }
    }
}

// (Round 5's fused variant -- count, CSR offsets by decoupled look-back and fill in ONE kernel -- measured 2.65 ms against 1.45 for the
// stages: the count half is a chain of dependent loads that lives on four workgroups per CU, the fused kernel's registers left two.
// Removed in round 6; HISTORY.md has its design.)
__global__ void part_fold_total_kernel(unsigned long long *__restrict__ slots, unsigned long long *__restrict__ total)
{
    unsigned long long v = threadIdx.x < PT_SLOTS ? slots[threadIdx.x] : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (threadIdx.x == 0 && v) atomicAdd(total, v);
}

// Slice bounds of every bucket; depends only on the sealed index, so it is built once at seal().
__global__ void part_bounds_kernel(const int32_t *__restrict__ s_ord, const int32_t *__restrict__ e_sorted,
                                   const int32_t *__restrict__ pm, int n, PartGeom g, SliceBound *__restrict__ out)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= PT_NB) return;
    const long long W = 1ll << g.shift;
    const long long lo = b == 0 ? (long long)INT_MIN - 1 : (long long)g.cmin + (long long)b * W;          // qs >= lo
    const long long hi = b == PT_NB - 1 ? (long long)INT_MAX + 1 : (long long)g.cmin + (long long)(b + 1) * W;  // qs < hi
    auto rank_lt64 = [n](const int32_t *a, long long x) {
        int l = 0, h = n;
        while (l < h) {
            int mid = (int)(((unsigned)l + (unsigned)h) >> 1);
            if ((long long)a[mid] < x)
                l = mid + 1;
            else
                h = mid;
        }
        return l;
    };
    SliceBound sb;
    // ends: keys qs+1 lie in [lo+1, hi]
    sb.eLo = rank_lt64(e_sorted, lo + 1);
    sb.eHi = rank_lt64(e_sorted, hi + 1);
    // Each slice becomes a perfect tree of at most 2^13 - 1 keys (two trees = 64 KiB of LDS, two workgroups per
    // CU); a longer slice is sampled with the smallest stride that fits.
    const int nE = sb.eHi - sb.eLo;
    sb.sLo = rank_lt64(s_ord, lo);
    long long x = hi + (W >> 3) + 1;  // starts: keys qe of ordinary queries lie in [lo, hi + W/8]
    sb.sHi = rank_lt64(s_ord, x);
    const int nS = sb.sHi - sb.sLo;
    constexpr int TREE_KEYS = (1 << 13) - 1;
    sb.strideE = nE / TREE_KEYS + 1;
    sb.strideS = nS / TREE_KEYS + 1;
    int kE = 0, kS = 0;
    while ((1 << kE) - 1 < nE / sb.strideE) kE++;
    while ((1 << kS) - 1 < nS / sb.strideS) kS++;
    sb.kE = kE;
    sb.kS = kS;
    // prefix max of ends in tree order (monotone): #{pm <= qs} for qs in [lo, hi) lies in [#{pm < lo}, #{pm < hi}]
    sb.pLo = rank_lt64(pm, lo);
    sb.pHi = rank_lt64(pm, hi);
    const int nP = sb.pHi - sb.pLo;
    sb.strideP = nP / TREE_KEYS + 1;
    int kP = 0;
    while ((1 << kP) - 1 < nP / sb.strideP) kP++;
    sb.kP = kP;
    sb.qeLo = lo < INT_MIN ? INT_MIN : (int32_t)lo;
    sb.qeHi = x > INT_MAX ? INT_MAX : (int32_t)x;
    out[b] = sb;
}

// ---------------------------------------------------------------------------
// find kernels: window + count, then ballot-compacted fill
// ---------------------------------------------------------------------------

template <bool DPP>
__global__ __launch_bounds__(FIND_THREADS) void ivl_find_count_kernel(TreeDev S, TreeDev P, IndexDev ix,
                                                                     const int32_t *__restrict__ qs_arr,
                                                                     const int32_t *__restrict__ qe_arr, int64_t nq,
                                                                     int32_t *__restrict__ win_lo,
                                                                     int32_t *__restrict__ win_hi,
                                                                     int32_t *__restrict__ counts)
{
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    int32_t *ldsS = lds, *ldsP = lds + S.lds_ints;
    stage_tree(S, ldsS);
    stage_tree(P, ldsP);
    __syncthreads();
    const int sub = threadIdx.x & 7;
    const int64_t group = (int64_t)blockIdx.x * (FIND_THREADS / 8) + (threadIdx.x >> 3);
    const int64_t ngroups = (int64_t)gridDim.x * (FIND_THREADS / 8);
    for (int64_t q0 = group * FIND_Q; q0 < nq; q0 += ngroups * FIND_Q) {
        int qs[FIND_Q], qe[FIND_Q], kP[FIND_Q], hi[FIND_Q], lo[FIND_Q];
#pragma unroll
        for (int j = 0; j < FIND_Q; j++) {
            bool ok = q0 + j < nq;
            qs[j] = ok ? qs_arr[q0 + j] : 0;
            qe[j] = ok ? qe_arr[q0 + j] : 0;
            kP[j] = qs[j] == INT_MAX ? INT_MAX : qs[j] + 1;
        }
        tree_rank_lt<DPP, FIND_Q>(S, ldsS, qe, hi, sub);  // #{start < qe}
        tree_rank_lt<DPP, FIND_Q>(P, ldsP, kP, lo, sub);  // #{pm <= qs} = first k with pm[k] > qs
#pragma unroll
        for (int j = 0; j < FIND_Q; j++) {
            if (qs[j] == INT_MAX) lo[j] = ix.n;
            int c = lo[j] < hi[j] ? window_count<DPP>(ix.e_ord, lo[j], hi[j], qs[j], sub) : 0;
            if (sub == 0 && q0 + j < nq) {
                win_lo[q0 + j] = lo[j];
                win_hi[q0 + j] = hi[j];
                counts[q0 + j] = c;
            }
        }
    }
}

// One 8-lane group per query; every 32-element step is compacted with four
// wave ballots: the byte of this group in ballot j tells which of its lanes
// hit in slot j, so a lane's output position is a handful of popcounts.
__global__ __launch_bounds__(FIND_THREADS) void ivl_find_fill_kernel(IndexDev ix, const int32_t *__restrict__ qs_arr,
                                                                    int64_t nq, const int32_t *__restrict__ win_lo,
                                                                    const int32_t *__restrict__ win_hi,
                                                                    const int64_t *__restrict__ offsets,
                                                                    int32_t *__restrict__ hits)
{
    const int lane = lane_id();
    const int sub = lane & 7, gshift = lane & ~7;
    const unsigned below = (1u << sub) - 1u;
    const int64_t group = (int64_t)blockIdx.x * (FIND_THREADS / 8) + (threadIdx.x >> 3);
    const int64_t ngroups = (int64_t)gridDim.x * (FIND_THREADS / 8);
    for (int64_t q = group; q < nq; q += ngroups) {
        int lo = win_lo[q], hi = win_hi[q], qs = qs_arr[q];
        int64_t base = offsets[q];
        if (offsets[q + 1] == base) continue;
        for (int k0 = lo & ~(FAN - 1); k0 < hi; k0 += FAN) {
            int kb = k0 + sub * 4;
            int4 v = *reinterpret_cast<const int4 *>(ix.e_ord + kb);
            bool f0 = kb + 0 >= lo && kb + 0 < hi && v.x > qs;
            bool f1 = kb + 1 >= lo && kb + 1 < hi && v.y > qs;
            bool f2 = kb + 2 >= lo && kb + 2 < hi && v.z > qs;
            bool f3 = kb + 3 >= lo && kb + 3 < hi && v.w > qs;
            unsigned b0 = (unsigned)(__ballot(f0) >> gshift) & 0xffu;
            unsigned b1 = (unsigned)(__ballot(f1) >> gshift) & 0xffu;
            unsigned b2 = (unsigned)(__ballot(f2) >> gshift) & 0xffu;
            unsigned b3 = (unsigned)(__ballot(f3) >> gshift) & 0xffu;
            int step = __popc(b0) + __popc(b1) + __popc(b2) + __popc(b3);
            if (f0 | f1 | f2 | f3) {
                int64_t pos = base + __popc(b0 & below) + __popc(b1 & below) + __popc(b2 & below) + __popc(b3 & below);
                int4 id = *reinterpret_cast<const int4 *>(ix.idx + kb);
                if (f0) hits[pos++] = id.x;
                if (f1) hits[pos++] = id.y;
                if (f2) hits[pos++] = id.z;
                if (f3) hits[pos++] = id.w;
            }
            base += step;
        }
    }
}


// ---- one query, one launch: the latency path behind the per-call find() of the drop-in classes ----
// A single workgroup: 8 lanes walk the two search trees (all levels from L2), then the whole workgroup scans the
// window and compacts the hits with wave ballots straight into host-visible memory: launch + one stream sync.
constexpr int ONE_THREADS = 256;
__global__ __launch_bounds__(ONE_THREADS) void ivl_find_one_kernel(TreeDev S, TreeDev P, IndexDev ix, int qs, int qe,
                                                                  int32_t *__restrict__ out /* [0] = n (64-bit), hits from [2] */,
                                                                  int cap, unsigned long long seq)
{
    __shared__ int s_lo, s_hi;
    __shared__ int wave_tot[ONE_THREADS / 64];
    if (threadIdx.x < 8) {
        int key_s[1] = {qe}, key_p[1] = {qs == INT_MAX ? INT_MAX : qs + 1}, r_s[1], r_p[1];
        tree_rank_lt<true, 1>(S, nullptr, key_s, r_s, (int)threadIdx.x);
        tree_rank_lt<true, 1>(P, nullptr, key_p, r_p, (int)threadIdx.x);
        if (threadIdx.x == 0) {
            s_hi = r_s[0];
            s_lo = qs == INT_MAX ? ix.n : r_p[0];
        }
    }
    __syncthreads();
    const int lo = s_lo, hi = s_hi;
    long long run = 0;
    for (int b = lo; b < hi; b += ONE_THREADS) {
        const int k = b + (int)threadIdx.x;
        const bool f = k < hi && ix.e_ord[k] > qs;
        const unsigned long long m = __ballot(f);
        const int w = threadIdx.x >> 6;
        if (lane_id() == 0) wave_tot[w] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int i = 0; i < ONE_THREADS / 64; i++) {
            if (i < w) woff += wave_tot[i];
            tot += wave_tot[i];
        }
        const long long pos = run + woff + __popcll(m & lanemask_lt());
        if (f && pos < cap) out[2 + pos] = ix.idx[k];
        run += tot;
        __syncthreads();
    }
    // Every wave's hits must have LEFT the GPU before the completion word goes out: the barrier orders the waves, but a
    // workgroup-scope barrier does not wait for the other waves' stores to host memory, and thread 0's release only
    // covers its own wave's (seen as a rare wrong hit list in a per-line script).  So each wave drains its stores first.
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        *reinterpret_cast<long long *>(out) = run;
        publish_to_host(reinterpret_cast<unsigned long long *>(out + 2 + cap), seq);
    }
}

// before()/after() candidate filter over a window of the in-order arrays
// (single query, one workgroup): keeps k in [lo,hi) with vlo <= val[k] < vhi.
__global__ __launch_bounds__(256) void ivl_filter_window_kernel(const int32_t *__restrict__ val,
                                                               const int32_t *__restrict__ idx, int lo, int hi,
                                                               long long vlo, long long vhi, int reverse,
                                                               int32_t *__restrict__ out, int64_t cap,
                                                               unsigned long long *__restrict__ n_out)
{
    __shared__ int wave_tot[4];
    __shared__ long long run;
    if (threadIdx.x == 0) run = 0;
    __syncthreads();
    int span = hi - lo;
    for (int b = 0; b < span; b += 256) {
        int t = b + threadIdx.x;
        int k = reverse ? hi - 1 - t : lo + t;
        bool ok = t < span;
        bool f = false;
        if (ok) {
            long long v = val[k];
            f = v >= vlo && v < vhi;
        }
        unsigned long long m = __ballot(f);
        int w = threadIdx.x >> 6;
        if (lane_id() == 0) wave_tot[w] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int i = 0; i < 4; i++) {
            if (i < w) woff += wave_tot[i];
            tot += wave_tot[i];
        }
        long long pos = run + woff + __popcll(m & lanemask_lt());
        if (f && pos < cap) out[pos] = idx[k];
        __syncthreads();
        if (threadIdx.x == 0) run += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = (unsigned long long)run;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct Tree {
    DevBuf upper;  // levels 1.. (level 0 is the caller's padded sorted array)
    TreeDev dev{};
    int build(const int32_t *leaves, int64_t n, hipStream_t st)
    {
        dev = TreeDev{};
        int64_t nodes[MAXLEV];
        nodes[0] = n > 0 ? div_up(n, FAN) : 1;
        int nlev = 1;
        while (nodes[nlev - 1] > 1) {
            nodes[nlev] = div_up(nodes[nlev - 1], FAN);
            nlev++;
        }
        int64_t upper_ints = 0;
        for (int l = 1; l < nlev; l++) upper_ints += nodes[l] * FAN;
        BXMI_TRY(upper.reserve((size_t)(upper_ints + 4) * sizeof(int32_t)));
        int32_t *p = upper.as<int32_t>();
        dev.lev[0] = leaves;
        dev.lev_ints[0] = (int32_t)(nodes[0] * FAN);
        for (int l = 1; l < nlev; l++) {
            int64_t ints = nodes[l] * FAN;
            hipLaunchKernelGGL(ivl_tree_level_kernel, dim3(stream_grid(ints, 256)), dim3(256), 0, st, dev.lev[l - 1], nodes[l - 1],
                               p, ints);
            dev.lev[l] = p;
            dev.lev_ints[l] = (int32_t)ints;
            p += ints;
        }
        BXMI_LAUNCH_CHECK();
        dev.nlev = nlev;
        set_lds_budget(LDS_TREE_INTS);
        return BXMI_OK;
    }
    // Stage as many top levels as fit in `budget` ints.
    void set_lds_budget(int64_t budget)
    {
        int64_t used = 0;
        int from = dev.nlev;
        for (int l = dev.nlev - 1; l >= 0; --l) {
            if (used + dev.lev_ints[l] > budget) break;
            dev.lds_off[l] = (int32_t)used;
            used += dev.lev_ints[l];
            from = l;
        }
        dev.lds_from = from;
        dev.lds_ints = (int32_t)used;
    }
};

// ---------------------------------------------------------------------------
// distance clustering (ClusterTree)
// ---------------------------------------------------------------------------
// The reference keeps a treap of clusters and merges on insert (src/cluster.c:226-260, fix-ups :112-147); for
// max_dist >= 0 the outcome does not depend on the insertion order: walking the intervals by start, a new cluster
// begins exactly where  start - max_dist > (largest end so far)  -- verified against the reference's extension on
// 20 000 random trees.  The sealed index already holds the starts in order and the prefix maximum of the ends, so a
// cluster boundary is one comparison per interval.
// (*empty = 1 when some interval has end <= start: what decides whether max_dist = -1 has an answer, see bxmi_ivl_clusters)
__global__ void cluster_flag_kernel(const int32_t *__restrict__ s_ord, const int32_t *__restrict__ e_ord, const int32_t *__restrict__ pm, int n,
                                    int max_dist, int32_t *__restrict__ flag, int32_t *__restrict__ empty)
{
    bool mine = false;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        flag[i] = i == 0 || (long long)s_ord[i] - (long long)max_dist > (long long)pm[i - 1];
        mine |= e_ord[i] <= s_ord[i];
    }
    if (__any(mine) && lane_id() == 0) *empty = 1;  // (ordinary stores of one value: visible at the kernel's end)
}

// cluster id of every interval (inclusive scan of the flags, minus one) -> sort key (cluster, id), and the first position
// and start coordinate of each cluster
__global__ void cluster_keys_kernel(const int32_t *__restrict__ cid_incl, const int32_t *__restrict__ flag,
                                    const int32_t *__restrict__ s_ord, const int32_t *__restrict__ idx,
                                    const int32_t *__restrict__ ids /* per insertion index, may be NULL */, int n,
                                    unsigned long long *__restrict__ keys, int32_t *__restrict__ c_start,
                                    long long *__restrict__ c_off)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c = cid_incl[i] - 1;
        const int id = ids ? ids[idx[i]] : idx[i];
        keys[i] = ((unsigned long long)(unsigned)c << 32) | (unsigned long long)((uint32_t)id ^ 0x80000000u);
        if (flag[i]) {
            c_start[c] = s_ord[i];
            c_off[c] = i;
        }
    }
}

__global__ void cluster_finish_kernel(const unsigned long long *__restrict__ keys_sorted, const int32_t *__restrict__ pm,
                                      long long *__restrict__ c_off, int nclusters, int n, int32_t *__restrict__ c_end,
                                      int32_t *__restrict__ members)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        members[i] = (int32_t)((uint32_t)keys_sorted[i] ^ 0x80000000u);
        if (i < nclusters) {
            const long long last = (i + 1 < nclusters ? c_off[i + 1] : (long long)n) - 1;
            c_end[i] = pm[last];  // max_dist >= 0: every earlier cluster ends before this one starts
        }
        if (i == 0) c_off[nclusters] = n;
    }
}

static int64_t g_opt_partition = -1;  // -1 = auto (large batches), 0 = never, 1 = always
static int64_t g_opt_partition_min = 4 << 20;  // auto: partition batches of at least this many queries
constexpr int PT_MAX_SUB = 1;  // scratch regions are addressed per sub-batch; one region since sub-batch pipelining was dropped
constexpr int PT_SLOT_STRIDE = PT_SLOTS + 8;  // per sub-batch: the partial totals, then the "unsorted" flag
static int64_t g_opt_sorted_path = 1;  // 1 = batches whose starts are already sorted skip the bucketing (detected on the device)
static int64_t g_opt_bitmap = -1;      // second-generation count pass (count_bitmap.hpp): -1 = when the index qualifies, 0 = never, 1 = same as -1
static int64_t g_opt_bitmap_min = 2 << 20;  // auto: batches of at least this many queries take it (when the index qualifies)
static int64_t g_opt_bm_variant = -1;  // tile kernel shape: -1 = by batch size, 0 = 512 threads x 32 queries, 1 = 1024 x 16, 2 = 1024 x 32 (32768-query tiles)
static int64_t g_opt_fx_direct = -1;   // find() through the exchange: 1 = the fill writes straight into the CSR list (query-order prefixes from the un-permute
                                       // kernel, no copy), 0 = scratch + copy, -1 = by the size of the list the handle expects (see ivl_find_fx)
static int64_t g_opt_find_sliced = 1;  // large unsorted find() batches through the exchange (count_slices.hpp) where the slice stage fits; 0 = the bucketed find
static int64_t g_opt_slice = -1;       // search stage on staged key slices (count_slices.hpp): -1 = where the images do not pay or fit, 0 = never, 1 = wherever it fits
static int64_t g_opt_sl_f = -1;        // buckets per slice unit = 2^f: -1 = by run length and LDS, else forced (tests)
static int64_t g_opt_bm_chunk = 0;     // queries per search work item (0 = BM_CHUNK, twice that for bucket pairs)
static int64_t g_opt_sl_run_cap = 160;  // a slice unit grows only while its expected (tile, unit) run stays within this many records
static int64_t g_opt_sl_flat = 1;      // 1 = count-only passes on key slices take the flat 16-byte walk of count_dense.hpp (16-bit counts, unit run table), 0 = the 16 / 64 lanes-per-run kernels of count_slices.hpp
static int64_t g_opt_sl_rbits = 20;    // a slice unit's offsets take at most this many bits of the 32-bit record (the rest holds the length)
static int64_t g_opt_sl_lanes = 0;     // lanes per (tile, unit) run: 0 = by expected run length, 16 or 64, -1 (set as 1) = the flat walk for long runs
static int64_t g_opt_bm_hard_ppm = 2000;  // an index qualifies while its hard cells stay below this many per million cells
static int64_t g_opt_flat = -1;       // the flat 16-byte walk on cell images of 2^18-coordinate units (count_dense.hpp, bp_*): -1 = dense indexes that qualify, 0 = never, 1 = every index that qualifies
static int64_t g_opt_sparse = -1;     // offset-cell images for sparse indexes (offset_cells.hpp; the persistent walk): -1 = sparse indexes that qualify, batches that bring enough queries per unit; 0 = never; 1 = whatever the batch size
static int64_t g_opt_bo_cell_log2 = 0;  // their cell width: 0 = from the index's density, 6..8 = forced
static int64_t g_opt_bo_min_per_unit = 4096;  // queries per unit image a batch must bring (an image is 72 KB to load whatever the batch)
static int64_t g_opt_sorted_cells = 1;  // sorted batches on indexes with cell images: 1 = answered from the images stretch by stretch (bs_*), 0 = the first-generation kernel for sorted batches
static int64_t g_opt_dense = -1;      // search stage on dense unit images (count_dense.hpp): -1 = dense indexes that qualify, 0 = never, 1 = every index that qualifies
static int64_t g_opt_bd_chunk = 0;    // queries per search work item of the dense stage (0 = 256 Ki: one item per unit on a uniform 100 M batch)
static int64_t g_opt_bd_table_from = 0;  // dense images: overflow entries from which a cell gets a rank table (read when an index is prepared); 0 = 2 where the overflow area has the room, else 6
static int64_t g_opt_bd_blocks = 0;   // 1 = dense images with block-relative ranks even where unit-relative ones fit (tests)
static int64_t g_opt_bd_w8 = -1;      // 8-bit counts out of place: -1 = by index and feedback, 0 = never, 1 = whenever the layout allows
static int64_t g_opt_order_skip = -1;  // -1 = stop launching the order check after two batches in a row were not sorted (a probe of 8192 starts rides on the parameter kernel then), 0 = always check
static int64_t g_opt_host_chunk = 8 << 20;  // queries per chunk of the host-pointer count (upload of chunk k+1 / pass on k / download of k-1 at once); 0 = one piece
static int64_t g_opt_host_touchers = 2;  // host threads that touch the output array's pages ahead of the downloads (0 = the download faults them in)
static int64_t g_opt_bd_unit_log2 = 0;   // coordinates per unit of the dense images (read when an index is prepared): 0 = 19 if the duplicated coordinates fit its 12 KiB of overflow, else 18 (64 KiB: rank tables of clumped cells); 12 .. 19 = forced

// The option table: every knob of the interval path, its variable and how a value is normalised.  bxmi_set_option writes through
// it, bxmi_get_option / bxmi_option_at read it back -- the tests take their "defaults" from the library at import instead of
// keeping a copy.  Results never depend on an option.
struct IvlOpt {
    const char *key;
    int64_t *var;
    int64_t (*norm)(int64_t);
};
static const IvlOpt IVL_OPTS[] = {
    {"ivl.partition", &g_opt_partition, nullptr},
    {"ivl.sorted_path", &g_opt_sorted_path, [](int64_t value) -> int64_t { return value != 0; }},
    {"ivl.partition_min", &g_opt_partition_min, nullptr},
    {"ivl.bitmap_min", &g_opt_bitmap_min, nullptr},
    {"ivl.bitmap", &g_opt_bitmap, nullptr},
    {"ivl.bm_variant", &g_opt_bm_variant, [](int64_t value) -> int64_t { return value < 0 || value > 2 ? -1 : value; }},
    {"ivl.find_sliced", &g_opt_find_sliced, [](int64_t value) -> int64_t { return value != 0; }},
    {"ivl.fx_direct", &g_opt_fx_direct, [](int64_t value) -> int64_t { return value < 0 ? -1 : value != 0; }},
    {"ivl.slice", &g_opt_slice, nullptr},
    {"ivl.sl_f", &g_opt_sl_f, [](int64_t value) -> int64_t { return value > SL_MAX_F ? SL_MAX_F : value; }},
    {"ivl.bm_chunk", &g_opt_bm_chunk, [](int64_t value) -> int64_t { return value < 0 ? 0 : value; }},
    {"ivl.sl_run_cap", &g_opt_sl_run_cap, [](int64_t value) -> int64_t { return value < 8 ? 8 : value; }},
    {"ivl.sl_flat", &g_opt_sl_flat, [](int64_t value) -> int64_t { return value != 0; }},
    {"ivl.sl_rbits", &g_opt_sl_rbits, [](int64_t value) -> int64_t { return value < 17 ? 17 : (value > 24 ? 24 : value); }},
    {"ivl.sl_lanes", &g_opt_sl_lanes, [](int64_t value) -> int64_t { return value == 16 || value == 64 ? value : (value == 1 ? -1 : 0); /* 1 = the flat walk */ }},
    {"ivl.bm_hard_ppm", &g_opt_bm_hard_ppm, nullptr},
    {"ivl.flat", &g_opt_flat, nullptr},
    {"ivl.dense", &g_opt_dense, nullptr},
    {"ivl.sparse", &g_opt_sparse, nullptr},
    {"ivl.sorted_cells", &g_opt_sorted_cells, [](int64_t value) -> int64_t { return value != 0; }},
    {"ivl.bo_cell_log2", &g_opt_bo_cell_log2, [](int64_t value) -> int64_t { return value < BO_MIN_K || value > BO_MAX_K ? 0 : value; }},
    {"ivl.bo_min_per_unit", &g_opt_bo_min_per_unit, [](int64_t value) -> int64_t { return value < 0 ? 0 : value; }},
    {"ivl.bd_chunk", &g_opt_bd_chunk, [](int64_t value) -> int64_t { return value < 0 ? 0 : value; }},
    {"ivl.bd_blocks", &g_opt_bd_blocks, [](int64_t value) -> int64_t { return value != 0; }},
    {"ivl.bd_table_from", &g_opt_bd_table_from, [](int64_t value) -> int64_t { return value < 1 || value > 64 ? 0 : value; }},
    {"ivl.bd_w8", &g_opt_bd_w8, nullptr},
    {"ivl.order_skip", &g_opt_order_skip, nullptr},
    {"ivl.host_chunk", &g_opt_host_chunk, [](int64_t value) -> int64_t { return value <= 0 ? 0 : ((value + 4095) & ~(int64_t)4095); }},
    {"ivl.host_touchers", &g_opt_host_touchers, [](int64_t value) -> int64_t { return value < 0 ? 0 : (value > 16 ? 16 : value); }},
    {"ivl.bd_unit_log2", &g_opt_bd_unit_log2, [](int64_t value) -> int64_t { return value < 12 || value > BD_UNIT_LOG2 ? 0 : value; }},
};
constexpr int IVL_NOPTS = (int)(sizeof(IVL_OPTS) / sizeof(IVL_OPTS[0]));

int ivl_set_option(const char *key, int64_t value)
{
    for (int i = 0; i < IVL_NOPTS; i++)
        if (!strcmp(key, IVL_OPTS[i].key)) {
            *IVL_OPTS[i].var = IVL_OPTS[i].norm ? IVL_OPTS[i].norm(value) : value;
            return 1;
        }
    return 0;
}

int ivl_option_count() { return IVL_NOPTS; }

int ivl_option_at(int i, const char **key, int64_t *value)
{
    if (i < 0 || i >= IVL_NOPTS) return 0;
    *key = IVL_OPTS[i].key;
    *value = *IVL_OPTS[i].var;
    return 1;
}

}  // namespace bxmi

using namespace bxmi;

struct bxmi_ivl {
    // host staging of appended intervals (insertion order)
    std::vector<int32_t> h_start, h_end;
    // device copies in insertion order
    DevBuf d_start, d_end;
    int64_t n_dev = 0;   // intervals resident on device (insertion order)
    int64_t n = 0;       // intervals in the sealed index
    bool sealed = false;
    int has_reversed = 0;
    // sealed index
    DevBuf keys_a, keys_b, ekeys_a, ekeys_b;
    DevBuf s_ord, e_ord, idx, pm, e_sorted, flag;
    Tree treeS, treeE, treeP;
    SortScratch sort_scratch;
    DevBuf scan_scratch;
    // query scratch
    DevBuf q_s, q_e, q_cnt, q_lo, q_hi, q_off, q_hits, q_total;
    // partitioned count path
    PartGeom geom{0, 0};
    bool images_ready = false;
    DevBuf slice_bounds, cell_images, cell_meta, p_hist, p_table, p_pairs, p_dest, p_cnt, p_plan, p_slots, p_lo, p_hi, p_boffs;
    // second-generation count pass (count_bitmap.hpp)
    int32_t cmax = 0;            // largest end of the sealed index
    DevBuf bm_recs, bm_slots, bm_tbl, bm_runT, bm_grpcnt, bm_items, bm_params;
    // slice search (count_slices.hpp)
    int sl_state = 0;            // 0 = not decided yet, 1 = boundary table built and a single bucket's keys fit the LDS, -1 = they do not
    unsigned sl_need[SL_MAX_F + 1] = {0, 0, 0, 0, 0, 0, 0};  // most keys a unit of 2^f buckets stages
    DevBuf sl_meta, sl_stats, sl_unitcnt, sl_cnt, sl_loff, sl_hits, sl_eid;
    // find() through the exchange, second generation (find_exchange.hpp)
    int fx_state = 0;            // 0 = not decided yet, 1 = the half-bucket ranks are built, -1 = the grid has no half buckets (shift 0)
    std::vector<int2> fx_meta2_host;   // the ranks at the half-bucket boundaries (read back once per sealed index)
    std::vector<FxPiece> fx_pieces_host;
    int fx_pieces_f = -1;        // the unit size (2^f buckets) the piece list was cut for
    double fx_hits_per_q = -1.0; // hits per query of the handle's latest find() through the exchange (predicts the next list's size), < 0 = none yet
    DevBuf lf_state;             // sorted find() in one kernel: the chunks' look-back words and the ticket
    DevBuf fx_meta2, fx_pieces, fx_tbl2, fx_runT2, fx_hc, fx_svq, fx_parts, fx_tile_tot, fx_tile_base, fx_work;
    // dense unit images (count_dense.hpp)
    int bd_state = 0;            // 0 = not decided yet, 1 = images built and the index qualifies, -1 = it does not
    unsigned bd_worst[2] = {0, 0};  // what bd_image_kernel reported: most keys of one block, most overflow entries of one unit
    BmGeom bd_geom{0, 0, 0, 0, 0, 0, 0, BD_RSHIFT, 0};
    int bp_state = 0;            // cell images of units for the flat walk: 0 = not decided yet, 1 = built and the index qualifies, -1 = it does not
    int64_t bp_hard_cells = 0;
    BmGeom bp_geom{0, 0, 0, 0, 0, 0, 0, BP_RSHIFT, 0};
    DevBuf bp_images, bp_stats;
    int bo_state = 0;            // offset-cell images of units (sparse indexes): 0 = not decided yet, 1 = built and the index qualifies, -1 = it does not
    int64_t bo_hard_cells = 0;
    BmGeom bo_geom{0, 0, 0, 0, 0, 0, 0, BP_RSHIFT, 0};  // dshift = cell width - 5
    DevBuf bo_images;
    DevBuf bs_plan;              // sorted batches on cell images: [unit bounds][item count][items]
    bool bd_blocks = false;      // the images' ranks are relative to blocks of 1024 cells (more than 32767 keys in some unit's slice)
    DevBuf bd_images, bd_stats, bd_cnt16, bd_unitT, bd_tend;
    // 8-bit counts between the search and the un-permute kernel (bm_count_segments): the un-permute kernel keeps a running
    // total of the counts that did not fit and mirrors it into host memory
    DevBuf bd_fb;                              // the running total (device)
    unsigned long long *bd_fb_host = nullptr;  // its mirror (host memory the device can write)
    int64_t w8_queries = 0;                    // queries of the passes launched with 8-bit counts
    bool w8_off = false;                       // too many of them did not fit: this index keeps 16-bit counts
    // [1] of the same host words: what the order check of an earlier pass found (ivl_local_count_kernel writes it)
    unsigned long long order_seq = 0, order_seen = 0;  // passes launched with an order check / the last one the host has seen the answer of
    int unsorted_streak = 0;                           // consecutive answers "not sorted"
    bool order_skip = false;                           // the order check is not launched at present (the tile sort reports the order)
    bool sl_eid_ready = false;
    int32_t *one_buf = nullptr;  // host-visible result of bxmi_ivl_find_one: [n:int64][ONE_CAP hits][completion word:int64]
    unsigned long long one_seq = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream_up = nullptr, stream_down = nullptr;  // the host-pointer entry points' copies either side of the pass (ivl_count_host_chunks)
    int device = 0;
};


static IndexDev index_dev(const bxmi_ivl *h);
template <typename Kern>
static int allow_big_lds(Kern k, size_t bytes);

// Everything a bucketed pass needs to know about one (sub-)batch.
struct PartPlan {
    int64_t ntiles;
    unsigned tgrid;        // tile-kernel grid: 8 XCD ranges of ceil(ntiles/8) tiles
    unsigned *table;       // [ntiles][PT_NB] destination of every (tile, bucket) run; row 0 = bucket offsets
    int32_t *plan;         // first search workgroup of every bucket
    int2 *bq;              // the queries in bucket order, (qs, qe) pairs
    unsigned short *lpos;  // per query (original order): slot inside its tile's sorted order
};

// Bucket the batch: histogram, column scan of the tile-major table, LDS-ordered scatter.  `sub`/`q0` select the
// scratch regions of a sub-batch (regions are addressed by query offset; sub-batches start on tile boundaries).
static int part_prepare(bxmi_ivl *h, int sub, int64_t q0, const int32_t *qs, const int32_t *qe, int64_t nq, bool want_lpos, hipStream_t st,
                        PartPlan *pp, unsigned *unsorted /* zeroed flag, set by the histogram pass when the starts are not sorted */,
                        bool skip_sorted /* the passes after the histogram exit at once on a sorted batch (count path) */)
{
    const unsigned *gate = skip_sorted ? unsorted : nullptr;
    pp->ntiles = div_up(nq, PT_TILE);
    pp->tgrid = (unsigned)(((pp->ntiles + 7) >> 3) << 3);
    const int rows_per_block = (int)div_up(pp->ntiles, 64);  // ~64 row blocks: the serial middle kernel stays short
    const int nrb = (int)div_up(pp->ntiles, rows_per_block);
    pp->table = h->p_table.as<unsigned>() + (q0 / PT_TILE) * PT_NB;
    unsigned *partial = h->p_hist.as<unsigned>() + (int64_t)sub * 80 * PT_NB;
    pp->plan = h->p_plan.as<int32_t>() + (int64_t)sub * (PT_NB + 8);
    pp->bq = h->p_pairs.as<int2>() + q0;
    pp->lpos = h->p_dest.as<unsigned short>() + q0;  // written by the histogram pass, read by the scatter (and the gather)
    hipLaunchKernelGGL(part_hist_kernel, dim3(pp->tgrid), dim3(PT_THREADS), 0, st, qs, nq, h->geom, pp->table, pp->ntiles,
                       pp->lpos, unsorted);
    hipLaunchKernelGGL(part_colsum_kernel, dim3(nrb), dim3(PT_THREADS), 0, st, pp->table, pp->ntiles, rows_per_block, partial, gate);
    hipLaunchKernelGGL(part_colbase_kernel, dim3(1), dim3(PT_THREADS), 0, st, partial, nrb, nq, pp->plan, gate);
    hipLaunchKernelGGL(part_colscan_kernel, dim3(nrb), dim3(PT_THREADS), 0, st, pp->table, pp->ntiles, rows_per_block, partial, gate);
    BXMI_LAUNCH_CHECK();
    const size_t scat_lds = (size_t)PT_TILE * 8 + PT_NB * sizeof(unsigned);
    hipLaunchKernelGGL(part_scatter_kernel, dim3(pp->tgrid), dim3(PT_THREADS), scat_lds, st, qs, qe, nq, h->geom, pp->table, pp->ntiles, pp->bq,
                       pp->lpos, gate);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// Scratch for bucketing a batch of nq queries (grow-only).
static int part_reserve(bxmi_ivl *h, int64_t nq, bool want_lpos)
{
    BXMI_TRY(h->p_pairs.reserve((size_t)(nq + 4) * 8));
    BXMI_TRY(h->p_plan.reserve((size_t)PT_MAX_SUB * (PT_NB + 8) * sizeof(int32_t)));
    BXMI_TRY(h->p_slots.reserve((size_t)PT_MAX_SUB * PT_SLOT_STRIDE * sizeof(unsigned long long)));
    BXMI_TRY(h->p_table.reserve((size_t)(div_up(nq, PT_TILE) + PT_MAX_SUB) * PT_NB * sizeof(unsigned)));
    BXMI_TRY(h->p_hist.reserve((size_t)PT_MAX_SUB * 80 * PT_NB * sizeof(unsigned)));
    BXMI_TRY(h->p_dest.reserve((size_t)(nq + 8) * 2));
    if (want_lpos) BXMI_TRY(h->p_cnt.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(allow_big_lds(part_scatter_kernel, (size_t)PT_TILE * 8 + PT_NB * sizeof(unsigned)));
    return BXMI_OK;
}

// One sub-batch of the partitioned count, all on stream `st`.
static int ivl_count_part_sub(bxmi_ivl *h, int sub, int64_t q0, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t *counts,
                              int64_t *total_dev, hipStream_t st)
{
    PartPlan pp;
    // [PT_SLOTS partial totals][flag: 1 = the starts are NOT sorted], zeroed together
    unsigned long long *slots = h->p_slots.as<unsigned long long>() + (int64_t)sub * PT_SLOT_STRIDE;
    unsigned *unsorted = g_opt_sorted_path ? reinterpret_cast<unsigned *>(slots + PT_SLOTS) : nullptr;
    BXMI_HIP(hipMemsetAsync(slots, 0, PT_SLOT_STRIDE * sizeof(unsigned long long), st));
    BXMI_TRY(part_prepare(h, sub, q0, qs, qe, nq, counts != nullptr, st, &pp, unsorted, true));
    if (unsorted) {
        // sorted batch: one pass over the queries as they lie (exits at once otherwise)
        TreeDev S = h->treeS.dev, E = h->treeE.dev;
        S.lds_from = S.nlev, S.lds_ints = 0, E.lds_from = E.nlev, E.lds_ints = 0;  // walk the global levels only
        hipLaunchKernelGGL(ivl_local_count_kernel, dim3((unsigned)div_up(nq, LC_CHUNK)), dim3(LC_THREADS), 0, st, S, E, index_dev(h),
                           h->e_sorted.as<int32_t>(), qs, qe, nq, counts, total_dev ? slots : nullptr, unsorted);
    }
    const unsigned grid = (unsigned)(div_up(nq, PT_CHUNK) + PT_NB);
    unsigned short *cnt16 = counts ? h->p_cnt.as<unsigned short>() + q0 : nullptr;  // counts in bucket order, 16 bits + escape
    hipLaunchKernelGGL(part_count_cells_kernel<unsigned short>, dim3(grid), dim3(PT_THREADS), (size_t)PT_LDS_INTS * 4, st, index_dev(h),
                       h->e_sorted.as<int32_t>(), h->slice_bounds.as<SliceBound>(), h->cell_images.as<int32_t>(),
                       h->cell_meta.as<CellsMeta>(), pp.plan, pp.table, pp.bq, nq, h->geom, cnt16, total_dev ? slots : nullptr, unsorted);
    BXMI_LAUNCH_CHECK();
    if (counts) {
        hipLaunchKernelGGL(part_gather_kernel<unsigned short>, dim3(pp.tgrid), dim3(PT_THREADS), 0, st, cnt16, pp.lpos, pp.table, pp.ntiles, nq,
                           counts, unsorted, index_dev(h), h->e_sorted.as<int32_t>(), qs, qe);
        BXMI_LAUNCH_CHECK();
    }
    if (total_dev) {
        hipLaunchKernelGGL(part_fold_total_kernel, dim3(1), dim3(64), 0, st, slots, reinterpret_cast<unsigned long long *>(total_dev));
        BXMI_LAUNCH_CHECK();
    }
    return BXMI_OK;
}

// Large-batch count: bucket the queries, search each bucket against LDS-resident slices, gather back.
// (Cutting the batch into sub-batches on forked streams, so that one sub-batch's search overlaps the next one's scatter,
// measured -2 % at depth 2 and +10 % at depth 4 and was removed.)
static int ivl_count_partitioned(bxmi_ivl *h, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t *counts,
                                 int64_t *total_dev, hipStream_t st)
{
    if (nq >= ((int64_t)1 << 31)) return fail(BXMI_EINVAL, "bxmi_ivl_count: more than 2^31 queries in one batch");
    BXMI_TRY(part_reserve(h, nq, counts != nullptr));
    BXMI_TRY(allow_big_lds(part_count_cells_kernel<unsigned short>, (size_t)PT_LDS_INTS * 4));
    if (!h->images_ready) {
        // LDS images of every bucket for the search (159 MB, a property of the sealed index): built by the first large
        // batch, so the many small per-chromosome trees of the drop-in classes never pay for them
        BXMI_TRY(h->cell_images.reserve((size_t)PT_NB * PT_LDS_INTS * sizeof(int32_t)));
        BXMI_TRY(h->cell_meta.reserve(PT_NB * sizeof(CellsMeta)));
        BXMI_TRY(allow_big_lds(part_cells_image_kernel, (size_t)PT_LDS_INTS * 4));
        hipLaunchKernelGGL(part_cells_image_kernel, dim3(PT_NB), dim3(PT_THREADS), (size_t)PT_LDS_INTS * 4, st, index_dev(h),
                           h->e_sorted.as<int32_t>(), h->slice_bounds.as<SliceBound>(), h->geom, h->cell_images.as<int32_t>(),
                           h->cell_meta.as<CellsMeta>());
        BXMI_LAUNCH_CHECK();
        h->images_ready = true;
    }
    return ivl_count_part_sub(h, 0, 0, qs, qe, nq, counts, total_dev, st);
}


// (end, insertion index) pairs in start order (count_slices.hpp: sl_pack_eid_kernel), once per sealed index.
static int sl_ensure_eid(bxmi_ivl *h, hipStream_t st)
{
    if (h->sl_eid_ready) return BXMI_OK;
    BXMI_TRY(h->sl_eid.reserve(((size_t)h->n + SL_WALK) * sizeof(int2)));
    hipLaunchKernelGGL(sl_pack_eid_kernel, dim3((unsigned)div_up(h->n + SL_WALK, 256)), dim3(256), 0, st, h->e_ord.as<int32_t>(), h->idx.as<int32_t>(),
                       (int)h->n, h->sl_eid.as<int2>());
    BXMI_LAUNCH_CHECK();
    h->sl_eid_ready = true;
    return BXMI_OK;
}

// find() on a batch with sorted starts: windows in query order -> scan -> fill, no bucketing.
static int ivl_find_local(bxmi_ivl *h, const int32_t *qs, const int32_t *qe, int64_t nq, int64_t *offsets, int32_t *hits, int64_t cap,
                          int64_t *total_host, hipStream_t st)
{
    BXMI_TRY(h->p_lo.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->p_hi.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_cnt.reserve((size_t)(nq + 4) * 4));
    // (hi, count) from the sorted-batch count kernel, the offsets from one scan over the chunks' totals, then a walk down the pairs
    const bool chunk_scan = ((uintptr_t)offsets & 15) == 0;  // (lf_offsets_kernel stores 16 bytes at a time)
    {
        TreeDev S = h->treeS.dev, E = h->treeE.dev;
        S.lds_from = S.nlev, S.lds_ints = 0, E.lds_from = E.nlev, E.lds_ints = 0;  // walk the global levels only
        const int64_t nchunks = div_up(nq, LC_CHUNK);
        if (chunk_scan) BXMI_TRY(h->lf_state.reserve((size_t)(2 * nchunks + 4) * 8));
        hipLaunchKernelGGL(ivl_local_count_kernel, dim3((unsigned)nchunks), dim3(LC_THREADS), 0, st, S, E, index_dev(h),
                           h->e_sorted.as<int32_t>(), qs, qe, nq, h->q_cnt.as<int32_t>(), (unsigned long long *)nullptr, (const unsigned *)nullptr,
                           h->p_hi.as<int32_t>(), (unsigned long long *)nullptr, 0ull, chunk_scan ? h->lf_state.as<unsigned long long>() : nullptr);
        if (chunk_scan) {
            long long *chunk_base = h->lf_state.as<long long>() + nchunks;  // [nchunks + 1]
            hipLaunchKernelGGL(fx_tile_scan_kernel, dim3(1), dim3(1024), 0, st, h->lf_state.as<unsigned long long>(), nchunks, chunk_base,
                               reinterpret_cast<long long *>(offsets) + nq);
            hipLaunchKernelGGL(lf_offsets_kernel, dim3((unsigned)nchunks), dim3(LC_THREADS), 0, st, h->q_cnt.as<int32_t>(), chunk_base, nq,
                               reinterpret_cast<long long *>(offsets));
        }
    }
    BXMI_LAUNCH_CHECK();
    if (!chunk_scan)
        BXMI_TRY((device_scan<int32_t, long long, OpSum, false>(h->q_cnt.as<int32_t>(), reinterpret_cast<long long *>(offsets), nq, 0ll,
                                                               reinterpret_cast<long long *>(offsets) + nq, h->scan_scratch, st)));
    int64_t total = 0;
    BXMI_HIP(hipMemcpyAsync(&total, offsets + nq, 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    if (total_host) *total_host = total;
    if (total > cap) return fail(BXMI_ERANGE, "bxmi_ivl_find: %lld hits need a larger buffer than cap=%lld", (long long)total, (long long)cap);
    if (total == 0) return BXMI_OK;
    BXMI_TRY(sl_ensure_eid(h, st));
    hipLaunchKernelGGL(part_fill_flat_kernel, dim3(device_props().cus * 8), dim3(FIND_THREADS), 0, st, h->sl_eid.as<int2>() + SL_WALK, qs, nq,
                       h->p_hi.as<int32_t>(), h->q_cnt.as<int32_t>(), reinterpret_cast<const long long *>(offsets), hits);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// Partitioned find(): same bucketing as the count path, then window+count per query in bucket order, counts gathered
// back for the CSR offsets, offsets carried to bucket order, hits written from bucket order.
static int ivl_find_partitioned(bxmi_ivl *h, const int32_t *qs, const int32_t *qe, int64_t nq, int64_t *offsets, int32_t *hits,
                                int64_t cap, int64_t *total_host, hipStream_t st)
{
    if (nq >= ((int64_t)1 << 31)) return fail(BXMI_EINVAL, "bxmi_ivl_find: more than 2^31 queries in one batch");
    BXMI_TRY(part_reserve(h, nq, true));
    BXMI_TRY(h->p_lo.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->p_hi.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_cnt.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->p_boffs.reserve((size_t)(nq + 4) * 8));
    PartPlan pp;
    unsigned *unsorted = reinterpret_cast<unsigned *>(h->p_slots.as<unsigned long long>() + PT_SLOTS);  // hint only on this path
    BXMI_HIP(hipMemsetAsync(unsorted, 0, sizeof(unsigned), st));
    BXMI_TRY(part_prepare(h, 0, 0, qs, qe, nq, true, st, &pp, unsorted, false));
    const int64_t ntiles = pp.ntiles;
    const unsigned tgrid = pp.tgrid;
    unsigned *table = pp.table;
    unsigned short *lpos = pp.lpos;
    const size_t lds_bytes = (size_t)PT_LDS_INTS * 4;
    BXMI_TRY(allow_big_lds(part_window_kernel, lds_bytes));
    const unsigned grid = (unsigned)(div_up(nq, PT_CHUNK) + PT_NB);
    hipLaunchKernelGGL(part_window_kernel, dim3(grid), dim3(PT_THREADS), lds_bytes, st, index_dev(h), h->slice_bounds.as<SliceBound>(), pp.plan,
                       table, pp.bq, nq, h->p_lo.as<int32_t>(), h->p_hi.as<int32_t>(), h->p_cnt.as<int32_t>());
    hipLaunchKernelGGL(part_gather_kernel<int32_t>, dim3(tgrid), dim3(PT_THREADS), 0, st, h->p_cnt.as<int32_t>(), lpos, table, ntiles, nq,
                       h->q_cnt.as<int32_t>(), (const unsigned *)nullptr, index_dev(h), (const int32_t *)nullptr, (const int32_t *)nullptr,
                       (const int32_t *)nullptr);
    BXMI_LAUNCH_CHECK();
    BXMI_TRY((device_scan<int32_t, long long, OpSum, false>(h->q_cnt.as<int32_t>(), reinterpret_cast<long long *>(offsets), nq, 0ll,
                                                           reinterpret_cast<long long *>(offsets) + nq, h->scan_scratch, st)));
    int64_t total = 0;
    BXMI_HIP(hipMemcpyAsync(&total, offsets + nq, 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    if (total_host) *total_host = total;
    if (total > cap) return fail(BXMI_ERANGE, "bxmi_ivl_find: %lld hits need a larger buffer than cap=%lld", (long long)total, (long long)cap);
    if (total == 0) return BXMI_OK;
    const size_t perm_lds = (size_t)PT_TILE * 8 + (PT_NB + 2) * 2 + PT_NB * 4 + 64;
    BXMI_TRY(allow_big_lds(part_permute_i64_kernel, perm_lds));
    hipLaunchKernelGGL(part_permute_i64_kernel, dim3(tgrid), dim3(PT_THREADS), perm_lds, st, reinterpret_cast<const long long *>(offsets), lpos,
                       table, ntiles, nq, h->p_boffs.as<long long>());
    int fgrid = device_props().cus * 8;
    hipLaunchKernelGGL(part_fill_kernel, dim3(fgrid), dim3(FIND_THREADS), 0, st, index_dev(h), reinterpret_cast<const int32_t *>(pp.bq), 2, nq,
                       h->p_lo.as<int32_t>(),
                       h->p_hi.as<int32_t>(), h->p_cnt.as<int32_t>(), h->p_boffs.as<long long>(), hits);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// ---- the large-batch count pass (count_bitmap.hpp: tile sort, run table, plan, un-permute; its search stages in count_dense.hpp / count_slices.hpp) ----
// Cell images of 2^18-coordinate units for the flat walk (count_dense.hpp, bp_*): built once per sealed index.
static int bp_prepare_index(bxmi_ivl *h, hipStream_t st)
{
    h->bp_state = -1;
    const int shift = h->geom.shift;
    // (units of at least two buckets: the 1024-thread tile sorts can then start every unit's run on a whole 16-byte slot)
    if (h->has_reversed || h->n < 4096 || shift > BP_UNIT_LOG2 - 1 || shift < BM_MIN_SHIFT) return BXMI_OK;
    BmGeom g;
    g.cmin = h->geom.cmin;
    g.cmax = h->cmax;
    g.shift = shift;
    const int f = BP_UNIT_LOG2 - shift;
    g.f = f > BD_MAX_F ? BD_MAX_F : f;
    g.rshift = BP_RSHIFT;
    g.dshift = 0;
    const BpLayout L = bp_layout(g.shift + g.f);
    g.nce = L.nce, g.ncs = L.ncs;
    g.stride = L.bytes >> 4;
    const int units = BM_NB >> g.f;
    BXMI_TRY(h->bp_images.reserve((size_t)units * L.bytes));
    BXMI_TRY(h->bp_stats.reserve(64));
    BXMI_HIP(hipMemsetAsync(h->bp_stats.p, 0, 64, st));
    const size_t lds = (size_t)4 * L.ncs * sizeof(int32_t);
    BXMI_TRY(allow_big_lds(bp_image_kernel, lds));
    hipLaunchKernelGGL(bp_image_kernel, dim3((unsigned)units), dim3(BD_THREADS), lds, st, h->s_ord.as<int32_t>(), h->e_sorted.as<int32_t>(), (int)h->n, g,
                       h->bp_images.as<unsigned char>(), h->bp_stats.as<unsigned>());
    BXMI_LAUNCH_CHECK();
    unsigned stats[2] = {0, 0};
    BXMI_HIP(hipMemcpyAsync(stats, h->bp_stats.p, sizeof(stats), hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    h->bp_geom = g;
    h->bp_hard_cells = stats[0];
    // cells that queries can land in: the span of the index, twice (ends and starts)
    const int64_t cells = 2 * ((((int64_t)h->cmax - (int64_t)h->geom.cmin) >> 5) + 1);
    if (stats[1] == 0 && (int64_t)stats[0] * 1000000 <= cells * g_opt_bm_hard_ppm) h->bp_state = 1;
    return BXMI_OK;
}

// Offset-cell images (offset_cells.hpp) for a sparse index: the cell width from the density, units of 4096 cells (fewer when the
// span is small: a unit is at most 2^BD_MAX_F buckets), built once per sealed index.
static int bo_prepare_index(bxmi_ivl *h, hipStream_t st)
{
    h->bo_state = -1;
    const int shift = h->geom.shift;
    int64_t span = (int64_t)h->cmax - (int64_t)h->geom.cmin + 1;
    const int k = g_opt_bo_cell_log2 ? (int)g_opt_bo_cell_log2 : bo_cell_log2_for(span, h->n);
    // (units of at least two buckets, as for bitmap cells: the tile sort can then start every unit's run on a whole slot)
    if (h->has_reversed || h->n < 4096 || k == 0 || shift > bo_rshift(k) - 1 || shift < BM_MIN_SHIFT) return BXMI_OK;
    BmGeom g;
    g.cmin = h->geom.cmin;
    g.cmax = h->cmax;
    g.shift = shift;
    const int f = bo_rshift(k) - shift;
    g.f = f > BD_MAX_F ? BD_MAX_F : f;
    g.rshift = bo_rshift(k);
    g.dshift = k - 5;
    const BpLayout L = bp_layout(g.shift + g.f, k);
    g.nce = L.nce, g.ncs = L.ncs;
    g.stride = L.bytes >> 4;
    const int units = BM_NB >> g.f;
    BXMI_TRY(h->bo_images.reserve((size_t)units * L.bytes));
    BXMI_TRY(h->bp_stats.reserve(64));
    BXMI_HIP(hipMemsetAsync(h->bp_stats.p, 0, 64, st));
    const size_t lds = (size_t)L.ncs * sizeof(int32_t);
    BXMI_TRY(allow_big_lds(bo_image_kernel, lds));
    hipLaunchKernelGGL(bo_image_kernel, dim3((unsigned)units), dim3(BD_THREADS), lds, st, h->s_ord.as<int32_t>(), h->e_sorted.as<int32_t>(), (int)h->n, g,
                       h->bo_images.as<unsigned char>(), h->bp_stats.as<unsigned>());
    BXMI_LAUNCH_CHECK();
    unsigned stats[2] = {0, 0};
    BXMI_HIP(hipMemcpyAsync(stats, h->bp_stats.p, sizeof(stats), hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    h->bo_geom = g;
    h->bo_hard_cells = stats[0];
    const int64_t cells = 2 * ((span >> k) + 1);  // cells that queries can land in: the span of the index, ends and starts
    if (stats[1] == 0 && (int64_t)stats[0] * 1000000 <= cells * g_opt_bm_hard_ppm) h->bo_state = 1;
    return BXMI_OK;
}

// Dense unit images (count_dense.hpp): built once per sealed index; the kernel reports whether the index fits the format.
static int bd_prepare_index(bxmi_ivl *h, hipStream_t st)
{
    h->bd_state = -1;
    const int shift = h->geom.shift;
    if (h->has_reversed || h->n < 4096 || shift > BD_MAX_SHIFT || shift < BM_MIN_SHIFT) return BXMI_OK;
    // Units of 2^19 coordinates first (runs twice as long, 12 KiB of LDS for duplicated coordinates: enough for an index
    // whose duplicates are accidents), then units of 2^18 (64 KiB: rank tables for clumped cells).
    int tried = -1;  // unit width (log2) of the geometry tried last
    for (int ulog = g_opt_bd_unit_log2 ? (int)g_opt_bd_unit_log2 : BD_UNIT_LOG2; ulog >= 18 || ulog == (int)g_opt_bd_unit_log2; ulog--) {
        BmGeom g;
        g.cmin = h->geom.cmin;
        g.cmax = h->cmax;
        g.shift = shift;
        const int f = ulog - shift;
        g.f = f < 0 ? 0 : (f > BD_MAX_F ? BD_MAX_F : f);
        if (g.shift + g.f == tried) break;  // (buckets wider than the unit asked for: f clamps to 0, the same images again)
        tried = g.shift + g.f;
        g.rshift = BD_RSHIFT;
        g.dshift = 0;
        const BdLayout L = bd_layout(g.shift + g.f);
        g.nce = L.nce, g.ncs = L.ncs;
        g.stride = L.bytes >> 4;
        const int units = BM_NB >> g.f;
        BXMI_TRY(h->bd_images.reserve((size_t)units * L.bytes));
        BXMI_TRY(h->bd_stats.reserve(64));
        const size_t lds = (size_t)8 * L.ncs * sizeof(int32_t);
        BXMI_TRY(allow_big_lds(bd_image_kernel, lds));
        h->bd_geom = g;
        // ranks relative to the whole unit when every slice holds fewer than 2^15 keys (no table read per lookup), else
        // relative to blocks of 1024 cells
        // rank tables for every cell with two duplicated coordinates where the overflow area has the room, else from six (count_dense.hpp)
        const int tf_first = g_opt_bd_table_from ? (int)g_opt_bd_table_from : BD_TABLE_FROM_FIRST;
        const int tf_last = g_opt_bd_table_from ? (int)g_opt_bd_table_from : BD_TABLE_FROM_LAST;
        for (int table_from = tf_first; table_from <= tf_last; table_from += BD_TABLE_FROM_LAST - BD_TABLE_FROM_FIRST)
        for (int bshift = g_opt_bd_blocks ? 10 : 13; bshift >= 10; bshift -= 3) {
            BXMI_HIP(hipMemsetAsync(h->bd_stats.p, 0, 64, st));
            hipLaunchKernelGGL(bd_image_kernel, dim3((unsigned)units), dim3(BD_THREADS), lds, st, h->s_ord.as<int32_t>(), h->e_sorted.as<int32_t>(),
                               (int)h->n, g, bshift, h->bd_images.as<unsigned char>(), h->bd_stats.as<unsigned>(), table_from);
            BXMI_LAUNCH_CHECK();
            BXMI_HIP(hipMemcpyAsync(h->bd_worst, h->bd_stats.p, sizeof(h->bd_worst), hipMemcpyDeviceToHost, st));
            BXMI_HIP(hipStreamSynchronize(st));
            h->bd_blocks = bshift == 10;
            if (h->bd_worst[1] > (unsigned)L.ov_cap) break;  // too many duplicated coordinates: blocks do not help
            if (h->bd_worst[0] <= 32767u) {
                h->bd_state = 1;
                return BXMI_OK;
            }
        }
        if (g_opt_bd_unit_log2 || g.shift + g.f < ulog) break;  // forced, or the span is so small that the unit cannot shrink with ulog
    }
    return BXMI_OK;
}

// Slice search: the ranks at every bucket boundary, and how many keys a unit of 2^f buckets would have to stage.
static int sl_prepare_index(bxmi_ivl *h, hipStream_t st)
{
    h->sl_state = -1;
    if (h->has_reversed || h->n < 1) return BXMI_OK;
    BXMI_TRY(h->sl_meta.reserve((size_t)(BM_NB + 1) * sizeof(int4)));
    BXMI_TRY(h->sl_stats.reserve(64));
    BXMI_HIP(hipMemsetAsync(h->sl_stats.p, 0, 64, st));
    hipLaunchKernelGGL(sl_meta_kernel, dim3((BM_NB + 1 + 255) / 256), dim3(256), 0, st, h->s_ord.as<int32_t>(), h->e_sorted.as<int32_t>(), (int)h->n,
                       h->geom.cmin, h->geom.shift, h->sl_meta.as<int4>());
    hipLaunchKernelGGL(sl_fit_kernel, dim3(1), dim3(1024), 0, st, h->sl_meta.as<int4>(), h->sl_stats.as<unsigned>());
    BXMI_LAUNCH_CHECK();
    BXMI_HIP(hipMemcpyAsync(h->sl_need, h->sl_stats.p, sizeof(h->sl_need), hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    if (h->sl_need[0] <= (unsigned)SL_CAP) h->sl_state = 1;
    return BXMI_OK;
}

// The slice geometry of one index for a batch with `tile` queries per tile: the unit grows while its keys fit, its
// offsets leave 12 bits for the record's length, and its runs stay short enough for one pass of a wave.
static BmGeom sl_geom(const bxmi_ivl *h, int64_t tile, size_t *lds_bytes, int64_t *run_len)
{
    BmGeom g;
    g.cmin = h->geom.cmin, g.cmax = h->cmax, g.shift = h->geom.shift;
    const int64_t span = (int64_t)h->cmax - (int64_t)h->geom.cmin;
    const int64_t nb_used = ((span > 0 ? span : 0) >> g.shift) + 1;
    int f = 0;
    if (g_opt_sl_f >= 0) {
        for (f = (int)g_opt_sl_f; f > 0 && h->sl_need[f] > (unsigned)SL_CAP; f--) {}
    } else {
        for (int k = 1; k <= SL_MAX_F; k++) {
            // ... and while a (tile, unit) run of a uniform batch stays within ~2.5 waves: longer runs go through the
            // leftover passes / the workgroup's long-run list, which cost small chromosomes (narrow buckets, f = 5)
            // 15 % of the genome pass
            // (the flat walk keeps every lane busy whatever the run length, but a unit's directory has 2047 cells however
            // wide the unit is: a larger unit means more keys per cell and more halvings per lookup -- measured: the genome
            // pass 1.0 -> 2.4 ms with units as large as the LDS allows -- so the same cap serves both walks)
            if (h->sl_need[k] > (unsigned)SL_CAP || g.shift + k > g_opt_sl_rbits || (tile << k) / nb_used > g_opt_sl_run_cap) break;
            f = k;
        }
    }
    g.f = f;
    g.rshift = g.shift + f > 17 ? g.shift + f : 17;
    const int64_t Wu = (int64_t)1 << (g.shift + f);
    int d = 0;
    while (((Wu + SL_MARGIN) >> d) + 1 > SL_DIR_CELLS) d++;
    g.dshift = d;
    g.ncs = (int32_t)((Wu + SL_MARGIN) >> d) + 1;
    g.nce = (int32_t)(Wu >> d) + 1;
    g.stride = 0;
    *lds_bytes = 2 * ((size_t)h->sl_need[f] + 16) + 2 * (size_t)((g.ncs + 2 + 7) & ~7) + 2 * (size_t)(g.nce + 2 + 8);
    *run_len = (tile << f) / nb_used;
    return g;
}

// Everything the kernels of one batch need, as they are handed to every launch.
struct BmLaunch {
    const BmSeg *segs;             // device: the batch's segments
    const unsigned short *tile_seg;  // device: segment of every tile (padded numbering)
    bxmi_ivl *owner;               // whose scratch the batch uses
    int64_t ntp;                   // tiles in the padded numbering (a multiple of BM_GROUP_TILES)
    int ngroups, tile_log2;
    size_t search_lds;
    const unsigned *gate;
    bool pad = false;  // the units' runs on whole 16-byte slots (bm_tile_sort_kernel<.., PAD>): tile stride TILE + BM_PAD_ROOM
    bool w8 = false;   // 8-bit counts between the search and the un-permute kernel (padded layout, cell images)
    bool wide = false; // the cell images are offset cells (sparse indexes)
    unsigned *descent = nullptr;  // no order check in this pass: bm_params_kernel's probe raises this word when it sees a descent
    unsigned *xcd_next = nullptr;  // eight item counters of the persistent search, zeroed with the partial totals
    // the parameter block written by the tile sort's first workgroup (count_bitmap.hpp: bm_write_params) instead of bm_params_kernel
    BmSegChunk par;
    int npar = 0;
    BmParOut par_out;
    int n_segs = 1;
};

template <int THREADS, int ITEMS>
static int bm_launch_tiles(const BmLaunch &L, hipStream_t st, bool sub = false)
{
    constexpr int TILE = THREADS * ITEMS;
    bxmi_ivl *h = L.owner;
    if (sub) {  // find(): ordered by half buckets, both tables (find_exchange.hpp)
        if (THREADS != 1024 || L.pad) return fail(BXMI_ESTATE, "bm_launch_tiles: half buckets need a 1024-thread shape on packed runs");
        constexpr int T2 = THREADS == 1024 ? THREADS : 1024;  // (only the 1024-thread shapes instantiate the kernel)
        constexpr int I2 = TILE / T2;
        const size_t lds = (size_t)TILE * 4 + FX_NBK * 4 + FX_NBK * 2 + 64;
        BXMI_TRY(allow_big_lds((bm_tile_sort_kernel<T2, I2, false, 2>), lds));
        hipLaunchKernelGGL((bm_tile_sort_kernel<T2, I2, false, 2>), dim3((unsigned)L.ntp), dim3(T2), lds, st, L.segs, L.tile_seg,
                           h->bm_recs.as<unsigned>(), h->bm_slots.as<unsigned short>(), h->bm_tbl.as<unsigned short>(), L.gate, (unsigned *)nullptr,
                           h->fx_tbl2.as<unsigned short>(), L.par, L.npar, L.par_out);
        BXMI_LAUNCH_CHECK();
        return BXMI_OK;
    }
    if (L.pad) {
        const size_t lds = (size_t)(TILE + 3 * THREADS) * 4 + BM_NB * 4 + BM_NB * 2 + 64;
        BXMI_TRY(allow_big_lds((bm_tile_sort_kernel<THREADS, ITEMS, true>), lds));
        hipLaunchKernelGGL((bm_tile_sort_kernel<THREADS, ITEMS, true>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, L.segs, L.tile_seg,
                           h->bm_recs.as<unsigned>(), h->bm_slots.as<unsigned short>(), h->bm_tbl.as<unsigned short>(), L.gate, h->bd_tend.as<unsigned>(),
                           (unsigned short *)nullptr, L.par, L.npar, L.par_out);
    } else {
        const size_t lds = (size_t)TILE * 4 + BM_NB * 4 + BM_NB * 2 + 64;
        BXMI_TRY(allow_big_lds((bm_tile_sort_kernel<THREADS, ITEMS, false>), lds));
        hipLaunchKernelGGL((bm_tile_sort_kernel<THREADS, ITEMS, false>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, L.segs, L.tile_seg,
                           h->bm_recs.as<unsigned>(), h->bm_slots.as<unsigned short>(), h->bm_tbl.as<unsigned short>(), L.gate, (unsigned *)nullptr,
                           (unsigned short *)nullptr, L.par, L.npar, L.par_out);
    }
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

template <int THREADS, int ITEMS>
static int bm_launch_unpermute(const BmLaunch &L, unsigned long long *slots, hipStream_t st, const unsigned *cnt = nullptr, unsigned *loff = nullptr,
                               int fx = 0 /* 0, or FIND = 2 / 3 */)
{
    bxmi_ivl *h = L.owner;
    const size_t lds = (size_t)THREADS * ITEMS * sizeof(unsigned);
    if (!cnt) cnt = h->bm_recs.as<unsigned>();
    if (loff && fx == 3) {
        BXMI_TRY(allow_big_lds((bm_unpermute_kernel<THREADS, ITEMS, 3>), lds));
        hipLaunchKernelGGL((bm_unpermute_kernel<THREADS, ITEMS, 3>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, cnt,
                           h->bm_slots.as<unsigned short>(), L.segs, L.tile_seg, slots, L.gate, loff, h->fx_svq.as<unsigned>(),
                           h->fx_parts.as<unsigned long long>(), h->fx_tile_tot.as<unsigned long long>());
    } else if (loff && fx) {
        BXMI_TRY(allow_big_lds((bm_unpermute_kernel<THREADS, ITEMS, 2>), lds));
        hipLaunchKernelGGL((bm_unpermute_kernel<THREADS, ITEMS, 2>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, cnt,
                           h->bm_slots.as<unsigned short>(), L.segs, L.tile_seg, slots, L.gate, loff, h->fx_svq.as<unsigned>(),
                           h->fx_parts.as<unsigned long long>(), h->fx_tile_tot.as<unsigned long long>());
    } else {
        BXMI_TRY(allow_big_lds((bm_unpermute_kernel<THREADS, ITEMS>), lds));
        hipLaunchKernelGGL((bm_unpermute_kernel<THREADS, ITEMS>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, cnt,
                           h->bm_slots.as<unsigned short>(), L.segs, L.tile_seg, slots, L.gate, (unsigned *)nullptr);
    }
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// The bitmap-cell pass over a batch of n segments (n sealed, qualifying indexes with their queries): [order check ->]
// tile sort (with the batch's parameter block) -> run table -> plan -> search -> un-permute -> totals, all on `st`, six launches
// for up to 16 segments (five when nobody asks for totals); with the order check in front or more segments the parameter kernel
// is a launch of its own.
// counts[i] may be NULL (total only: nothing is stored per query); totals_dev[i] may be.  The scratch of hs[0] serves the whole batch.
#ifndef SL_FIND_U
#define SL_FIND_U 2  // runs per lane group and round of find()'s count half
#endif
template <int LANES>
static int sl_launch_search(const BmLaunch &L, unsigned grid, hipStream_t st, unsigned *out, unsigned *hc = nullptr)
{
    bxmi_ivl *h = L.owner;
    if (out != h->bm_recs.as<unsigned>()) {
        BXMI_TRY(allow_big_lds((sl_search_pipe_kernel<LANES, SL_FIND_U, true>), L.search_lds));
        hipLaunchKernelGGL((sl_search_pipe_kernel<LANES, SL_FIND_U, true>), dim3(grid), dim3(SL_THREADS), L.search_lds, st, L.segs, h->bm_items.as<int4>() + 1,
                           h->bm_items.as<int>(), h->bm_runT.as<unsigned>(), L.ntp, h->bm_recs.as<unsigned>(), out, L.tile_log2, L.gate, hc);
    } else {
        BXMI_TRY(allow_big_lds((sl_search_pipe_kernel<LANES, 2, false>), L.search_lds));
        hipLaunchKernelGGL((sl_search_pipe_kernel<LANES, 2, false>), dim3(grid), dim3(SL_THREADS), L.search_lds, st, L.segs, h->bm_items.as<int4>() + 1,
                           h->bm_items.as<int>(), h->bm_runT.as<unsigned>(), L.ntp, h->bm_recs.as<unsigned>(), (unsigned *)nullptr, L.tile_log2, L.gate);
    }
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// find() through the exchange (count_slices.hpp): what the count half leaves behind for the fill half.
struct BmFindCtx {
    BmLaunch L;      // the batch as launched (items, run table and records stay in the owner's scratch)
    unsigned sgrid;
    int lanes;       // 16 or 64
    int variant;     // tile shape
    bool sub = false;  // in: the tile sort orders by half buckets and everything find_exchange.hpp needs is left behind
    bool direct = false;  // in (with sub): the offsets the un-permute kernel leaves are query-order prefixes (FIND = 3)
    int f = 0;         // out: the unit size the slice geometry picked (2^f buckets)
    int64_t ntiles = 0;
};

static int sl_launch_search_flat(const BmLaunch &L, unsigned grid, hipStream_t st)
{
    bxmi_ivl *h = L.owner;
    BXMI_TRY(allow_big_lds((sl_search_flat_kernel<4>), L.search_lds));
    hipLaunchKernelGGL((sl_search_flat_kernel<4>), dim3(grid), dim3(SL_THREADS), L.search_lds, st, L.segs, h->bm_items.as<int4>() + 1,
                       h->bm_items.as<int>(), h->bm_runT.as<unsigned>(), L.ntp, h->bm_recs.as<unsigned>(), L.tile_log2, L.gate);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

template <int FMT, bool QB, int EXP, int DEPTH, bool PIPE, bool PAD = false, bool W8 = false>
static int bd_launch_search_t(const BmLaunch &L, unsigned grid, hipStream_t st)
{
    bxmi_ivl *h = L.owner;
    BXMI_TRY(allow_big_lds((bd_search_kernel<FMT, QB, EXP, DEPTH, PIPE, PAD, W8>), L.search_lds));
    hipLaunchKernelGGL((bd_search_kernel<FMT, QB, EXP, DEPTH, PIPE, PAD, W8>), dim3(grid), dim3(BD_THREADS), L.search_lds, st, L.segs,
                       h->bm_items.as<int4>() + 1, h->bm_items.as<int>(), h->bd_unitT.as<unsigned short>(), L.ntp, h->bm_recs.as<unsigned>(),
                       h->bd_cnt16.as<unsigned short>(), L.tile_log2, L.gate);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// the persistent walk on cell images (count_dense.hpp, bw_*): one workgroup per CU, items handed out per XCD
template <bool W8, bool WIDE>
static int bw_launch_search(const BmLaunch &L, hipStream_t st)
{
    bxmi_ivl *h = L.owner;
    constexpr int THREADS = WIDE ? BD_THREADS / 2 : BD_THREADS;  // offset cells: two workgroups per CU
    constexpr int DEPTH = 3;  // (offset cells with rings of 2 / 3 / 4: genome pass 0.719 / 0.722 / 0.705 ms, an eighth of it 0.150 / 0.150 / 0.152)
    BXMI_TRY(allow_big_lds((bw_search_kernel<W8, DEPTH, WIDE, THREADS>), L.search_lds));
    hipLaunchKernelGGL((bw_search_kernel<W8, DEPTH, WIDE, THREADS>), dim3(WIDE ? 512 : 256), dim3(THREADS), L.search_lds, st, L.segs, h->bm_items.as<int4>() + 1,
                       h->bm_items.as<int>(), h->bd_unitT.as<unsigned short>(), L.ntp, h->bm_recs.as<unsigned>(), h->bd_cnt16.as<unsigned short>(),
                       L.tile_log2, L.gate, L.xcd_next);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// One shape per stage and layout (round 3 kept every ring depth, both pipelines and the diagnostics behind knobs: 66 kernels):
//   cell images   always on padded runs (bp_prepare_index only takes geometries the tile sort can pad): the persistent walk;
//   dense images  padded runs: the ring of three hand-issued loads; packed runs (units of a single bucket): two sets of four;
//   key slices    the lean shape (two passes per round, compiler-issued loads: 64 registers) -- two workgroups share a CU when
//                 the units are small, one stages its unit while the other searches (a third of a sparse index's search time).
static int bd_launch_search(const BmLaunch &L, unsigned grid, int fmt /* 0 dense, 1 cells, 2 slices */, bool blocks, hipStream_t st)
{
    if (fmt == 1) {
        if (!L.pad) return fail(BXMI_ESTATE, "bd_launch_search: cell images on packed runs");
        if (L.wide) return L.w8 ? bw_launch_search<true, true>(L, st) : bw_launch_search<false, true>(L, st);
        return L.w8 ? bw_launch_search<true, false>(L, st) : bw_launch_search<false, false>(L, st);
    }
    if (fmt == 2) return bd_launch_search_t<2, false, 0, 2, false>(L, grid, st);  // (never padded: see bm_count_segments)
    if (L.pad) return blocks ? bd_launch_search_t<0, true, 0, 3, true, true>(L, grid, st) : bd_launch_search_t<0, false, 0, 3, true, true>(L, grid, st);
    return blocks ? bd_launch_search_t<0, true, 0, 4, true>(L, grid, st) : bd_launch_search_t<0, false, 0, 4, true>(L, grid, st);
}

template <int THREADS, int ITEMS>
static int bd_launch_unpermute(const BmLaunch &L, unsigned long long *slots, hipStream_t st)
{
    bxmi_ivl *h = L.owner;
    if (L.pad && L.w8) {
        const size_t lds = (size_t)(THREADS * ITEMS + BM_PAD_ROOM);
        BXMI_TRY(allow_big_lds((bd_unpermute_kernel<THREADS, ITEMS, true, true>), lds));
        hipLaunchKernelGGL((bd_unpermute_kernel<THREADS, ITEMS, true, true>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, h->bd_cnt16.as<unsigned short>(),
                           h->bm_slots.as<unsigned short>(), L.segs, L.tile_seg, slots, L.gate, h->bd_tend.as<unsigned>(),
                           h->bd_fb.as<unsigned long long>(), h->bd_fb_host);
    } else if (L.pad) {
        const size_t lds = (size_t)(THREADS * ITEMS + BM_PAD_ROOM) * sizeof(unsigned short);
        BXMI_TRY(allow_big_lds((bd_unpermute_kernel<THREADS, ITEMS, true>), lds));
        hipLaunchKernelGGL((bd_unpermute_kernel<THREADS, ITEMS, true>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, h->bd_cnt16.as<unsigned short>(),
                           h->bm_slots.as<unsigned short>(), L.segs, L.tile_seg, slots, L.gate, h->bd_tend.as<unsigned>());
    } else {
        const size_t lds = (size_t)THREADS * ITEMS * sizeof(unsigned short);
        BXMI_TRY(allow_big_lds((bd_unpermute_kernel<THREADS, ITEMS, false>), lds));
        hipLaunchKernelGGL((bd_unpermute_kernel<THREADS, ITEMS, false>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, h->bd_cnt16.as<unsigned short>(),
                           h->bm_slots.as<unsigned short>(), L.segs, L.tile_seg, slots, L.gate, (const unsigned *)nullptr);
    }
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// Two 64-bit words of host memory the kernels of a pass write for the NEXT calls on this handle (never waited for):
// [0] counts that did not fit 8 bits so far, [1] what the latest order check found.
static int ensure_feedback(bxmi_ivl *h, hipStream_t st)
{
    if (h->bd_fb_host) return BXMI_OK;
    BXMI_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->bd_fb_host), 64, hipHostMallocDefault));
    memset(h->bd_fb_host, 0, 64);
    BXMI_TRY(h->bd_fb.reserve(64));
    BXMI_HIP(hipMemsetAsync(h->bd_fb.p, 0, 64, st));
    return BXMI_OK;
}

// `kind`: what a search workgroup keeps in LDS -- 2 = key slices (count_slices.hpp), 3 = dense unit images, 4 = cell images of
// units (both count_dense.hpp: the flat walk, counts out of place); every index of the batch must have qualified for it.
// (kind 1 -- images of single buckets / bucket pairs, round 2's search -- is gone: an index it served qualifies for the cell
// images of units as well, so no input selected it any more.)
static int bm_count_segments(bxmi_ivl *const *hs, int n, const int32_t *const *qs, const int32_t *const *qe, const int64_t *nq,
                             int32_t *const *counts, int64_t *const *totals_dev, hipStream_t st, int kind, BmFindCtx *fx = nullptr)
{
    if (kind < 2 || kind > 5) return fail(BXMI_EINVAL, "bm_count_segments: no such search stage (%d)", kind);
    const bool wide = kind == 5;  // offset cells: the cell images of sparse indexes
    const bool slices = kind == 2, cells = kind == 4 || wide;
    const bool fxsub = fx && fx->sub;
    // (find() needs 32-bit counts apart from the records and the tile-sorted offsets: the flat walk has that form for find_exchange.hpp only)
    const bool slices_flat = slices && !fx && g_opt_sl_flat != 0;
    const bool dense = kind == 3 || cells || slices_flat /* the flat walk */;
    bxmi_ivl *h = hs[0];
    int64_t nq_all = 0;
    for (int i = 0; i < n; i++) nq_all += nq[i];
    if (nq_all >= ((int64_t)1 << 31)) return fail(BXMI_EINVAL, "bxmi_ivl_count: more than 2^31 queries in one batch");
    if (n > 4096) return fail(BXMI_EINVAL, "bxmi_ivl_count_multi: more than 4096 indexes in one batch");
    // tile shape: 32768-query tiles halve the number of (tile, bucket) runs the search has to fetch, but their sort
    // kernel runs one workgroup per CU and wants a grid of several hundred full tiles
    // (a batch over several indexes: every segment starts on a group of 64 tiles, so the big tiles only where the segments are
    // big too -- a genome of 100 M queries, not its eighth on one of eight GPUs)
    int variant = g_opt_bm_variant >= 0 ? (int)g_opt_bm_variant : (nq_all >= ((int64_t)32 << 20) * (n == 1 ? 1 : 2) ? 2 : 0);
    // cell images are searched on padded runs only: a unit of two buckets needs a tile sort whose threads own two buckets each
    // (the 1024-thread shapes), the 512-thread shape owns four
    // ... and the 1024-thread shape is the faster sort for cell images whatever the unit (a rank's share of a genome, offset cells,
    // f >= 2: 0.413 / 0.241 / 0.138 ms for 50 / 25 / 13 M queries against 0.432 / 0.259 / 0.147 with 512 threads x 32 queries)
    if (cells && variant == 0 && g_opt_bm_variant < 0) variant = 1;
    if (cells && variant == 0)
        for (int i = 0; i < n; i++)
            if ((wide ? hs[i]->bo_geom : hs[i]->bp_geom).f < 2) variant = 1;
    if (fxsub && (n != 1 || !slices)) return fail(BXMI_ESTATE, "bm_count_segments: the half-bucket order serves find() on one index's slices");
    if (fxsub && variant == 0) variant = 1;  // (the half-bucket tile sort has the 1024-thread shapes only)
    const int tile_log2 = variant == 2 ? 15 : 14;
    const int64_t tile = (int64_t)1 << tile_log2;
    // the batch's tile numbering: every segment starts on a plan-group boundary
    std::vector<BmSeg> segs((size_t)n);
    int64_t ntp = 0;
    size_t max_stride = 0, sl_lds = 0;
    int64_t sl_run = INT64_MAX;  // shortest expected (tile, unit) run of the batch
    bool any_total = false, any_blocks = false;
    for (int i = 0; i < n; i++) {
        BmSeg &sg = segs[(size_t)i];
        any_blocks |= kind == 3 && hs[i]->bd_blocks;
        if (slices) {
            size_t lds = 0;
            int64_t run_len = 0;
            sg.g = sl_geom(hs[i], tile, &lds, &run_len);
            if (lds > sl_lds) sl_lds = lds;
            if (run_len < sl_run) sl_run = run_len;
        } else if (cells) {
            sg.g = wide ? hs[i]->bo_geom : hs[i]->bp_geom;
        } else {
            sg.g = hs[i]->bd_geom;
        }
        sg.qs = qs[i], sg.qe = qe[i], sg.counts = counts[i];
        sg.nq = nq[i];
        sg.tile0 = ntp;
        sg.ntiles = div_up(nq[i], tile);
        sg.dimages = hs[i]->bd_images.as<unsigned char>();
        sg.pimages = wide ? hs[i]->bo_images.as<unsigned char>() : hs[i]->bp_images.as<unsigned char>();
        sg.smeta = slices ? hs[i]->sl_meta.as<int4>() : nullptr;
        sg.ix = index_dev(hs[i]);
        sg.e_sorted = hs[i]->e_sorted.as<int32_t>();
        ntp += div_up(sg.ntiles, BM_GROUP_TILES) * BM_GROUP_TILES;
        sg.tile_end = ntp;
        if ((size_t)sg.g.stride > max_stride) max_stride = (size_t)sg.g.stride;
        any_total |= totals_dev && totals_dev[i];
    }
    if (ntp == 0) return BXMI_OK;
    const int ngroups = (int)(ntp / BM_GROUP_TILES);
    // PAIR: a search workgroup holds the images of two neighbouring buckets (needs both in one CU's LDS)
    int chunk = dense ? (g_opt_bd_chunk ? (int)g_opt_bd_chunk : (cells || slices_flat ? 2 : 4) * BM_CHUNK) : g_opt_bm_chunk ? (int)g_opt_bm_chunk : BM_CHUNK;
    // a small batch (one rank's share of a genome on eight GPUs: 13 M queries) cut into items of 128 Ki queries is a hundred
    // workgroups on 256 CUs (measured: search 154 us of a 255 us pass); items of nq / 512, at least a tile
    // (every item stages its unit's keys again: at 25 M queries, 192 items, the smaller items already cost more than
    // the idle CUs did -- 0.33 -> 0.38 ms -- so only batches that leave a third of the chip idle are cut finer)
    if (!(dense ? g_opt_bd_chunk : g_opt_bm_chunk) && !(kind == 3 || cells) && nq_all / chunk < 160) {
        const int64_t c = nq_all / 512;
        chunk = (int)(c < 16384 ? 16384 : c);
    }
    int64_t max_items = (int64_t)n * (BM_NB + 2) + 2 * (nq_all / chunk) + 2;
    if (dense) {  // every segment has at most BM_NB >> f units; empty workgroups of 157 KB of LDS are not free
        // (the padded layout counts up to three more slots per tile and unit as "queries" of the unit)
        int64_t pad_slots = 0;
        for (int i = 0; i < n; i++) pad_slots += 3 * (int64_t)(BM_NB >> segs[(size_t)i].g.f) * (segs[(size_t)i].tile_end - segs[(size_t)i].tile0);
        max_items = 2 * ((nq_all + pad_slots) / chunk) + 2;
        for (int i = 0; i < n; i++) max_items += (BM_NB >> segs[(size_t)i].g.f) + 2;
    }
    // PAD: every unit's run of a tile on whole 16-byte slots (the search's load ring needs one store per pass); the tile
    // sort's scan keeps a unit inside one thread or a few neighbouring lanes
    bool pad = kind == 3 || cells;
    for (int i = 0; i < n && pad; i++) pad = (1 << segs[(size_t)i].g.f) >= (variant == 0 ? 4 : 2);
    const int64_t tile_stride = tile + (pad ? BM_PAD_ROOM : 0);
    BXMI_TRY(h->bm_recs.reserve((size_t)ntp * tile_stride * 4));
    if (pad) BXMI_TRY(h->bd_tend.reserve((size_t)ntp * 4));
    BXMI_TRY(h->bm_slots.reserve((size_t)ntp * tile * 2));
    BXMI_TRY(h->bm_tbl.reserve((size_t)ntp * BM_NB * 2));
    if (!dense) BXMI_TRY(h->bm_runT.reserve((size_t)ntp * BM_NB * 4));
    if (dense) BXMI_TRY(h->bd_unitT.reserve((size_t)ntp * (BM_NB + 1) * 2));  // (+ the row behind the last unit)
    BXMI_TRY(h->bm_grpcnt.reserve((size_t)ngroups * BM_NB * 4));
    BXMI_TRY(h->sl_unitcnt.reserve((size_t)ngroups * BM_NB * 4));
    if (dense && !fxsub) BXMI_TRY(h->bd_cnt16.reserve((size_t)ntp * tile_stride * 2));
    BXMI_TRY(h->bm_items.reserve((size_t)(max_items + 2) * sizeof(int4)));  // [0] = the item count, items from [1]
    if (fx) {  // find(): counts apart from the records, and the tile-sorted offsets
        BXMI_TRY(h->sl_cnt.reserve((size_t)ntp * tile * 4));
        BXMI_TRY(h->sl_loff.reserve((size_t)ntp * tile * 4));
    }
    if (fxsub) {  // ... and what find_exchange.hpp's fill and copy read
        BXMI_TRY(h->fx_tbl2.reserve((size_t)ntp * FX_NBK * 2));
        BXMI_TRY(h->fx_runT2.reserve((size_t)ntp * FX_NBK * 4));
        BXMI_TRY(h->fx_hc.reserve((size_t)ntp * tile * 4));
        BXMI_TRY(h->fx_svq.reserve((size_t)ntp * tile * 4));
        BXMI_TRY(h->fx_parts.reserve((size_t)ntp * (tile / BM_PART_Q) * 8));
        BXMI_TRY(h->fx_tile_tot.reserve((size_t)ntp * 8));
        BXMI_TRY(h->fx_tile_base.reserve((size_t)(ntp + 2) * 8));  // (+ the grand total, + the largest tile total)
        fx->f = segs[0].g.f;
        fx->ntiles = segs[0].ntiles;
    }
    unsigned *search_out = fx ? h->sl_cnt.as<unsigned>() : h->bm_recs.as<unsigned>();
    // parameter block in HBM: [segments][totals pointers][tile -> segment], written by bm_params_kernel from its arguments
    const size_t seg_bytes = (size_t)n * sizeof(BmSeg), tot_bytes = (size_t)n * sizeof(void *);
    const size_t tile_off = (seg_bytes + tot_bytes + 15) & ~(size_t)15, par_bytes = tile_off + (size_t)ntp * sizeof(unsigned short);
    BXMI_TRY(h->bm_params.reserve(par_bytes));
    // [segments][PT_SLOTS partial totals], then the flag: 1 = the starts are NOT sorted
    // ([+0] the order flag, [+4 .. +8) the search's item counters)
    BXMI_TRY(h->p_slots.reserve(((size_t)n * PT_SLOTS + 8) * sizeof(unsigned long long)));
    unsigned long long *slots = h->p_slots.as<unsigned long long>();
    // (several indexes: only the walk on cell images has a sorted-batch form over segments)
    bool multi_sorted = n > 1 && !fx && g_opt_sorted_path && g_opt_sorted_cells != 0 && cells && pad;
    for (int i = 0; i < n && multi_sorted; i++) multi_sorted = nq[i] < ((int64_t)1 << 32) - 8;
    unsigned *unsorted = g_opt_sorted_path && (n == 1 || multi_sorted) && !fx ? reinterpret_cast<unsigned *>(slots + (size_t)n * PT_SLOTS) : nullptr;
    // The order check and the stand-down of the sorted-batch kernel cost a shuffled batch 24 us (of 750).  What the order
    // checks find is mirrored into host memory (ivl_local_count_kernel, bs_walk_kernel or the workgroup that runs the probe write it, nobody waits for
    // it): after two batches in a row that were NOT sorted the check is no longer launched -- every kernel of the exchange
    // runs unconditionally -- and a PROBE rides on the parameter kernel instead: 8192 consecutive starts; a descent among
    // them says "shuffled" for certain, none brings the exact check back with the next call.  A sorted batch that arrives
    // in between goes through the exchange (0.78 instead of 0.62 ms per 100 M), exact as ever.  (Watching the order
    // exactly inside the tile sort, which has every start in registers, cost that kernel 13-19 us -- what the check costs.)
    unsigned *descent = nullptr;
    unsigned long long order_seq = 0;
    if (unsorted) {
        BXMI_TRY(ensure_feedback(h, st));
        const unsigned long long seen = reinterpret_cast<volatile unsigned long long *>(h->bd_fb_host)[1];
        if ((seen >> 1) > h->order_seen) {
            h->unsorted_streak = (seen & 1ull) ? h->unsorted_streak + 1 : 0;
            h->order_seen = seen >> 1;
            h->order_skip = h->unsorted_streak >= 2;
        }
        if (h->order_seq == 0 && g_opt_order_skip != 0) {
            // The handle's first large batch: nothing is known about the caller's order yet, and this call has waited for the
            // device already (it built the index's images) -- so the probe is asked alone and its answer read back: a descent
            // among its 8192 starts drops the exact check from this very pass (a cold pass paid 24 us of 700 for it).
            hipLaunchKernelGGL(bm_probe_kernel, dim3(1), dim3(256), 0, st, qs[0], nq[0], unsorted);
            unsigned seen_descent = 0;
            BXMI_HIP(hipMemcpyAsync(&seen_descent, unsorted, sizeof(unsigned), hipMemcpyDeviceToHost, st));
            BXMI_HIP(hipStreamSynchronize(st));
            if (seen_descent) h->unsorted_streak = 2, h->order_skip = true;
        }
        order_seq = ++h->order_seq;
        if (h->order_skip && g_opt_order_skip != 0) descent = unsorted, unsorted = nullptr;  // (the word is zeroed with the partial totals)
    }
    BmLaunch L;
    memset(&L.par, 0, sizeof(L.par));
    // Folded: whenever the tile sort is the batch's first kernel.  With the order check in front (a handle's first batches, sorted
    // input) or more than BM_PAR_CHUNK segments (a whole genome on one GPU) the parameter kernel stays a launch of its own.
    const bool fold_params = n <= BM_PAR_CHUNK && !unsorted;
    const int n_zero = n * PT_SLOTS + 8;
    if (fold_params) {
        for (int i = 0; i < n; i++) {
            L.par.seg[i] = segs[(size_t)i];
            L.par.total[i] = totals_dev ? reinterpret_cast<unsigned long long *>(totals_dev[i]) : nullptr;
        }
        L.npar = n;
        L.par_out.segs = h->bm_params.as<BmSeg>();
        L.par_out.totals = reinterpret_cast<unsigned long long **>(h->bm_params.as<unsigned char>() + seg_bytes);
        L.par_out.tile_seg = reinterpret_cast<unsigned short *>(h->bm_params.as<unsigned char>() + tile_off);
        L.par_out.zero_u64 = slots, L.par_out.n_zero = n_zero;
        L.par_out.n_items = h->bm_items.as<int>();
        L.par_out.probe = descent;
        L.par_out.order_host = descent ? h->bd_fb_host + 1 : nullptr, L.par_out.order_seq = order_seq;
    }
    for (int first = 0; first < n && !fold_params; first += BM_PAR_CHUNK) {
        BmSegChunk c;
        memset(&c, 0, sizeof(c));
        const int cnt = n - first < BM_PAR_CHUNK ? n - first : BM_PAR_CHUNK;
        for (int i = 0; i < cnt; i++) {
            c.seg[i] = segs[(size_t)(first + i)];
            c.total[i] = totals_dev ? reinterpret_cast<unsigned long long *>(totals_dev[first + i]) : nullptr;
        }
        hipLaunchKernelGGL(bm_params_kernel, dim3((unsigned)cnt), dim3(256), 0, st, c, first, h->bm_params.as<BmSeg>(),
                           reinterpret_cast<unsigned long long **>(h->bm_params.as<unsigned char>() + seg_bytes),
                           reinterpret_cast<unsigned short *>(h->bm_params.as<unsigned char>() + tile_off), slots, n_zero,
                           h->bm_items.as<int>(), first == 0 ? descent : (unsigned *)nullptr, descent ? h->bd_fb_host + 1 : (unsigned long long *)nullptr,
                           order_seq);
    }
    BXMI_LAUNCH_CHECK();
    unsigned long long *tslots = any_total ? slots : nullptr;
    L.segs = h->bm_params.as<BmSeg>();
    L.tile_seg = reinterpret_cast<const unsigned short *>(h->bm_params.as<unsigned char>() + tile_off);
    L.owner = h;
    L.ntp = ntp, L.ngroups = ngroups, L.tile_log2 = tile_log2;
    L.search_lds = slices ? sl_lds : max_stride * 16;
    if (slices_flat && L.search_lds < 4096) L.search_lds = 4096;
    L.gate = unsorted;
    L.descent = descent;
    L.pad = pad;
    L.wide = wide;
    L.xcd_next = reinterpret_cast<unsigned *>(slots + (size_t)n * PT_SLOTS + 4);
    L.n_segs = n;
    // 8-bit counts (0xFF = recomputed by the un-permute kernel, exact either way): half the bytes of the second exchange
    // when the counts are small.  Cell images only serve indexes without piled-up coordinates, so the density says what to
    // expect: fewer than 128 targets per 2048 coordinates (configs[1]: 82; a count of 255 needs a query of ~6000).  What the
    // prediction misses -- long queries, targets crowded into part of the span -- the feedback catches: once more than one
    // count in 64 did not fit, the index keeps 16-bit counts (worst case before that: every count recomputed, ~2 x the pass).
    L.w8 = false;
    // (a batch over several indexes -- a genome -- keeps the feedback with its first index: every index has to be sparse enough,
    // none may have switched the narrow counts off)
    if (pad && cells && g_opt_bd_w8 != 0) {
        BXMI_TRY(ensure_feedback(h, st));
        const unsigned long long wide_counts = *reinterpret_cast<volatile unsigned long long *>(h->bd_fb_host);
        if ((int64_t)wide_counts * 64 > h->w8_queries && wide_counts > 4096) h->w8_off = true;
        bool narrow = true;
        for (int i = 0; i < n; i++) {
            const int64_t span = (int64_t)hs[i]->cmax - (int64_t)hs[i]->geom.cmin + 1;
            narrow = narrow && !hs[i]->w8_off && (int64_t)hs[i]->n * 2048 < span * 128;
        }
        L.w8 = g_opt_bd_w8 > 0 || narrow;
        if (L.w8) h->w8_queries += nq_all;
    }
    if (unsorted) {
        // one index, its batch possibly sorted by start already: one pass over the queries as they lie then, and every
        // kernel below stands down (the local kernel exits at once otherwise)
        // Cell images (bitmap or offset cells): a sorted batch is answered straight from them, stretch by stretch (count_dense.hpp,
        // bs_*): the order check leaves where every unit's queries begin, a plan cuts long stretches, the walk loads a unit's
        // image and answers its queries as they lie.  Other stages keep the first-generation kernel for sorted batches below.
        const bool sorted_on_cells = cells && pad && g_opt_sorted_cells != 0 && nq[0] < ((int64_t)1 << 32) - 8;
        if (multi_sorted) {
            // a batch over several indexes: order check and plan per segment, one walk (count_dense.hpp, bs_*_multi)
            const unsigned chunk = (unsigned)(g_opt_bd_chunk ? g_opt_bd_chunk : (wide ? 1 : 2) * BM_CHUNK);
            size_t max_sorted_items = 4;
            for (int i = 0; i < n; i++) max_sorted_items += (size_t)(BM_NB >> segs[(size_t)i].g.f) + 4 + (size_t)(nq[i] / chunk);
            const size_t bounds_bytes = ((size_t)n * BS_BOUNDS_ROW * 4 + 16 + 15) & ~(size_t)15;
            BXMI_TRY(h->bs_plan.reserve(bounds_bytes + (max_sorted_items + 1) * sizeof(int4)));
            unsigned *bounds_all = h->bs_plan.as<unsigned>();
            int *n_sorted = reinterpret_cast<int *>(bounds_all + (size_t)n * BS_BOUNDS_ROW);
            int4 *sorted_items = reinterpret_cast<int4 *>(h->bs_plan.as<unsigned char>() + bounds_bytes);
            BXMI_HIP(hipMemsetAsync(n_sorted, 0, sizeof(int), st));
            hipLaunchKernelGGL(bs_check_multi_kernel, dim3((unsigned)(ntp < 2048 ? ntp : 2048)), dim3(256), 0, st, L.segs, L.tile_seg, ntp, tile_log2, unsorted, bounds_all);
            hipLaunchKernelGGL(bs_plan_multi_kernel, dim3((unsigned)n), dim3(1024), 0, st, L.segs, bounds_all, chunk, sorted_items, n_sorted, unsorted);
            if (wide) {
                BXMI_TRY(allow_big_lds((bs_walk_kernel<true, BD_THREADS / 2>), L.search_lds));
                hipLaunchKernelGGL((bs_walk_kernel<true, BD_THREADS / 2>), dim3(512), dim3(BD_THREADS / 2), L.search_lds, st, L.segs, sorted_items, n_sorted, tslots,
                                   unsorted, L.xcd_next, h->bd_fb_host + 1, order_seq);
            } else {
                BXMI_TRY(allow_big_lds((bs_walk_kernel<false, BD_THREADS>), L.search_lds));
                hipLaunchKernelGGL((bs_walk_kernel<false, BD_THREADS>), dim3(256), dim3(BD_THREADS), L.search_lds, st, L.segs, sorted_items, n_sorted, tslots,
                                   unsorted, L.xcd_next, h->bd_fb_host + 1, order_seq);
            }
            BXMI_LAUNCH_CHECK();
        } else if (sorted_on_cells) {
            const BmGeom &g0 = segs[0].g;
            const int units = BM_NB >> g0.f;
            const unsigned chunk = (unsigned)(g_opt_bd_chunk ? g_opt_bd_chunk : (wide ? 1 : 2) * BM_CHUNK);
            const size_t max_sorted_items = (size_t)units + 4 + (size_t)(nq[0] / chunk);
            BXMI_TRY(h->bs_plan.reserve((size_t)(units + 2) * 4 + 16 + (max_sorted_items + 1) * sizeof(int4)));
            BmBounds B;
            B.bounds = h->bs_plan.as<unsigned>(), B.cmin = g0.cmin, B.ulog = g0.shift + g0.f, B.units = units;
            int *n_sorted = reinterpret_cast<int *>(B.bounds + units + 2);
            int4 *sorted_items = reinterpret_cast<int4 *>(h->bs_plan.as<unsigned char>() + (((size_t)(units + 2) * 4 + 16 + 15) & ~(size_t)15));
            hipLaunchKernelGGL(bm_sorted_check_kernel<true>, dim3(2048), dim3(256), 0, st, qs[0], nq[0], unsorted, B);
            hipLaunchKernelGGL(bs_plan_kernel, dim3(1), dim3(1024), 0, st, B.bounds, units, (unsigned)nq[0], chunk, sorted_items, n_sorted, unsorted);
            if (wide) {
                BXMI_TRY(allow_big_lds((bs_walk_kernel<true, BD_THREADS / 2>), L.search_lds));
                hipLaunchKernelGGL((bs_walk_kernel<true, BD_THREADS / 2>), dim3(512), dim3(BD_THREADS / 2), L.search_lds, st, L.segs, sorted_items, n_sorted, tslots,
                                   unsorted, L.xcd_next, h->bd_fb_host + 1, order_seq);
            } else {
                BXMI_TRY(allow_big_lds((bs_walk_kernel<false, BD_THREADS>), L.search_lds));
                hipLaunchKernelGGL((bs_walk_kernel<false, BD_THREADS>), dim3(256), dim3(BD_THREADS), L.search_lds, st, L.segs, sorted_items, n_sorted, tslots,
                                   unsorted, L.xcd_next, h->bd_fb_host + 1, order_seq);
            }
            BXMI_LAUNCH_CHECK();
        } else {
        hipLaunchKernelGGL(bm_sorted_check_kernel<false>, dim3(2048), dim3(256), 0, st, qs[0], nq[0], unsorted, BmBounds{nullptr, 0, 0, 0});
        TreeDev S = h->treeS.dev, E = h->treeE.dev;
        S.lds_from = S.nlev, S.lds_ints = 0, E.lds_from = E.nlev, E.lds_ints = 0;  // walk the global levels only
        const int64_t nchunks = div_up(nq[0], LC_CHUNK);
        hipLaunchKernelGGL(ivl_local_count_kernel, dim3((unsigned)nchunks), dim3(LC_THREADS), 0, st, S, E, index_dev(h), h->e_sorted.as<int32_t>(),
                           qs[0], qe[0], nq[0], counts[0], tslots, unsorted, (int32_t *)nullptr, h->bd_fb_host + 1, order_seq);
        BXMI_LAUNCH_CHECK();
        }
    }
    if (variant == 2)
        BXMI_TRY((bm_launch_tiles<1024, 32>(L, st, fxsub)));
    else if (variant == 1)
        BXMI_TRY((bm_launch_tiles<1024, 16>(L, st, fxsub)));
    else
        BXMI_TRY((bm_launch_tiles<512, 32>(L, st)));
    if (fxsub) {  // the half-bucket run table, half-major (nobody needs its group counts: the fill has its own plan)
        hipLaunchKernelGGL(bm_transpose_kernel<FX_NBK>, dim3((unsigned)ngroups, FX_NBK / 64), dim3(256), 0, st, h->fx_tbl2.as<unsigned short>(), L.segs,
                           L.tile_seg, tile_log2, h->fx_runT2.as<unsigned>(), ntp, (unsigned *)nullptr, unsorted);
        BXMI_LAUNCH_CHECK();
    }
    if (dense) {
        hipLaunchKernelGGL(bd_transpose_kernel, dim3((unsigned)ngroups, BM_NB / 64), dim3(256), 0, st, h->bm_tbl.as<unsigned short>(), L.segs, L.tile_seg,
                           tile_log2, h->bd_unitT.as<unsigned short>(), ntp, h->sl_unitcnt.as<unsigned>(), unsorted,
                           pad ? h->bd_tend.as<unsigned>() : (const unsigned *)nullptr);
        if (n <= BD_PLAN_SEGS)
            hipLaunchKernelGGL(bd_plan_kernel, dim3(1), dim3(1024), 0, st, h->sl_unitcnt.as<unsigned>(), n, L.segs, chunk, h->bm_items.as<int4>() + 1,
                               h->bm_items.as<int>(), unsorted);
        else
            hipLaunchKernelGGL(bm_plan_kernel<2>, dim3(BM_PLAN_BLOCKS), dim3(BM_PLAN_THREADS), 0, st, h->sl_unitcnt.as<unsigned>(), ngroups, L.segs, L.tile_seg,
                               chunk, h->bm_items.as<int4>() + 1, h->bm_items.as<int>(), unsorted);
    } else {
    hipLaunchKernelGGL(bm_transpose_kernel<BM_NB>, dim3((unsigned)ngroups, BM_NB / 64), dim3(256), 0, st, h->bm_tbl.as<unsigned short>(), L.segs, L.tile_seg,
                       tile_log2, h->bm_runT.as<unsigned>(), ntp, h->bm_grpcnt.as<unsigned>(), unsorted);
    hipLaunchKernelGGL(sl_unit_sums_kernel, dim3((unsigned)ngroups), dim3(1024), 0, st, h->bm_grpcnt.as<unsigned>(), L.segs, L.tile_seg,
                       h->sl_unitcnt.as<unsigned>(), unsorted);
    if (n <= BD_PLAN_SEGS)  // (the one-workgroup plan: 11 us where bm_plan_kernel<2> takes 37 on configs[4]'s 1526 tiles)
        hipLaunchKernelGGL(bd_plan_kernel, dim3(1), dim3(1024), 0, st, h->sl_unitcnt.as<unsigned>(), n, L.segs, chunk, h->bm_items.as<int4>() + 1,
                           h->bm_items.as<int>(), unsorted);
    else
        hipLaunchKernelGGL(bm_plan_kernel<2>, dim3(BM_PLAN_BLOCKS), dim3(BM_PLAN_THREADS), 0, st, h->sl_unitcnt.as<unsigned>(), ngroups, L.segs, L.tile_seg,
                           chunk, h->bm_items.as<int4>() + 1, h->bm_items.as<int>(), unsorted);
    }
    BXMI_LAUNCH_CHECK();
    const unsigned sgrid = (unsigned)(div_up(max_items, 8) * 8);
    if (slices_flat)
        BXMI_TRY(bd_launch_search(L, sgrid, 2, false, st));
    else if (slices) {
        // long runs (sparse index, big units): the flat walk; else L lanes per run
        int lanes = g_opt_sl_lanes < 0 ? 0 : (g_opt_sl_lanes ? (int)g_opt_sl_lanes : (sl_run >= 96 ? 0 : (sl_run >= 40 ? 64 : 16)));
        if (fx && lanes == 0) lanes = 64;  // (the fill half has no flat walk)
        if (lanes == 0)
            BXMI_TRY(sl_launch_search_flat(L, sgrid, st));
        else if (lanes == 64)
            BXMI_TRY(sl_launch_search<64>(L, sgrid, st, search_out, fxsub ? h->fx_hc.as<unsigned>() : nullptr));
        else
            BXMI_TRY(sl_launch_search<16>(L, sgrid, st, search_out, fxsub ? h->fx_hc.as<unsigned>() : nullptr));
        if (fx) fx->L = L, fx->sgrid = sgrid, fx->lanes = lanes, fx->variant = variant;
    } else
        BXMI_TRY(bd_launch_search(L, sgrid, cells ? 1 : 0, any_blocks, st));
    unsigned *loff = fx ? h->sl_loff.as<unsigned>() : nullptr;
    if (dense && !fxsub) {
        if (variant == 2)
            BXMI_TRY((bd_launch_unpermute<1024, 32>(L, tslots, st)));
        else
            BXMI_TRY((bd_launch_unpermute<1024, 16>(L, tslots, st)));
    } else if (variant == 2)
        BXMI_TRY((bm_launch_unpermute<1024, 32>(L, tslots, st, search_out, loff, fxsub ? (fx->direct ? 3 : 2) : 0)));
    else
        BXMI_TRY((bm_launch_unpermute<1024, 16>(L, tslots, st, search_out, loff, fxsub ? (fx->direct ? 3 : 2) : 0)));
    if (any_total) {
        hipLaunchKernelGGL(bm_fold_totals_kernel, dim3((unsigned)n), dim3(64), 0, st, slots,
                           reinterpret_cast<unsigned long long *const *>(h->bm_params.as<unsigned char>() + seg_bytes));
        BXMI_LAUNCH_CHECK();
    }
    return BXMI_OK;
}

// The ranks at the half-bucket boundaries (find_exchange.hpp), once per sealed index, with a copy on the host: the piece
// lists are cut there.
static int fx_prepare_index(bxmi_ivl *h, hipStream_t st)
{
    h->fx_state = -1;
    if (h->has_reversed || h->n < 1 || h->geom.shift < 1) return BXMI_OK;
    BXMI_TRY(h->fx_meta2.reserve((size_t)(FX_NBK + 1) * sizeof(int2)));
    BXMI_TRY(h->fx_work.reserve(64));
    hipLaunchKernelGGL(fx_meta_kernel, dim3((FX_NBK + 1 + 255) / 256), dim3(256), 0, st, h->s_ord.as<int32_t>(), (int)h->n, h->geom.cmin, h->geom.shift,
                       h->fx_meta2.as<int2>());
    BXMI_LAUNCH_CHECK();
    h->fx_meta2_host.resize((size_t)FX_NBK + 1);
    BXMI_HIP(hipMemcpyAsync(h->fx_meta2_host.data(), h->fx_meta2.p, (size_t)(FX_NBK + 1) * sizeof(int2), hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    h->fx_pieces_f = -1;
    h->fx_state = 1;
    return BXMI_OK;
}

// The piece list for units of 2^f buckets: consecutive half buckets of one unit while the ranks their records can reach --
// [#{start < first coordinate} - FX_BACK, #{start < last coordinate + SL_MARGIN}) -- fit one LDS window.  A single half
// bucket that does not fit (a pile) is a piece of its own: the window holds the top of its range, the rest is read from HBM.
// The upload is stream-ordered and the caller synchronises `st` before the next call can change the host list.
static int fx_ensure_pieces(bxmi_ivl *h, int f, hipStream_t st)
{
    if (h->fx_pieces_f == f) return BXMI_OK;
    const std::vector<int2> &m = h->fx_meta2_host;
    std::vector<FxPiece> &out = h->fx_pieces_host;
    out.clear();
    const int per_unit = 1 << (f + 1);
    for (int u0 = 0; u0 < FX_NBK; u0 += per_unit) {
        const int u1 = u0 + per_unit < FX_NBK ? u0 + per_unit : FX_NBK;
        int sb = u0;
        while (sb < u1) {
            const int lo = m[(size_t)sb].x > FX_BACK ? m[(size_t)sb].x - FX_BACK : 0;
            int e = sb + 1;
            while (e < u1 && m[(size_t)e + 1].y - lo <= FX_CAPW) e++;
            FxPiece pc;
            pc.sb0 = sb, pc.sb1 = e;
            pc.whi = m[(size_t)e].y;
            pc.wlo = pc.whi - lo > FX_CAPW ? pc.whi - FX_CAPW : lo;
            out.push_back(pc);
            sb = e;
        }
    }
    BXMI_TRY(h->fx_pieces.reserve(out.size() * sizeof(FxPiece)));
    BXMI_HIP(hipMemcpyAsync(h->fx_pieces.p, out.data(), out.size() * sizeof(FxPiece), hipMemcpyHostToDevice, st));
    h->fx_pieces_f = f;
    return BXMI_OK;
}

// find() through the exchange, second generation (find_exchange.hpp): the count half on a tile order by half buckets, the
// CSR offsets from one scan over the tiles' totals, the fill on LDS windows, the copy that finishes the offsets.
static int ivl_find_fx(bxmi_ivl *h, const int32_t *qs, const int32_t *qe, int64_t nq, int64_t *offsets, int32_t *hits, int64_t cap,
                       int64_t *total_host, hipStream_t st)
{
    BXMI_TRY(h->q_cnt.reserve((size_t)(nq + 4) * 4));
    BmFindCtx fx;
    fx.sub = true;
    // Straight into the CSR list, a record's ~20 bytes of hits land anywhere in its tile's 0.6 MB of the list: lines that are
    // completed by other pieces much later.  While the whole list stays in the memory-side cache that is free and the copy is
    // saved (configs[4]'s index, 8 M / 16 M queries: 0.76 / 1.27 ms against 0.84 / 1.37); beyond it every partial line costs HBM
    // a read-modify-write (24 M: equal; 50 M: 3.70 ms against 3.04).  The size of the list is only known after the count half,
    // which has to know the layout -- so the handle's previous batch predicts it (hits per query; 5 before the first).
    {
        const double per_q = h->fx_hits_per_q >= 0.0 ? h->fx_hits_per_q : 5.0;
        fx.direct = g_opt_fx_direct < 0 ? per_q * (double)nq * 4.0 <= 400e6 : g_opt_fx_direct != 0;
    }
    int32_t *counts = h->q_cnt.as<int32_t>();
    int64_t *no_total = nullptr;
    BXMI_TRY(bm_count_segments(&h, 1, &qs, &qe, &nq, &counts, &no_total, st, 2, &fx));
    const int64_t ntp = fx.L.ntp;
    // (only the tiles that hold queries: the un-permute kernel leaves the padding up to the plan group alone)
    hipLaunchKernelGGL(fx_tile_scan_kernel, dim3(1), dim3(1024), 0, st, h->fx_tile_tot.as<unsigned long long>(), fx.ntiles, h->fx_tile_base.as<long long>(),
                       reinterpret_cast<long long *>(offsets) + nq, h->fx_tile_base.as<long long>() + fx.ntiles + 1);
    BXMI_LAUNCH_CHECK();
    BXMI_TRY(fx_ensure_pieces(h, fx.f, st));
    BXMI_HIP(hipMemsetAsync(h->fx_work.p, 0, 64, st));
    if (fx.direct) {  // the offsets are final before the capacity is known to the host; the escapes' hits wait for it on the device
        hipLaunchKernelGGL(fx_offsets_kernel, dim3((unsigned)div_up(nq, 1024)), dim3(256), 0, st, fx.L.segs, h->fx_svq.as<unsigned>(),
                           h->fx_tile_base.as<long long>(), fx.ntiles, fx.L.tile_log2, reinterpret_cast<long long *>(offsets), hits, cap);
        BXMI_LAUNCH_CHECK();
    }
    int64_t total = 0;
    BXMI_HIP(hipMemcpyAsync(&total, offsets + nq, 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    if (total_host) *total_host = total;
    h->fx_hits_per_q = (double)total / (double)nq;
    if (total >= ((int64_t)1 << 31)) {
        // The scratch offsets inside a tile are 31-bit prefixes: a tile of 16 Ki / 32 Ki queries with 2^31 hits or more (queries
        // that each cover a pile of targets) does not fit them.  Only a list this long can hold such a tile: one more word to read then.
        long long max_tile = 0;
        BXMI_HIP(hipMemcpyAsync(&max_tile, h->fx_tile_base.as<long long>() + fx.ntiles + 1, 8, hipMemcpyDeviceToHost, st));
        BXMI_HIP(hipStreamSynchronize(st));
        if (max_tile >= ((long long)1 << 31)) return ivl_find_partitioned(h, qs, qe, nq, offsets, hits, cap, total_host, st);
    }
    if (total > cap && fx.direct) return fail(BXMI_ERANGE, "bxmi_ivl_find: %lld hits need a larger buffer than cap=%lld", (long long)total, (long long)cap);
    if (total > cap) {
        // (the contract: BXMI_ERANGE comes with valid offsets.  The copy kernel, which finishes them, does not run: scan the counts.)
        BXMI_TRY((device_scan<int32_t, long long, OpSum, false>(h->q_cnt.as<int32_t>(), reinterpret_cast<long long *>(offsets), nq, 0ll,
                                                               reinterpret_cast<long long *>(offsets) + nq, h->scan_scratch, st)));
        BXMI_HIP(hipStreamSynchronize(st));
        return fail(BXMI_ERANGE, "bxmi_ivl_find: %lld hits need a larger buffer than cap=%lld", (long long)total, (long long)cap);
    }
    if (total == 0) {  // (the copy kernel writes the offsets: nothing to copy, so they are zeroed here)
        if (!fx.direct) BXMI_HIP(hipMemsetAsync(offsets, 0, (size_t)(nq + 1) * 8, st));
        return BXMI_OK;
    }
    if (!fx.direct) BXMI_TRY(h->sl_hits.reserve((size_t)(total + 16) * 4));
    BXMI_TRY(sl_ensure_eid(h, st));
    const int npieces = (int)h->fx_pieces_host.size();
    // enough (piece, tile chunk) pairs to balance 256 persistent workgroups; a chunk is whole groups of 64 tiles (a wave's batch)
    int64_t nchunks = div_up(2048, npieces);
    if (nchunks > div_up(fx.ntiles, 64)) nchunks = div_up(fx.ntiles, 64);
    if (nchunks < 1) nchunks = 1;
    const int64_t tiles_per_chunk = div_up(div_up(fx.ntiles, nchunks), 64) * 64;
    nchunks = div_up(fx.ntiles, tiles_per_chunk);
    BXMI_TRY(allow_big_lds(fx_fill_kernel, FX_LDS_BYTES));
    hipLaunchKernelGGL(fx_fill_kernel, dim3((unsigned)device_props().cus), dim3(FX_THREADS), FX_LDS_BYTES, st, fx.L.segs, h->fx_pieces.as<FxPiece>(), npieces,
                       (int)nchunks, (int)tiles_per_chunk, h->fx_runT2.as<unsigned>(), ntp, h->bm_recs.as<unsigned>(), h->fx_hc.as<unsigned>(),
                       h->sl_cnt.as<unsigned>(), h->sl_loff.as<unsigned>(), h->fx_tile_base.as<long long>(), h->sl_eid.as<int2>() + SL_WALK,
                       h->fx_meta2.as<int2>(), fx.direct ? hits : h->sl_hits.as<int32_t>(), fx.L.tile_log2, h->fx_work.as<unsigned>());
    BXMI_LAUNCH_CHECK();
    if (fx.direct) return BXMI_OK;  // (every record's hits went where the CSR offsets say)
    const unsigned cgrid = (unsigned)(div_up(ntp, 8) * 8 * (((int64_t)1 << fx.L.tile_log2) / BM_PART_Q));
    {  // two consecutive queries per lane (one: 0.91 ms on configs[4], two: 0.78, four: 0.85)
        auto launch = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(cgrid), dim3(BM_PART_Q / 2), 0, st, fx.L.segs, h->fx_svq.as<unsigned>(), h->fx_tile_base.as<long long>(),
                               h->fx_parts.as<unsigned long long>(), h->sl_hits.as<int32_t>(), reinterpret_cast<long long *>(offsets), hits, ntp);
        };
        if (fx.variant == 2) launch(fx_hits_copy2_kernel<32768, 2>); else launch(fx_hits_copy2_kernel<16384, 2>);
    }
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// Which search stage serves a sealed index in the large-batch pass: 0 = neither (older paths), 1 = bucket images,
// 2 = key slices, 3 = dense unit images (dense indexes try them before the bucket images).  Images cost 0.5 B per coordinate of the span and win on dense indexes; sparse ones (fewer than one
// target per 64 coordinates) and spans whose bucket image outgrows the LDS take slices.  Prepared on first use.
static int bm_choose_stage(bxmi_ivl *h, hipStream_t st, int *kind, int64_t nq)
{
    *kind = 0;
    if (h->has_reversed || h->n < 4096) return BXMI_OK;
    int64_t span = (int64_t)h->cmax - (int64_t)h->geom.cmin;
    if (span < 0) span = 0;
    // Sparse indexes: offset-cell images on the persistent walk, when the batch brings enough queries per unit image (a unit's
    // image is 72 KB to load however few queries it serves; key slices stage a few KB)
    // (ivl.flat / ivl.dense / ivl.slice set to force or forbid a stage leave this one out, ivl.sparse = 1 forces it)
    if ((g_opt_sparse > 0 || (g_opt_sparse < 0 && g_opt_flat < 0 && g_opt_dense < 1 && g_opt_slice < 1)) &&
        (g_opt_bo_cell_log2 || bo_cell_log2_for(span + 1, h->n))) {
        if (h->bo_state == 0) BXMI_TRY(bo_prepare_index(h, st));
        if (h->bo_state == 1) {
            const int64_t units = (span >> (h->bo_geom.shift + h->bo_geom.f)) + 1;
            if (g_opt_sparse > 0 || nq >= units * g_opt_bo_min_per_unit) {
                *kind = 5;
                return BXMI_OK;
            }
        }
    }
    const bool slices_first = g_opt_dense != 1 && g_opt_flat != 1 && (g_opt_slice == 1 || (g_opt_slice < 0 && (span / h->n >= 64 || h->geom.shift > BD_MAX_SHIFT)));
    if (slices_first) {
        if (h->sl_state == 0) BXMI_TRY(sl_prepare_index(h, st));
        if (h->sl_state == 1) {
            *kind = 2;
            return BXMI_OK;
        }
    }
    if (g_opt_flat != 0) {
        if (h->bp_state == 0) BXMI_TRY(bp_prepare_index(h, st));
        if (h->bp_state == 1) {
            *kind = 4;
            return BXMI_OK;
        }
    }
    if (g_opt_dense != 0) {
        if (h->bd_state == 0) BXMI_TRY(bd_prepare_index(h, st));
        if (h->bd_state == 1) {
            *kind = 3;
            return BXMI_OK;
        }
    }
    if (!slices_first && g_opt_slice != 0) {
        if (h->sl_state == 0) BXMI_TRY(sl_prepare_index(h, st));
        if (h->sl_state == 1) *kind = 2;
    }
    return BXMI_OK;
}

static int ivl_stream(bxmi_ivl *h)
{
    if (!h->stream) BXMI_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    return BXMI_OK;
}

extern "C" int bxmi_ivl_create(bxmi_ivl_t **out)
{
    if (!out) return fail(BXMI_EINVAL, "bxmi_ivl_create: out is NULL");
    int dev = 0;
    BXMI_HIP(hipGetDevice(&dev));
    bxmi_ivl *h = new (std::nothrow) bxmi_ivl();
    if (!h) return fail(BXMI_ENOMEM, "bxmi_ivl_create: host allocation failed");
    h->device = dev;
    *out = h;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_destroy(bxmi_ivl_t *h)
{
    if (!h) return BXMI_OK;
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->stream_up) (void)hipStreamDestroy(h->stream_up);
    if (h->stream_down) (void)hipStreamDestroy(h->stream_down);
    if (h->one_buf) (void)hipHostFree(h->one_buf);
    if (h->bd_fb_host) (void)hipHostFree(h->bd_fb_host);
    delete h;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_append(bxmi_ivl_t *h, const int32_t *start, const int32_t *end, int64_t n)
{
    if (!h || n < 0 || (n > 0 && (!start || !end))) return fail(BXMI_EINVAL, "bxmi_ivl_append: bad arguments");
    if ((int64_t)h->h_start.size() + h->n_dev + n >= ((int64_t)1 << 31) - 64)
        return fail(BXMI_EINVAL, "bxmi_ivl_append: more than 2^31 intervals");
    try {
        h->h_start.insert(h->h_start.end(), start, start + n);
        h->h_end.insert(h->h_end.end(), end, end + n);
    } catch (...) {
        return fail(BXMI_ENOMEM, "bxmi_ivl_append: host allocation failed");
    }
    if (n) h->sealed = false;
    return BXMI_OK;
}

// Move host-staged intervals to the device arrays (insertion order preserved).
static int ivl_flush_host(bxmi_ivl *h, hipStream_t st)
{
    int64_t add = (int64_t)h->h_start.size();
    if (!add) return BXMI_OK;
    int64_t tot = h->n_dev + add;
    BXMI_TRY(h->d_start.reserve((size_t)tot * 4, true, st));
    BXMI_TRY(h->d_end.reserve((size_t)tot * 4, true, st));
    BXMI_HIP(hipMemcpyAsync(h->d_start.as<int32_t>() + h->n_dev, h->h_start.data(), (size_t)add * 4, hipMemcpyHostToDevice, st));
    BXMI_HIP(hipMemcpyAsync(h->d_end.as<int32_t>() + h->n_dev, h->h_end.data(), (size_t)add * 4, hipMemcpyHostToDevice, st));
    BXMI_HIP(hipStreamSynchronize(st));
    h->n_dev = tot;
    h->h_start.clear();
    h->h_end.clear();
    return BXMI_OK;
}

extern "C" int bxmi_ivl_append_dev(bxmi_ivl_t *h, const int32_t *start, const int32_t *end, int64_t n, void *stream)
{
    if (!h || n < 0 || (n > 0 && (!start || !end))) return fail(BXMI_EINVAL, "bxmi_ivl_append_dev: bad arguments");
    hipStream_t st = as_stream(stream);
    BXMI_TRY(ivl_flush_host(h, st));
    int64_t tot = h->n_dev + n;
    if (tot >= ((int64_t)1 << 31) - 64) return fail(BXMI_EINVAL, "bxmi_ivl_append_dev: more than 2^31 intervals");
    BXMI_TRY(h->d_start.reserve((size_t)tot * 4, true, st));
    BXMI_TRY(h->d_end.reserve((size_t)tot * 4, true, st));
    BXMI_HIP(hipMemcpyAsync(h->d_start.as<int32_t>() + h->n_dev, start, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    BXMI_HIP(hipMemcpyAsync(h->d_end.as<int32_t>() + h->n_dev, end, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    h->n_dev = tot;
    if (n) h->sealed = false;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_seal(bxmi_ivl_t *h, void *stream)
{
    if (!h) return fail(BXMI_EINVAL, "bxmi_ivl_seal: NULL handle");
    hipStream_t st = as_stream(stream);
    BXMI_TRY(ivl_flush_host(h, st));
    const int64_t n = h->n_dev;
    const int64_t n_pad = (n > 0 ? div_up(n, FAN) : 1) * FAN;
    const size_t pad_bytes = (size_t)(n_pad + FAN) * 4;

    BXMI_TRY(h->keys_a.reserve((size_t)(n + 1) * 8));
    BXMI_TRY(h->keys_b.reserve((size_t)(n + 1) * 8));
    BXMI_TRY(h->ekeys_a.reserve((size_t)(n + 1) * 4));
    BXMI_TRY(h->ekeys_b.reserve((size_t)(n + 1) * 4));
    BXMI_TRY(h->s_ord.reserve(pad_bytes));
    BXMI_TRY(h->e_ord.reserve(pad_bytes));
    BXMI_TRY(h->idx.reserve(pad_bytes));
    BXMI_TRY(h->pm.reserve(pad_bytes));
    BXMI_TRY(h->e_sorted.reserve(pad_bytes));
    BXMI_TRY(h->flag.reserve(64));

    unsigned *d_rev = h->flag.as<unsigned>();
    BXMI_HIP(hipMemsetAsync(d_rev, 0, 4, st));
    const int g = stream_grid(n_pad, 256 * 4);
    if (n > 0) {
        hipLaunchKernelGGL(ivl_make_keys_kernel, dim3(g), dim3(256), 0, st, h->d_start.as<int32_t>(), h->d_end.as<int32_t>(), n,
                           h->keys_a.as<unsigned long long>(), h->ekeys_a.as<uint32_t>(), d_rev);
        BXMI_LAUNCH_CHECK();
    }
    unsigned long long *sorted_keys = nullptr;
    uint32_t *sorted_ends = nullptr;
    BXMI_TRY(radix_sort_keys<unsigned long long>(h->keys_a.as<unsigned long long>(), h->keys_b.as<unsigned long long>(), n,
                                                 &sorted_keys, h->sort_scratch, st));
    BXMI_TRY(radix_sort_keys<uint32_t>(h->ekeys_a.as<uint32_t>(), h->ekeys_b.as<uint32_t>(), n, &sorted_ends, h->sort_scratch, st));
    hipLaunchKernelGGL(ivl_unpack_kernel, dim3(g), dim3(256), 0, st, sorted_keys, h->d_end.as<int32_t>(), n, n_pad,
                       h->s_ord.as<int32_t>(), h->e_ord.as<int32_t>(), h->idx.as<int32_t>());
    hipLaunchKernelGGL(ivl_unbias_kernel, dim3(g), dim3(256), 0, st, sorted_ends, n, n_pad, h->e_sorted.as<int32_t>());
    BXMI_LAUNCH_CHECK();
    // prefix max of ends in tree order; padding = INT_MAX so the pm tree's leaves stay sorted
    BXMI_TRY((device_scan<int32_t, int32_t, OpMax, true>(h->e_ord.as<int32_t>(), h->pm.as<int32_t>(), n, INT_MIN, nullptr,
                                                        h->scan_scratch, st)));
    if (n_pad > n) {
        hipLaunchKernelGGL(ivl_pad_kernel, dim3((unsigned)div_up(n_pad - n, 64)), dim3(64), 0, st, h->pm.as<int32_t>(), n, n_pad, INT_MAX);
        BXMI_LAUNCH_CHECK();
    }
    BXMI_TRY(h->treeS.build(h->s_ord.as<int32_t>(), n, st));
    BXMI_TRY(h->treeE.build(h->e_sorted.as<int32_t>(), n, st));
    BXMI_TRY(h->treeP.build(h->pm.as<int32_t>(), n, st));
    unsigned rev = 0;
    int32_t cmin = 0, cmax = 0;
    BXMI_HIP(hipMemcpyAsync(&rev, d_rev, 4, hipMemcpyDeviceToHost, st));
    if (n > 0) {
        BXMI_HIP(hipMemcpyAsync(&cmin, h->s_ord.as<int32_t>(), 4, hipMemcpyDeviceToHost, st));
        BXMI_HIP(hipMemcpyAsync(&cmax, h->e_sorted.as<int32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, st));
    }
    BXMI_HIP(hipStreamSynchronize(st));
    h->has_reversed = rev != 0;
    h->n = n;
    // bucket grid of the partitioned count path: PT_NB buckets of width 2^shift over [min start, max end]
    {
        int64_t span = (int64_t)cmax - (int64_t)cmin;
        if (span < 0) span = 0;
        int shift = 0;
        while ((span >> shift) >= PT_NB) shift++;
        h->geom.cmin = cmin;
        h->geom.shift = shift;
        h->cmax = cmax;
        h->sl_state = 0;
        h->fx_state = 0, h->fx_pieces_f = -1;
        h->bd_state = 0;
        h->bp_state = 0;
        h->bo_state = 0;
        h->w8_off = false, h->w8_queries = 0;  // (the feedback of the 8-bit counts belongs to the index that was)
        if (h->bd_fb_host) {
            BXMI_HIP(hipMemsetAsync(h->bd_fb.p, 0, 64, st));
            *h->bd_fb_host = 0;
        }
        h->unsorted_streak = 0, h->order_skip = false;
        h->sl_eid_ready = false;
        BXMI_TRY(h->slice_bounds.reserve(PT_NB * sizeof(SliceBound)));
        hipLaunchKernelGGL(part_bounds_kernel, dim3(PT_NB / 256), dim3(256), 0, st, h->s_ord.as<int32_t>(), h->e_sorted.as<int32_t>(),
                           h->pm.as<int32_t>(), (int)n, h->geom, h->slice_bounds.as<SliceBound>());
        BXMI_LAUNCH_CHECK();
        h->images_ready = false;  // built by the first large batch that needs them
        BXMI_HIP(hipStreamSynchronize(st));
    }
    h->sealed = true;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_size(const bxmi_ivl_t *h, int64_t *n)
{
    if (!h || !n) return fail(BXMI_EINVAL, "bxmi_ivl_size: bad arguments");
    *n = h->n_dev + (int64_t)h->h_start.size();
    return BXMI_OK;
}

extern "C" int bxmi_ivl_has_reversed(const bxmi_ivl_t *h, int *flag)
{
    if (!h || !flag) return fail(BXMI_EINVAL, "bxmi_ivl_has_reversed: bad arguments");
    if (!h->sealed) return fail(BXMI_ESTATE, "bxmi_ivl_has_reversed: index not sealed");
    *flag = h->has_reversed;
    return BXMI_OK;
}

static int need_sealed(const bxmi_ivl *h, const char *who)
{
    if (!h) return fail(BXMI_EINVAL, "%s: NULL handle", who);
    if (!h->sealed) return fail(BXMI_ESTATE, "%s: index not sealed (call bxmi_ivl_seal after appending)", who);
    return BXMI_OK;
}

extern "C" int bxmi_ivl_flat_state(const bxmi_ivl_t *h, int *state, int64_t *hard_cells)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_flat_state"));
    if (state) *state = h->bp_state;
    if (hard_cells) *hard_cells = h->bp_hard_cells;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_sparse_state(const bxmi_ivl_t *h, int *state, int64_t *hard_cells, int *cell_log2)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_sparse_state"));
    if (state) *state = h->bo_state;
    if (hard_cells) *hard_cells = h->bo_hard_cells;
    if (cell_log2) *cell_log2 = h->bo_state == 1 ? 5 + h->bo_geom.dshift : 0;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_count_width(const bxmi_ivl_t *h, int *bits, int64_t *wide_counts)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_count_width"));
    const unsigned long long wide = h->bd_fb_host ? *reinterpret_cast<volatile unsigned long long *>(h->bd_fb_host) : 0ull;
    const int64_t span = (int64_t)h->cmax - (int64_t)h->geom.cmin + 1;
    if (bits) *bits = !h->w8_off && (int64_t)h->n * 2048 < span * 128 ? 8 : 16;
    if (wide_counts) *wide_counts = (int64_t)wide;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_order_state(const bxmi_ivl_t *h, int *skipping, int64_t *answers_seen)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_order_state"));
    if (skipping) *skipping = h->order_skip && g_opt_order_skip != 0 ? 1 : 0;
    if (answers_seen) *answers_seen = (int64_t)h->order_seen;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_dense_state(const bxmi_ivl_t *h, int *state, int64_t *worst)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_dense_state"));
    if (state) *state = h->bd_state;
    if (worst) worst[0] = h->bd_worst[0], worst[1] = h->bd_worst[1];
    return BXMI_OK;
}

extern "C" int bxmi_ivl_slice_state(const bxmi_ivl_t *h, int *state, int64_t *unit_keys)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_slice_state"));
    if (state) *state = h->sl_state;
    if (unit_keys)
        for (int f = 0; f <= SL_MAX_F; f++) unit_keys[f] = h->sl_need[f];
    return BXMI_OK;
}

extern "C" int bxmi_ivl_order_dev(const bxmi_ivl_t *h, const int32_t **idx_dev, const int32_t **start_dev,
                                  const int32_t **end_dev)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_order_dev"));
    if (idx_dev) *idx_dev = h->idx.as<int32_t>();
    if (start_dev) *start_dev = h->s_ord.as<int32_t>();
    if (end_dev) *end_dev = h->e_ord.as<int32_t>();
    return BXMI_OK;
}

extern "C" int bxmi_ivl_order(const bxmi_ivl_t *h, int32_t *idx_out)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_order"));
    if (h->n && !idx_out) return fail(BXMI_EINVAL, "bxmi_ivl_order: idx_out is NULL");
    if (h->n) BXMI_HIP(hipMemcpy(idx_out, h->idx.p, (size_t)h->n * 4, hipMemcpyDeviceToHost));
    return BXMI_OK;
}

static IndexDev index_dev(const bxmi_ivl *h)
{
    IndexDev ix;
    ix.s_ord = h->s_ord.as<int32_t>();
    ix.e_ord = h->e_ord.as<int32_t>();
    ix.idx = h->idx.as<int32_t>();
    ix.pm = h->pm.as<int32_t>();
    ix.n = (int32_t)h->n;
    ix.has_reversed = h->has_reversed;
    return ix;
}

template <typename Kern>
static int allow_big_lds(Kern k, size_t bytes)
{
    BXMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return BXMI_OK;
}

extern "C" int bxmi_ivl_count_dev(bxmi_ivl_t *h, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t *counts,
                                  int64_t *total_dev, void *stream)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_count_dev"));
    if (nq < 0 || (nq > 0 && (!qs || !qe))) return fail(BXMI_EINVAL, "bxmi_ivl_count_dev: bad arguments");
    if (nq == 0) return BXMI_OK;
    if (((uintptr_t)qs | (uintptr_t)qe | (uintptr_t)counts) & 15)
        return fail(BXMI_EINVAL, "bxmi_ivl_count_dev: query/count arrays must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const bool partition = !h->has_reversed && h->n > 0 &&
                           (g_opt_partition == 1 || (g_opt_partition < 0 && nq >= g_opt_partition_min && h->n >= 4096));
    // the bitmap-cell pass pays off earlier than the bucketed one (its fixed cost is one read of the bucket images)
    // (a caller that wants the total only takes the same pass into a scratch array of counts: 0.65 ms per 100 M where round 1's
    // bucketed pass, which can leave the counts out, takes 0.85)
    const bool bitmap = (counts || total_dev) && g_opt_bitmap != 0 && !h->has_reversed && h->n >= 4096 &&
                        (g_opt_partition == 1 || (g_opt_partition < 0 && nq >= g_opt_bitmap_min));
    if (bitmap) {
        int kind = 0;
        BXMI_TRY(bm_choose_stage(h, st, &kind, nq));
        if (kind) {
            // (counts == NULL: the same pass, its un-permute kernel sums without storing -- no scratch array of counts)
            return bm_count_segments(&h, 1, &qs, &qe, &nq, &counts, &total_dev, st, kind);
        }
    }
    if (partition) return ivl_count_partitioned(h, qs, qe, nq, counts, total_dev, st);
    TreeDev S = h->treeS.dev, E = h->treeE.dev;
    size_t lds_bytes = (size_t)(S.lds_ints + E.lds_ints) * 4 + (CNT_THREADS / 64) * sizeof(long long);
    int grid = device_props().cus;
    int64_t need = div_up(nq, (int64_t)(CNT_THREADS / 8) * CNT_Q);
    if (need < grid) grid = (int)need;
    unsigned long long *tot = reinterpret_cast<unsigned long long *>(total_dev);
    BXMI_TRY(allow_big_lds(ivl_count_kernel<true>, lds_bytes));
    hipLaunchKernelGGL(ivl_count_kernel<true>, dim3(grid), dim3(CNT_THREADS), lds_bytes, st, S, E, index_dev(h), qs, qe, nq, counts, tot);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

extern "C" int bxmi_ivl_count_multi_dev(bxmi_ivl_t *const *hs, int n, const int32_t *const *qs, const int32_t *const *qe, const int64_t *nq,
                                        int32_t *const *counts, int64_t *const *totals_dev, void *stream)
{
    if (n < 0 || (n > 0 && (!hs || !qs || !qe || !nq))) return fail(BXMI_EINVAL, "bxmi_ivl_count_multi_dev: bad arguments");
    hipStream_t st = as_stream(stream);
    // indexes whose batch can ride the bitmap-cell pass are answered together (one pass, six launches); the others one by one
    std::vector<bxmi_ivl *> fh[5];  // by search stage (kind - 1): [1] slices, [2] dense unit images, [3] cell images of units, [4] offset cells
    std::vector<const int32_t *> fqs[5], fqe[5];
    std::vector<int64_t> fnq[5];
    std::vector<int32_t *> fc[5];
    std::vector<int64_t *> ft[5];
    std::vector<int> rest;
    int64_t nq_all = 0;
    for (int i = 0; i < n; i++) {
        BXMI_TRY(need_sealed(hs[i], "bxmi_ivl_count_multi_dev"));
        if (nq[i] < 0 || (nq[i] > 0 && (!qs[i] || !qe[i]))) return fail(BXMI_EINVAL, "bxmi_ivl_count_multi_dev: bad arguments for index %d", i);
        if (((uintptr_t)qs[i] | (uintptr_t)qe[i] | (uintptr_t)(counts ? counts[i] : nullptr)) & 15)
            return fail(BXMI_EINVAL, "bxmi_ivl_count_multi_dev: query/count arrays must be 16-byte aligned");
        nq_all += nq[i];
    }
    const bool fused = g_opt_bitmap != 0 && (counts || totals_dev) && (g_opt_partition == 1 || (g_opt_partition < 0 && nq_all >= g_opt_bitmap_min));
    for (int i = 0; i < n; i++) {
        bxmi_ivl *h = hs[i];
        if (nq[i] == 0) continue;
        int kind = 0;
        // a segment occupies whole groups of 64 tiles of scratch whatever its size: in a batch over hundreds of indexes
        // (a scaffold-level assembly) the ones with a handful of queries are answered one by one instead
        const bool tiny = n > 256 && nq[i] < 65536;
        // (an index whose caller wants neither counts nor a total has nothing to compute; total-only segments ride the same pass)
        const bool wanted = (counts && counts[i]) || (totals_dev && totals_dev[i]);
        if (fused && wanted && !tiny) BXMI_TRY(bm_choose_stage(h, st, &kind, nq[i]));
        if (kind) {
            const int k = kind - 1;
            fh[k].push_back(h), fqs[k].push_back(qs[i]), fqe[k].push_back(qe[i]), fnq[k].push_back(nq[i]), fc[k].push_back(counts ? counts[i] : nullptr);
            ft[k].push_back(totals_dev ? totals_dev[i] : nullptr);
        } else {
            rest.push_back(i);
        }
    }
    for (int k = 0; k < 5; k++)
        if (!fh[k].empty())
            BXMI_TRY(bm_count_segments(fh[k].data(), (int)fh[k].size(), fqs[k].data(), fqe[k].data(), fnq[k].data(), fc[k].data(), ft[k].data(), st,
                                       k + 1));
    for (int i : rest)
        BXMI_TRY(bxmi_ivl_count_dev(hs[i], qs[i], qe[i], nq[i], counts ? counts[i] : nullptr, totals_dev ? totals_dev[i] : nullptr, stream));
    return BXMI_OK;
}

static int upload_queries(bxmi_ivl *h, const int32_t *qs, const int32_t *qe, int64_t nq, hipStream_t st)
{
    BXMI_TRY(h->q_s.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_e.reserve((size_t)(nq + 4) * 4));
    BXMI_HIP(hipMemcpyAsync(h->q_s.p, qs, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    BXMI_HIP(hipMemcpyAsync(h->q_e.p, qe, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    return BXMI_OK;
}

// Host threads that touch the pages of an OUTPUT array chunk by chunk, ahead of the downloads into it.  A copy into fresh pageable
// memory (numpy.empty) pays for its page faults on the copying thread: 400 MB cost the download 10-16 ms, 32 ms per 100 M
// counts against 16 into touched memory.  A page is read and written back; the array is the call's output -- nobody else holds
// it -- and chunk k's download waits (wait_chunk) until its touchers are done with it, so a touch never lands on copied data.
struct PageToucher {
    char *base = nullptr;
    size_t bytes = 0, chunk = 0;
    int nchunks = 0, nthreads = 0;
    std::vector<std::atomic<int>> done;  // [chunk]: threads done with it
    std::atomic<bool> stop{false};
    std::vector<std::thread> threads;
    void start(void *p, size_t n, size_t chunk_bytes, int nthr)
    {
        base = static_cast<char *>(p), bytes = n, chunk = chunk_bytes, nthreads = nthr;
        nchunks = (int)((n + chunk_bytes - 1) / chunk_bytes);
        done = std::vector<std::atomic<int>>((size_t)nchunks);
        for (auto &d : done) d.store(0);
        for (int j = 0; j < nthreads; j++) threads.emplace_back([this, j] { run(j); });
    }
    void run(int j)
    {
        for (int k = 0; k < nchunks && !stop.load(); k++) {
            const size_t o = (size_t)k * chunk, m = std::min(chunk, bytes - o);
            volatile char *b = base + o;
            const size_t lo = m * (size_t)j / (size_t)nthreads, hi = m * (size_t)(j + 1) / (size_t)nthreads;
            for (size_t x = lo; x < hi; x += 4096) b[x] = b[x];
            if (hi > lo) b[hi - 1] = b[hi - 1];
            done[(size_t)k].fetch_add(1, std::memory_order_release);
        }
    }
    void wait_chunk(int k)
    {
        while (nthreads && done[(size_t)k].load(std::memory_order_acquire) < nthreads && !stop.load()) std::this_thread::yield();
    }
    void join()
    {
        for (auto &t : threads) t.join();
        threads.clear();
    }
    ~PageToucher()
    {
        stop.store(true);
        join();
    }
};

// Device memory -> a pageable host array the caller has not touched yet, at the link's rate: chunks of 32 MB, the touchers one
// or more chunks ahead of the copies.  `t` may be running already (started while the device was still computing); NULL = start here.
static int download_touched(void *dst, const void *src_dev, size_t bytes, hipStream_t st, PageToucher *t = nullptr)
{
    constexpr size_t CH = (size_t)32 << 20;
    if (bytes == 0) return BXMI_OK;
    PageToucher own;
    if (!t && bytes >= 2 * CH && g_opt_host_touchers > 0) own.start(dst, bytes, CH, (int)g_opt_host_touchers), t = &own;
    if (!t) {
        BXMI_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, st));
        return BXMI_OK;
    }
    for (int k = 0; k < t->nchunks; k++) {
        const size_t o = (size_t)k * t->chunk, m = std::min(t->chunk, bytes - o);
        t->wait_chunk(k);
        const hipError_t e = hipMemcpyAsync(static_cast<char *>(dst) + o, static_cast<const char *>(src_dev) + o, m, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) {
            t->stop.store(true);
            return fail(BXMI_EHIP, "download: %s", hipGetErrorString(e));
        }
    }
    return BXMI_OK;
}

// The host-pointer count in chunks: while the pass runs on chunk k (the handle's stream), chunk k+1 is on its way up (stream_up,
// this thread) and the counts of chunk k-1 on their way down (stream_down, a second host thread: a copy from or to pageable
// memory holds its caller until the runtime has staged it, and PCIe carries both directions at once only if two threads ask).
// tools/micro/pcie_probe.hip on the round's box: 56 GB/s either way alone, 47 + 47 GB/s together -- a 100 M batch is bounded by its
// 0.8 GB upload (~17 ms); one piece after the other (upload, pass, download) took 42-74 ms.
struct HostChunks {
    bxmi_ivl *h;
    int32_t *counts;
    int64_t nq, chunk;
    int nchunks;
    std::vector<hipEvent_t> done;  // chunk k's pass has finished (recorded on the handle's stream)
    std::mutex mu;
    std::condition_variable cv;
    int launched = 0;   // chunks whose pass has been launched and whose event is recorded
    std::atomic<bool> stop{false};  // the launching thread failed: nothing more will come
    int rc = BXMI_OK;
    std::string err;
    PageToucher touch;  // the output array's pages, chunk by chunk ahead of the downloads
};

static int host_chunks_download_one(HostChunks *c, int k)
{
    bxmi_ivl *h = c->h;
    const int64_t o = (int64_t)k * c->chunk, m = std::min(c->chunk, c->nq - o);
    c->touch.wait_chunk(k);
    BXMI_HIP(hipStreamWaitEvent(h->stream_down, c->done[k], 0));
    BXMI_HIP(hipMemcpyAsync(c->counts + o, h->q_cnt.as<int32_t>() + o, (size_t)m * 4, hipMemcpyDeviceToHost, h->stream_down));
    return BXMI_OK;
}

static void host_chunks_download(HostChunks *c)
{
    if (hipSetDevice(c->h->device) != hipSuccess) {
        c->rc = BXMI_EHIP, c->err = "hipSetDevice in the download thread failed";
        return;
    }
    for (int k = 0; k < c->nchunks; k++) {
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [&] { return c->launched > k || c->stop; });
            if (c->launched <= k) return;
        }
        const int rc = host_chunks_download_one(c, k);
        if (rc != BXMI_OK) {
            c->rc = rc, c->err = last_error();  // (last_error() is per thread: carried over to the caller's)
            return;
        }
    }
    if (hipStreamSynchronize(c->h->stream_down) != hipSuccess) c->rc = BXMI_EHIP, c->err = "hipStreamSynchronize(stream_down) failed";
}

static int ivl_count_host_chunks(bxmi_ivl *h, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t *counts, int64_t *total)
{
    if (!h->stream_up) BXMI_HIP(hipStreamCreateWithFlags(&h->stream_up, hipStreamNonBlocking));
    if (!h->stream_down) BXMI_HIP(hipStreamCreateWithFlags(&h->stream_down, hipStreamNonBlocking));
    HostChunks c;
    c.h = h, c.counts = counts, c.nq = nq, c.chunk = g_opt_host_chunk, c.nchunks = (int)div_up(nq, g_opt_host_chunk);
    BXMI_TRY(h->q_s.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_e.reserve((size_t)(nq + 4) * 4));
    if (counts) BXMI_TRY(h->q_cnt.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_total.reserve(64));
    BXMI_HIP(hipMemsetAsync(h->q_total.p, 0, 8, h->stream));
    c.done.assign((size_t)c.nchunks, nullptr);
    std::vector<hipEvent_t> up((size_t)c.nchunks, nullptr);
    auto drop_events = [&] {
        for (hipEvent_t e : c.done) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : up) if (e) (void)hipEventDestroy(e);
    };
    for (int k = 0; k < c.nchunks; k++)
        if (hipEventCreateWithFlags(&c.done[k], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&up[k], hipEventDisableTiming) != hipSuccess) {
            drop_events();
            return fail(BXMI_EHIP, "bxmi_ivl_count: hipEventCreate failed");
        }
    std::thread down;
    if (counts) {
        if (g_opt_host_touchers > 0) c.touch.start(counts, (size_t)nq * 4, (size_t)c.chunk * 4, (int)g_opt_host_touchers);
        down = std::thread(host_chunks_download, &c);
    }
    auto one = [&](int k) -> int {
        const int64_t o = (int64_t)k * c.chunk, m = std::min(c.chunk, nq - o);
        BXMI_HIP(hipMemcpyAsync(h->q_s.as<int32_t>() + o, qs + o, (size_t)m * 4, hipMemcpyHostToDevice, h->stream_up));
        BXMI_HIP(hipMemcpyAsync(h->q_e.as<int32_t>() + o, qe + o, (size_t)m * 4, hipMemcpyHostToDevice, h->stream_up));
        BXMI_HIP(hipEventRecord(up[k], h->stream_up));
        BXMI_HIP(hipStreamWaitEvent(h->stream, up[k], 0));
        BXMI_TRY(bxmi_ivl_count_dev(h, h->q_s.as<int32_t>() + o, h->q_e.as<int32_t>() + o, m, counts ? h->q_cnt.as<int32_t>() + o : nullptr,
                                    h->q_total.as<int64_t>(), h->stream));  // (the chunks' totals add up in the one word)
        BXMI_HIP(hipEventRecord(c.done[k], h->stream));
        return BXMI_OK;
    };
    int rc = BXMI_OK;
    for (int k = 0; k < c.nchunks && rc == BXMI_OK; k++) {
        rc = one(k);
        std::lock_guard<std::mutex> lk(c.mu);
        if (rc == BXMI_OK) c.launched = k + 1;
        else c.stop = true, c.touch.stop.store(true);
        c.cv.notify_one();
    }
    int64_t t = 0;
    if (rc == BXMI_OK) {
        hipError_t e = hipMemcpyAsync(&t, h->q_total.p, 8, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(BXMI_EHIP, "bxmi_ivl_count: reading the total: %s", hipGetErrorString(e));
    } else
        (void)hipStreamSynchronize(h->stream);  // nothing of this call stays in flight behind its return
    if (down.joinable()) down.join();
    c.touch.join();
    drop_events();
    if (rc == BXMI_OK && c.rc != BXMI_OK) rc = fail(c.rc, "%s", c.err.c_str());
    if (rc == BXMI_OK && total) *total = t;
    return rc;
}

extern "C" int bxmi_ivl_count(bxmi_ivl_t *h, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t *counts, int64_t *total)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_count"));
    if (nq < 0 || (nq > 0 && (!qs || !qe))) return fail(BXMI_EINVAL, "bxmi_ivl_count: bad arguments");
    if (total) *total = 0;
    if (nq == 0) return BXMI_OK;
    BXMI_TRY(ivl_stream(h));
    hipStream_t st = h->stream;
    if (g_opt_host_chunk > 0 && nq >= 2 * g_opt_host_chunk) return ivl_count_host_chunks(h, qs, qe, nq, counts, total);
    BXMI_TRY(upload_queries(h, qs, qe, nq, st));
    BXMI_TRY(h->q_cnt.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_total.reserve(64));
    BXMI_HIP(hipMemsetAsync(h->q_total.p, 0, 8, st));
    BXMI_TRY(bxmi_ivl_count_dev(h, h->q_s.as<int32_t>(), h->q_e.as<int32_t>(), nq, counts ? h->q_cnt.as<int32_t>() : nullptr,
                                h->q_total.as<int64_t>(), st));
    if (counts) BXMI_HIP(hipMemcpyAsync(counts, h->q_cnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    int64_t t = 0;
    BXMI_HIP(hipMemcpyAsync(&t, h->q_total.p, 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    if (total) *total = t;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_find_dev(bxmi_ivl_t *h, const int32_t *qs, const int32_t *qe, int64_t nq, int64_t *offsets,
                                 int32_t *hits, int64_t cap, int64_t *total_host, void *stream)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_find_dev"));
    if (nq < 0 || !offsets || (nq > 0 && (!qs || !qe)) || cap < 0 || (cap > 0 && !hits))
        return fail(BXMI_EINVAL, "bxmi_ivl_find_dev: bad arguments");
    hipStream_t st = as_stream(stream);
    if (nq == 0) {
        BXMI_HIP(hipMemsetAsync(offsets, 0, 8, st));
        if (total_host) *total_host = 0;
        return BXMI_OK;
    }
    // The batch passes read the queries 16 bytes per lane: a query array that is not aligned for that (a slice of a device
    // array; legal per bxmi.h) is answered by the direct kernels below, which read it element by element.
    const bool q_aligned = ((((uintptr_t)qs | (uintptr_t)qe) & 15) == 0);
    if (q_aligned && !h->has_reversed && h->n > 0 &&
        (g_opt_partition == 1 || (g_opt_partition < 0 && nq >= g_opt_partition_min && h->n >= 4096)))
    {
        if (g_opt_sorted_path) {
            // one cheap look at the starts decides the path on the host (find() synchronises for the total anyway)
            BXMI_TRY(h->p_slots.reserve((size_t)PT_MAX_SUB * PT_SLOT_STRIDE * sizeof(unsigned long long)));
            unsigned *flag = reinterpret_cast<unsigned *>(h->p_slots.as<unsigned long long>() + PT_SLOTS);
            unsigned unsorted = 0;
            // (a probe of 8192 starts first: a descent among them says "shuffled" for certain and saves the full read of the
            // starts -- 55 us per 50 M -- which only a batch that passes the probe pays)
            if (nq >= (1 << 20)) {
                hipLaunchKernelGGL(bm_probe_kernel, dim3(1), dim3(256), 0, st, qs, nq, flag);
                BXMI_HIP(hipMemcpyAsync(&unsorted, flag, sizeof(unsigned), hipMemcpyDeviceToHost, st));
                BXMI_HIP(hipStreamSynchronize(st));
            }
            if (!unsorted) {
                BXMI_HIP(hipMemsetAsync(flag, 0, sizeof(unsigned), st));
                hipLaunchKernelGGL(ivl_sorted_check_kernel, dim3(stream_grid(nq, 256)), dim3(256), 0, st, qs, nq, flag);
                BXMI_HIP(hipMemcpyAsync(&unsorted, flag, sizeof(unsigned), hipMemcpyDeviceToHost, st));
                BXMI_HIP(hipStreamSynchronize(st));
            }
            if (!unsorted) return ivl_find_local(h, qs, qe, nq, offsets, hits, cap, total_host, st);
        }
        if (g_opt_find_sliced && g_opt_bitmap != 0 && g_opt_slice != 0 && h->n >= 4096 && nq >= g_opt_bitmap_min &&
            !((uintptr_t)offsets & 15)) {  // (fx_offsets / fx_hits_copy2 store the offsets 16 bytes at a time)
            if (h->sl_state == 0) BXMI_TRY(sl_prepare_index(h, st));
            if (h->sl_state == 1) {
                if (h->fx_state == 0) BXMI_TRY(fx_prepare_index(h, st));
                if (h->fx_state == 1) return ivl_find_fx(h, qs, qe, nq, offsets, hits, cap, total_host, st);
            }
        }
        return ivl_find_partitioned(h, qs, qe, nq, offsets, hits, cap, total_host, st);
    }
    BXMI_TRY(h->q_cnt.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_lo.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_hi.reserve((size_t)(nq + 4) * 4));
    TreeDev S = h->treeS.dev, P = h->treeP.dev;
    size_t lds_bytes = (size_t)(S.lds_ints + P.lds_ints) * 4;
    IndexDev ix = index_dev(h);
    int grid = device_props().cus * 2;
    int64_t need = div_up(nq, (int64_t)(FIND_THREADS / 8) * FIND_Q);
    if (need < grid) grid = (int)need;
    BXMI_TRY(allow_big_lds(ivl_find_count_kernel<true>, lds_bytes));
    hipLaunchKernelGGL(ivl_find_count_kernel<true>, dim3(grid), dim3(FIND_THREADS), lds_bytes, st, S, P, ix, qs, qe, nq,
                       h->q_lo.as<int32_t>(), h->q_hi.as<int32_t>(), h->q_cnt.as<int32_t>());
    BXMI_LAUNCH_CHECK();
    // offsets[0..nq) = exclusive sum of counts, offsets[nq] = total
    BXMI_TRY((device_scan<int32_t, long long, OpSum, false>(h->q_cnt.as<int32_t>(), reinterpret_cast<long long *>(offsets), nq, 0ll,
                                                           reinterpret_cast<long long *>(offsets) + nq, h->scan_scratch, st)));
    int64_t total = 0;
    BXMI_HIP(hipMemcpyAsync(&total, offsets + nq, 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    if (total_host) *total_host = total;
    if (total > cap) return fail(BXMI_ERANGE, "bxmi_ivl_find: %lld hits need a larger buffer than cap=%lld", (long long)total, (long long)cap);
    if (total == 0) return BXMI_OK;
    int fgrid = stream_grid(nq, FIND_THREADS / 8);
    hipLaunchKernelGGL(ivl_find_fill_kernel, dim3(fgrid), dim3(FIND_THREADS), 0, st, ix, qs, nq, h->q_lo.as<int32_t>(),
                       h->q_hi.as<int32_t>(), offsets, hits);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

#ifdef BXMI_DEBUG_PEEK
// (debug builds only, never in the shipped library: a look at the scratch a find() left behind)
extern "C" int bxmi_debug_peek(bxmi_ivl_t *h, const char *name, void *host, size_t bytes)
{
    const DevBuf *b = nullptr;
    const std::string n(name);
    if (n == "recs") b = &h->bm_recs;
    else if (n == "slots") b = &h->bm_slots;
    else if (n == "hc") b = &h->fx_hc;
    else if (n == "cnt") b = &h->sl_cnt;
    else if (n == "loff") b = &h->sl_loff;
    else if (n == "svq") b = &h->fx_svq;
    else if (n == "tile_base") b = &h->fx_tile_base;
    else if (n == "runT2") b = &h->fx_runT2;
    else if (n == "tbl2") b = &h->fx_tbl2;
    else if (n == "tmp_hits") b = &h->sl_hits;
    else if (n == "meta2") b = &h->fx_meta2;
    else if (n == "pieces") b = &h->fx_pieces;
    else if (n == "eid") b = &h->sl_eid;
    else if (n == "qcnt") b = &h->q_cnt;
    if (!b) return fail(BXMI_EINVAL, "peek: no such buffer");
    if (bytes > b->cap) bytes = b->cap;
    BXMI_HIP(hipDeviceSynchronize());
    BXMI_HIP(hipMemcpy(host, b->p, bytes, hipMemcpyDeviceToHost));
    return (int)0;
}
#endif

extern "C" int bxmi_ivl_find(bxmi_ivl_t *h, const int32_t *qs, const int32_t *qe, int64_t nq, int64_t *offsets, int32_t *hits,
                             int64_t cap, int64_t *total)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_find"));
    if (nq < 0 || !offsets || (nq > 0 && (!qs || !qe)) || cap < 0 || (cap > 0 && !hits))
        return fail(BXMI_EINVAL, "bxmi_ivl_find: bad arguments");
    BXMI_TRY(ivl_stream(h));
    hipStream_t st = h->stream;
    if (nq == 0) {
        offsets[0] = 0;
        if (total) *total = 0;
        return BXMI_OK;
    }
    // BLOCKS until offsets / hits are written.  Large batches: host threads touch the pages of `offsets` while the queries go up and
    // the device works, those of `hits` (the part the total says will be written) while the offsets come down (PageToucher).
    constexpr size_t TOUCH_CHUNK = (size_t)32 << 20;
    const size_t off_bytes = (size_t)(nq + 1) * 8;
    PageToucher t_off, t_hits;
    const bool touch = g_opt_host_touchers > 0 && off_bytes >= 2 * TOUCH_CHUNK;
    if (touch) t_off.start(offsets, off_bytes, TOUCH_CHUNK, (int)g_opt_host_touchers);
    BXMI_TRY(upload_queries(h, qs, qe, nq, st));
    BXMI_TRY(h->q_off.reserve((size_t)(nq + 2) * 8));
    BXMI_TRY(h->q_hits.reserve((size_t)(cap + 4) * 4));
    int64_t tot = 0;
    int rc = bxmi_ivl_find_dev(h, h->q_s.as<int32_t>(), h->q_e.as<int32_t>(), nq, h->q_off.as<int64_t>(), h->q_hits.as<int32_t>(), cap,
                               &tot, st);
    if (total) *total = tot;
    if (rc != BXMI_OK && rc != BXMI_ERANGE) return rc;
    const bool want_hits = rc == BXMI_OK && tot > 0;
    const bool touch_hits = want_hits && g_opt_host_touchers > 0 && (size_t)tot * 4 >= 2 * TOUCH_CHUNK;
    if (touch_hits) t_hits.start(hits, (size_t)tot * 4, TOUCH_CHUNK, (int)g_opt_host_touchers);
    BXMI_TRY(download_touched(offsets, h->q_off.p, off_bytes, st, touch ? &t_off : nullptr));
    if (want_hits) BXMI_TRY(download_touched(hits, h->q_hits.p, (size_t)tot * 4, st, touch_hits ? &t_hits : nullptr));
    BXMI_HIP(hipStreamSynchronize(st));
    return rc;
}


constexpr int ONE_CAP = 4096 - 2;  // hits that fit the 16 KiB host-visible result buffer

// IntervalTree.find(start, end) for ONE query (intersection.pyx:400-406): one kernel launch, one stream sync.
extern "C" int bxmi_ivl_find_one(bxmi_ivl_t *h, int32_t qs, int32_t qe, int32_t *hits, int64_t cap, int64_t *n_hits)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_find_one"));
    if (!n_hits || cap < 0 || (cap > 0 && !hits)) return fail(BXMI_EINVAL, "bxmi_ivl_find_one: bad arguments");
    *n_hits = 0;
    if (h->n == 0) return BXMI_OK;
    BXMI_TRY(ivl_stream(h));
    if (!h->one_buf) {
        BXMI_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->one_buf), (4096 + 2) * sizeof(int32_t), hipHostMallocDefault));
        memset(h->one_buf, 0, (4096 + 2) * sizeof(int32_t));
    }
    const unsigned long long seq = ++h->one_seq;
    Tree tS = Tree(), tP = Tree();
    tS.dev = h->treeS.dev, tP.dev = h->treeP.dev;
    tS.set_lds_budget(0), tP.set_lds_budget(0);  // every level from global memory (L2): nothing to stage for one query
    hipLaunchKernelGGL(ivl_find_one_kernel, dim3(1), dim3(ONE_THREADS), 0, h->stream, tS.dev, tP.dev, index_dev(h), qs, qe, h->one_buf, ONE_CAP, seq);
    BXMI_LAUNCH_CHECK();
    BXMI_TRY(wait_for_host_flag(reinterpret_cast<const unsigned long long *>(h->one_buf + 2 + ONE_CAP), seq, h->stream));
    const int64_t n = *reinterpret_cast<volatile long long *>(h->one_buf);
    *n_hits = n;
    if (n > ONE_CAP) {  // a very popular region: take the batched path once
        if (n > cap) return fail(BXMI_ERANGE, "bxmi_ivl_find_one: %lld hits need a larger buffer than cap=%lld", (long long)n, (long long)cap);
        int64_t offs[2], total = 0;
        return bxmi_ivl_find(h, &qs, &qe, 1, offs, hits, cap, &total);
    }
    if (n > cap) return fail(BXMI_ERANGE, "bxmi_ivl_find_one: %lld hits need a larger buffer than cap=%lld", (long long)n, (long long)cap);
    if (n > 0) memcpy(hits, h->one_buf + 2, (size_t)n * sizeof(int32_t));
    return BXMI_OK;
}

// Two lower-bound ranks for the single-position neighbour API:
// out[t] = #{a_t[k] < x_t}, thresholds in 64 bits so position +/- max_dist cannot overflow.
__global__ void ivl_two_ranks_kernel(const int32_t *__restrict__ a0, long long x0, const int32_t *__restrict__ a1,
                                     long long x1, int n, int *out)
{
    if (threadIdx.x < 2) {
        const int32_t *a = threadIdx.x ? a1 : a0;
        long long x = threadIdx.x ? x1 : x0;
        int lo = 0, hi = n;
        while (lo < hi) {
            int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
            if ((long long)a[mid] < x)
                lo = mid + 1;
            else
                hi = mid;
        }
        out[threadIdx.x] = lo;
    }
}

extern "C" int bxmi_ivl_clusters(bxmi_ivl_t *h, const int32_t *ids, int32_t max_dist, int64_t *n_clusters, int32_t *starts,
                                 int32_t *ends, int64_t *offsets, int32_t *members)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_clusters"));
    if (!n_clusters) return fail(BXMI_EINVAL, "bxmi_ivl_clusters: n_clusters is NULL");
    // A negative distance: the reference merges an interval into a cluster when start <= cluster end - d AND end >= cluster
    // start + d (src/cluster.c:224-232, d = -max_dist), and an interval can then be both "right of" and "left of" a cluster: the
    // tree stops being ordered and the regions depend on the insertion order and on rand().  The committed experiment on the
    // reference's own C (tests/golden/cluster_negative_distance.txt, with the script that made it: -1 ... -8, 150 interval
    // sets, 12 insertion orders x 12 seeds) finds ONE distance with an answer: max_dist = -1 on intervals of positive length --
    // "overlap by at least one base", the connected components of the overlap graph, the same sweep as below -- and none
    // for -1 with zero-length intervals or for -2 and beyond.  Those are refused.
    if (max_dist < -1)
        return fail(BXMI_EINVAL, "bxmi_ivl_clusters: max_dist=%d; with a distance below -1 the reference's result depends on the "
                                 "insertion order and is not reproduced", (int)max_dist);
    if (h->has_reversed) return fail(BXMI_ESTATE, "bxmi_ivl_clusters: the index holds intervals with start > end");
    *n_clusters = 0;
    const int64_t n = h->n;
    if (n == 0) return BXMI_OK;
    if (!starts || !ends || !offsets || !members) return fail(BXMI_EINVAL, "bxmi_ivl_clusters: output arrays must hold n (+1) entries");
    BXMI_TRY(ivl_stream(h));
    hipStream_t st = h->stream;
    // scratch: flags / scanned flags in the query buffers, keys in the (free after seal) sort buffers
    BXMI_TRY(h->q_lo.reserve((size_t)(n + 4) * 4));
    BXMI_TRY(h->q_hi.reserve((size_t)(n + 4) * 4));
    BXMI_TRY(h->q_cnt.reserve((size_t)(n + 4) * 4));   // cluster starts
    BXMI_TRY(h->q_hits.reserve((size_t)(n + 4) * 4));  // cluster ends, then members
    BXMI_TRY(h->q_off.reserve((size_t)(n + 4) * 8));   // cluster offsets
    BXMI_TRY(h->q_s.reserve((size_t)(n + 4) * 4));     // ids
    BXMI_TRY(h->q_e.reserve((size_t)(n + 4) * 4));     // members
    BXMI_TRY(h->keys_a.reserve((size_t)(n + 1) * 8));
    BXMI_TRY(h->keys_b.reserve((size_t)(n + 1) * 8));
    int32_t *flag = h->q_lo.as<int32_t>(), *cid = h->q_hi.as<int32_t>();
    const int g = stream_grid(n, 256);
    int32_t *empty_dev = h->q_cnt.as<int32_t>() + n;  // (a spare word of the cluster-start buffer)
    BXMI_HIP(hipMemsetAsync(empty_dev, 0, 4, st));
    hipLaunchKernelGGL(cluster_flag_kernel, dim3(g), dim3(256), 0, st, h->s_ord.as<int32_t>(), h->e_ord.as<int32_t>(), h->pm.as<int32_t>(), (int)n,
                       (int)max_dist, flag, empty_dev);
    BXMI_LAUNCH_CHECK();
    BXMI_TRY((device_scan<int32_t, int32_t, OpSum, true>(flag, cid, n, 0, nullptr, h->scan_scratch, st)));
    int32_t nc = 0, any_empty = 0;
    BXMI_HIP(hipMemcpyAsync(&nc, cid + (n - 1), 4, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipMemcpyAsync(&any_empty, empty_dev, 4, hipMemcpyDeviceToHost, st));
    const int32_t *d_ids = nullptr;
    if (ids) {
        BXMI_HIP(hipMemcpyAsync(h->q_s.p, ids, (size_t)n * 4, hipMemcpyHostToDevice, st));
        d_ids = h->q_s.as<int32_t>();
    }
    hipLaunchKernelGGL(cluster_keys_kernel, dim3(g), dim3(256), 0, st, cid, flag, h->s_ord.as<int32_t>(), h->idx.as<int32_t>(), d_ids, (int)n,
                       h->keys_a.as<unsigned long long>(), h->q_cnt.as<int32_t>(), h->q_off.as<long long>());
    BXMI_LAUNCH_CHECK();
    BXMI_HIP(hipStreamSynchronize(st));  // nc
    if (max_dist < 0 && any_empty)
        return fail(BXMI_EINVAL, "bxmi_ivl_clusters: max_dist=-1 with zero-length intervals; the reference's result depends on the "
                                 "insertion order then and is not reproduced");
    unsigned long long *sorted = nullptr;
    BXMI_TRY(radix_sort_keys<unsigned long long>(h->keys_a.as<unsigned long long>(), h->keys_b.as<unsigned long long>(), n, &sorted,
                                                 h->sort_scratch, st));
    hipLaunchKernelGGL(cluster_finish_kernel, dim3(g), dim3(256), 0, st, sorted, h->pm.as<int32_t>(), h->q_off.as<long long>(), (int)nc, (int)n,
                       h->q_hits.as<int32_t>(), h->q_e.as<int32_t>());
    BXMI_LAUNCH_CHECK();
    BXMI_HIP(hipMemcpyAsync(starts, h->q_cnt.p, (size_t)nc * 4, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipMemcpyAsync(ends, h->q_hits.p, (size_t)nc * 4, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipMemcpyAsync(offsets, h->q_off.p, (size_t)(nc + 1) * 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipMemcpyAsync(members, h->q_e.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    *n_clusters = nc;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_neighbors(bxmi_ivl_t *h, int32_t position, int32_t max_dist, int dir, int32_t *out, int64_t cap,
                                  int64_t *n_out)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_neighbors"));
    if (!n_out || cap < 0 || (cap > 0 && !out) || dir == 0) return fail(BXMI_EINVAL, "bxmi_ivl_neighbors: bad arguments");
    *n_out = 0;
    if (h->n == 0) return BXMI_OK;
    BXMI_TRY(ivl_stream(h));
    hipStream_t st = h->stream;
    BXMI_TRY(h->q_cnt.reserve(64));
    BXMI_TRY(h->q_total.reserve(64));
    BXMI_TRY(h->q_hits.reserve((size_t)(cap + 4) * 4));
    int *d_r = h->q_cnt.as<int>();
    int r[2] = {0, 0};
    const int n = (int)h->n;
    long long vlo, vhi;
    int lo, hi;
    if (dir > 0) {
        // intersection.pyx:213-229,255: p = position + 1, keep 0 <= start - p < max_dist (in-order)
        long long p = (long long)position + 1;
        vlo = p, vhi = p + max_dist;
        hipLaunchKernelGGL(ivl_two_ranks_kernel, dim3(1), dim3(64), 0, st, h->s_ord.as<int32_t>(), vlo, h->s_ord.as<int32_t>(), vhi, n, d_r);
        BXMI_HIP(hipMemcpyAsync(r, d_r, 8, hipMemcpyDeviceToHost, st));
        BXMI_HIP(hipStreamSynchronize(st));
        lo = r[0], hi = r[1];
        hipLaunchKernelGGL(ivl_filter_window_kernel, dim3(1), dim3(256), 0, st, h->s_ord.as<int32_t>(), h->idx.as<int32_t>(), lo, hi, vlo, vhi,
                           0, h->q_hits.as<int32_t>(), cap, h->q_total.as<unsigned long long>());
    } else {
        // intersection.pyx:192-209,240: p = position - 1, keep 0 <= p - end < max_dist (reverse in-order)
        // i.e. p - max_dist < end <= p.  Candidates lie in [first k with pm[k] > p-max_dist, #{start <= p}):
        // the upper bound is the reference's own `minstart > position` prune (:196-197).
        // (With stored intervals whose start exceeds their end -- IntervalTree.insert accepts them -- the reference prunes by
        // SUBTREE (`minstart > position`, :196-197), so whether such an interval with start > position is still reported
        // there depends on the treap's random shape: it is when it shares a subtree with a start <= position.  An index
        // that holds reversed intervals therefore scans the whole window above `lo` and reports every interval whose END
        // qualifies -- everything the reference can report whatever its priorities were; Appendix A.3 of SURVEY.md pins
        // before/after for proper intervals only, for that reason.)
        long long p = (long long)position - 1;
        vlo = p - max_dist + 1, vhi = p + 1;
        hipLaunchKernelGGL(ivl_two_ranks_kernel, dim3(1), dim3(64), 0, st, h->pm.as<int32_t>(), vlo, h->s_ord.as<int32_t>(), vhi, n, d_r);
        BXMI_HIP(hipMemcpyAsync(r, d_r, 8, hipMemcpyDeviceToHost, st));
        BXMI_HIP(hipStreamSynchronize(st));
        lo = r[0], hi = h->has_reversed ? n : r[1];
        if (hi < lo) hi = lo;
        hipLaunchKernelGGL(ivl_filter_window_kernel, dim3(1), dim3(256), 0, st, h->e_ord.as<int32_t>(), h->idx.as<int32_t>(), lo, hi, vlo, vhi,
                           1, h->q_hits.as<int32_t>(), cap, h->q_total.as<unsigned long long>());
    }
    BXMI_LAUNCH_CHECK();
    unsigned long long cnt = 0;
    BXMI_HIP(hipMemcpyAsync(&cnt, h->q_total.p, 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    *n_out = (int64_t)cnt;
    int64_t ncopy = (int64_t)cnt < cap ? (int64_t)cnt : cap;
    if (ncopy > 0) BXMI_HIP(hipMemcpy(out, h->q_hits.p, (size_t)ncopy * 4, hipMemcpyDeviceToHost));
    return BXMI_OK;
}
