// find_sorted.hpp -- find() on bucketed and on sorted batches: window + count per query, offsets carried to bucket order, the fills (part_fill_kernel, part_fill_pipe_kernel), slice bounds.  intersection.pyx:180-189 (hit order), :400-406.
// Included by intervals.hip (one translation unit; the kernels share its constants and device helpers).
#pragma once

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
// (qs 16-byte aligned: the callers take this path for aligned query arrays only.  Four groups of four starts per thread and turn,
// their loads requested together: one start per thread and turn, two 4-byte loads each, took 57 us per 50 M where this takes ~40.)
__global__ void ivl_sorted_check_kernel(const int32_t *__restrict__ qs, int64_t nq, unsigned *__restrict__ unsorted)
{
    bool descent = false;
    const int64_t n4 = nq >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    const int4 *__restrict__ q4 = reinterpret_cast<const int4 *>(qs);
    constexpr int U = 4;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += stride * U) {
        int4 v[U];
        int nx[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t gi = g + u * stride;
            ok[u] = gi < n4;
            const int64_t ga = ok[u] ? gi : g;
            v[u] = q4[ga];
            nx[u] = 4 * ga + 4 < nq ? qs[4 * ga + 4] : INT_MAX;  // the first start of the next group (none behind the last start)
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            descent |= ok[u] && (v[u].x > v[u].y || v[u].y > v[u].z || v[u].z > v[u].w || v[u].w > nx[u]);
    }
    if (blockIdx.x == 0 && threadIdx.x < 3) {  // the starts behind the last whole group (its last start against them is checked above)
        const int64_t i = 4 * n4 + threadIdx.x;
        if (i + 1 < nq) descent |= qs[i] > qs[i + 1];
    }
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
// registers all at once, one batch EARLY, while the batch before is being walked.
struct FfStage {
    int2 pv[FF_PAIRS / 64];
    int wbase, kmax;
};

// part_fill_pipe_kernel (round 6): the walk above as a software pipeline.  Round 5's kernel (part_fill_flat_kernel, 0.72 ms on
// configs[4]) was a chain per batch of 64 queries: the pairs requested when the batch starts and waited for at once, and -- memory
// operations retire in order, stores included -- the wait for the next batch's numbers at the top of the loop also waited for this
// batch's hit stores to be acknowledged by the L2; worse, where an LDS access and a global one met in one value (a pair inside or
// below the staged window, a hit into the image or straight into the list) the compiler made FLAT instructions of both and
// drained every outstanding memory operation at every step of the walk.  Here
//   * a wave reads ONE CSR offset per batch (its first query's, a broadcast) and scans its own counts (-8 B per query: 0.72 -> 0.66 ms);
//   * every batch issues EXACTLY FF_HITS / 64 store instructions, straight line (lanes beyond the stretch store to a word
//     nobody reads), so the compiler can count them: the wait at the top of a batch is `vmcnt(10)` -- all but the last batch's
//     stores and two younger loads -- and what it lands was requested a whole batch earlier: the numbers of batch i + 1 and, older,
//     the pairs of batch i.  Then the pairs of batch i + 1 and the numbers of batch i + 2 are requested and batch i is walked in LDS;
//   * the loop is unrolled by two with the register sets swapped: no register is copied while a load is on its way to it;
//   * the rare global accesses of the walk are hand-issued, so the walk is LDS-only as far as the compiler's counting goes.
// 0.72 -> 0.57-0.59 ms; sorted find 1.38 -> 1.21-1.24 ms with the chain launched without waiting for the host (ivl_find_local).
__device__ __forceinline__ int wave_max_nonneg_dpp(int x)
{
    auto mx = [](int a, int b) { return a > b ? a : b; };
    x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));  // row_shr:1
    x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));  // row_shr:2
    x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));  // row_shr:4
    x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));  // row_shr:8
    x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3
    x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(x, 63);
}

// Hand-issued global accesses of the pipelined fill's rare paths (see the walk in part_fill_pipe_kernel).
typedef int ffp_v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int2 ffp_global_pair(const int2 *p)
{
    ffp_v2i v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_int2(v.x, v.y);
}
__device__ __forceinline__ long long ffp_global_offset(const long long *p)
{
    long long v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void ffp_global_store(int32_t *p, int v)
{
    asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ void ffp_load(const int2 *__restrict__ eid /* at index 0 */, int c, int hi, FfStage &S)
{
    const int lane = lane_id();
    S.kmax = wave_max_nonneg_dpp(c ? hi : 0);
    S.wbase = S.kmax > FF_PAIRS ? S.kmax - FF_PAIRS : 0;
    const int last = S.kmax > 0 ? S.kmax - 1 : 0;
#pragma unroll
    for (int j = 0; j < FF_PAIRS / 64; j++) {
        const int kk = S.wbase + 64 * j + lane;
        S.pv[j] = eid[kk < S.kmax ? kk : last];  // (a valid address: no branch around the loads)
    }
}

// (70 registers = 7 waves per SIMD; held to 64 / 8 waves it is no faster: 605 against 593 us on configs[4])
__global__ __launch_bounds__(FIND_THREADS) void part_fill_pipe_kernel(const int2 *__restrict__ eid /* at index 0 */, const int32_t *__restrict__ qs_arr,
                                                                     int64_t nq, const int32_t *__restrict__ his, const int32_t *__restrict__ cnt,
                                                                     const long long *__restrict__ offs, int32_t *__restrict__ hits, long long cap,
                                                                     int32_t *__restrict__ nobody /* a word nobody reads */, const unsigned *__restrict__ gate)
{
    if (gate && *gate != 0) return;  // (the batch turned out not to be sorted: the numbers behind it were never written)
    if (offs[nq] > cap) return;  // (launched before the host knows the total: a list that does not fit is not written)
    __shared__ int2 s_pairs[FIND_THREADS / 64][FF_PAIRS];
    __shared__ int32_t s_hits[FIND_THREADS / 64][FF_HITS];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int64_t per_xcd = ((int64_t)gridDim.x + 7) >> 3;
    const int64_t wg = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int64_t per_wg = (nq + gridDim.x - 1) / gridDim.x;
    const int64_t q0 = wg * per_wg, q1 = q0 + per_wg < nq ? q0 + per_wg : nq;
    if (q0 + 64 * wave >= q1) return;
    int2 *const wp = s_pairs[wave];
    int32_t *const wh = s_hits[wave];
    struct Q {
        int c, hi, qs;
        long long off;  // the CSR offset of the batch's FIRST query (one address for the wave: a broadcast)
    };
    auto load_q = [&](int64_t qb, Q &x) {  // (independent loads: one round trip; past the stretch: valid addresses, the counts masked when they are used)
        const int64_t q = qb + lane;
        const int64_t qa = q < q1 ? q : q1 - 1;
        x.c = cnt[qa], x.hi = his[qa], x.qs = qs_arr[qa], x.off = offs[qb < q1 ? qb : q1 - 1];
    };
    auto stores = [&](const long long base_off, const int total) {  // FF_HITS / 64 store instructions whatever `total` is
#pragma unroll
        for (int j = 0; j < FF_HITS / 64; j++) {
            const int i = 64 * j + lane;
            int32_t *const d = i < total ? hits + base_off + i : nobody;
            *d = wh[i];
        }
    };
    // One batch: `cur` = its numbers (landed a batch ago), Scur = its pairs (requested a batch ago), `nxt` = the next batch's
    // numbers (requested a batch ago, landed here), Snxt = where the next batch's pairs go; the numbers of the batch after that
    // are requested into `cur`, whose values have moved to working registers by then.  Called with the two sets swapped every
    // other batch (the loop is unrolled by two): no register is copied while a load is on its way to it.
    auto step = [&](Q &cur, Q &nxt, FfStage &Scur, FfStage &Snxt, const int64_t qb) {
        int c = qb + lane < q1 ? cur.c : 0;
        const int hi = cur.hi, qs = cur.qs;
        const long long base_off = cur.off;
        ffp_load(eid, qb + FIND_THREADS + lane < q1 ? nxt.c : 0, nxt.hi, Snxt);
        load_q(qb + 2 * FIND_THREADS, cur);
        // (counts clamped to "does not fit": 64 of them cannot overflow 32 bits, and a sum that fits is exact)
        const unsigned nc = (unsigned)c <= (unsigned)FF_HITS ? (unsigned)c : (unsigned)FF_HITS + 1u;
        const unsigned incl = wave_inclusive_sum_dpp(nc);
        const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
        const bool flat = tot <= (unsigned)FF_HITS;
        const int rel = (int)(incl - nc);
        if (tot != 0u) {
            int k = hi - 1;
            const int kmax = Scur.kmax, wbase = Scur.wbase;
#pragma unroll
            for (int j = 0; j < FF_PAIRS / 64; j++) {
                const int kk = wbase + 64 * j + lane;
                if (kk < kmax) wp[64 * j + lane] = Scur.pv[j];
            }
            // (lanes read what OTHER lanes staged: wave-level release + barrier, not just in-order DS issue)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // The walk, as the compiler sees it, touches LDS only.  The rare accesses to HBM -- a pair below the staged window, the
            // hits of a stretch that does not fit the image -- are hand-issued (ffp_global_pair waits for its own load, the stores
            // are not waited for): where an LDS access and a global one meet in one value the compiler makes FLAT instructions of
            // both and drains every outstanding memory operation around them, at every step of every batch.
            const int64_t q = qb + lane;
            int32_t *dst = hits;
            if (!flat) dst = hits + ffp_global_offset(offs + (q < q1 ? q : q1 - 1));  // (a wave-uniform branch)
            auto pair_at = [&](int kk) -> int2 {
                int2 p;
                if (kk >= wbase)
                    p = wp[kk - wbase];
                else
                    p = ffp_global_pair(eid + kk);
                return p;
            };
            auto put = [&](int at, int v) {  // hit number `at` of this lane's query
                if (flat)
                    wh[rel + at] = v;
                else
                    ffp_global_store(dst + at, v);
            };
            if (__all(c <= 0 || k - (LANE_WINDOW - 1) >= wbase)) {
                // (four candidates per round trip to the LDS instead of one read and one wait per candidate; what a lane reads beyond
                // its last hit is inside the staged window by the condition above)
                static_assert(LANE_WINDOW % 4 == 0, "the walk reads four candidates at a time");
                const int2 *wq = wp + (c > 0 ? k - wbase : LANE_WINDOW);
                int used = 0;
                for (int st = 0; st < LANE_WINDOW && __any(c > 0); st += 4) {
                    const int2 pa = wq[-st], pb = wq[-st - 1], pc = wq[-st - 2], pd = wq[-st - 3];
                    if (c > 0) { if (pa.x > qs) put(--c, pa.y); used++; }
                    if (c > 0) { if (pb.x > qs) put(--c, pb.y); used++; }
                    if (c > 0) { if (pc.x > qs) put(--c, pc.y); used++; }
                    if (c > 0) { if (pd.x > qs) put(--c, pd.y); used++; }
                }
                k -= used;
            } else
                for (int st = 0; st < LANE_WINDOW && c > 0 && k >= 0; st++, k--) {
                int2 p;
                if (__all(k >= wbase))
                    p = wp[k - wbase];
                else
                    p = pair_at(k);
                if (p.x > qs) put(--c, p.y);
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
                            ffp_global_store(D + (C - 1 - before), p.y);
                    }
                    C -= __popcll(fm);
                    K -= 64;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        stores(base_off, flat ? (int)tot : 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the next batch overwrites both images)
        __builtin_amdgcn_wave_barrier();
    };
    const int64_t first = q0 + 64 * wave;
    Q X, Y;
    FfStage SX, SY;
    load_q(first, X);
    load_q(first + FIND_THREADS, Y);
    ffp_load(eid, first + lane < q1 ? X.c : 0, X.hi, SX);
    {
        int zero;  // (not to the compiler: as many store instructions behind Y's loads on the way into the loop as around it)
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
        stores(0, zero);
    }
    for (int64_t qb = first; qb < q1; qb += 2 * FIND_THREADS) {  // (waves are on their own: no workgroup barrier in here)
        step(X, Y, SX, SY, qb);
        if (qb + FIND_THREADS >= q1) break;
        step(Y, X, SY, SX, qb + FIND_THREADS);
    }
}

// CSR offsets of a sorted find(): offsets[q] = chunk_base[chunk of q] + the exclusive prefix of the chunk's counts -- one read of
// the counts, one write of the offsets (the three-kernel scan read the counts twice and took 0.25 ms per 50 M).
// (Tried on top, round 5: the fill making the offsets itself -- a workgroup per run of chunks, a block scan per batch of 512 queries,
// no offsets kernel and no 8 bytes per query read back: 1.75 ms against 1.53 with this kernel + part_fill_flat_kernel, whose waves
// run free of barriers; forced to 8 waves per SIMD it spilled and took 1.86.  Not kept.)
// (Non-temporal stores of the offsets, round 6: 144 -> 345 us.  Plain stores.)
__global__ __launch_bounds__(LC_THREADS) void lf_offsets_kernel(const int32_t *__restrict__ cnt, const long long *__restrict__ chunk_base, int64_t nq,
                                                                long long *__restrict__ offsets, const unsigned *__restrict__ gate = nullptr)
{
    if (gate && *gate != 0) return;
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

}  // namespace bxmi
