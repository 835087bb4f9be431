// count_slices.hpp -- the search stage of the large-batch pass for indexes whose bucket images do not pay or do not
// fit ("sl_*" kernels).  Included by intervals.hip after count_bitmap.hpp: tile sort, run table, plan and un-permute
// are the bm_* kernels unchanged; only what a search workgroup keeps in LDS differs.
//
// The image pass spends 0.5 B of image per COORDINATE of the index's span.  That is nothing for 10M targets on 250M
// coordinates, but a genome's chromosomes carry one target per ~300 coordinates (1.5 GB of images for 100M queries,
// a 147 KB image load in front of every ~4000 queries), and a span of 2e9 (configs[4]) needs buckets whose image is
// larger than the LDS.  Here a workgroup stages the SORTED KEYS themselves: the starts and the ends that fall into
// its unit of 2^f neighbouring buckets, as 16-bit values under a directory --
//     dir[c]  = #{keys of the slice below cell c},   cell = 2^dshift coordinates, at most 2047 cells per slice
//     low[i]  = key i's offset inside its cell (dshift <= 16 bits)
// so a rank is two directory reads and a binary search over ONE cell's keys (a handful): 2 B of LDS per target,
// whatever the span.  Cost per query is about twice the image lookup's; there are no hard cells, no duplicates to
// describe, no images to build or to read, and the unit grows until its keys fill the LDS, which makes the (tile,
// unit) runs longer than the image pass's (tile, bucket pair) runs on sparse indexes.
//
// count(q) = (sLo + #{slice starts < qe}) - (eLo + #{slice ends <= qs})       (intersection.pyx:180-189)
// with sLo / eLo the ranks of the unit's first coordinate, from the per-index boundary table (sl_meta_kernel).
#pragma once

namespace bxmi {

constexpr int SL_MARGIN = 32768;     // the starts' slice reaches this far past the unit: every record's qe is covered
constexpr int SL_CAP = 61440;        // 16-bit keys of both slices together (120 KB of the CU's 160 KB)
constexpr int SL_DIR_CELLS = 2047;   // directory cells per slice, sentinel excluded
constexpr int SL_MAX_F = 6;
constexpr int SL_THREADS = 1024;

// ranks at bucket boundary b (first coordinate lo_b = cmin + b * W), b = 0 .. BM_NB:
//   x = #{start < lo_b}, y = #{end < lo_b}, z = #{start < lo_b + SL_MARGIN}, w = #{end <= lo_b}
__global__ __launch_bounds__(256) void sl_meta_kernel(const int32_t *__restrict__ s_ord, const int32_t *__restrict__ e_sorted, int n, int32_t cmin,
                                                      int shift, int4 *__restrict__ meta)
{
    const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b > BM_NB) return;
    const long long lo = (long long)cmin + ((long long)b << shift);
    meta[b] = make_int4(bm_rank_lt64(s_ord, n, lo), bm_rank_lt64(e_sorted, n, lo), bm_rank_lt64(s_ord, n, lo + SL_MARGIN),
                        bm_rank_lt64(e_sorted, n, lo + 1));
}

// need[f] = the most keys a unit of 2^f buckets stages (both slices), f = 0 .. SL_MAX_F
__global__ __launch_bounds__(1024) void sl_fit_kernel(const int4 *__restrict__ meta, unsigned *__restrict__ need)
{
    for (int f = 0; f <= SL_MAX_F; f++) {
        unsigned m = 0;
        for (int u = threadIdx.x; (u << f) < BM_NB; u += 1024) {
            const int b0 = u << f, b1 = (b0 + (1 << f)) < BM_NB ? b0 + (1 << f) : BM_NB;
            const int4 a = meta[b0], c = meta[b1];
            const unsigned k = (unsigned)(c.z - a.x) + (unsigned)(c.w - a.y);
            m = k > m ? k : m;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned o = __shfl_down(m, off, 64);
            m = o > m ? o : m;
        }
        if (lane_id() == 0) atomicMax(&need[f], m);
    }
}

// grpcnt[group][bucket] -> per unit of 2^f buckets (f is the segment's), in the same [group][BM_NB] layout so that
// bm_plan_kernel<2> plans units exactly as it plans single buckets.
__global__ __launch_bounds__(1024) void sl_unit_sums_kernel(const unsigned *__restrict__ grpcnt, const BmSeg *__restrict__ segs,
                                                            const unsigned short *__restrict__ tile_seg, unsigned *__restrict__ unitcnt,
                                                            const unsigned *__restrict__ gate)
{
    if (gate && *gate == 0) return;
    const int grp = blockIdx.x;
    const int f = segs[tile_seg[(int64_t)grp * BM_GROUP_TILES]].g.f;
    const unsigned *row = grpcnt + (int64_t)grp * BM_NB;
    for (int u = threadIdx.x; u < BM_NB; u += 1024) {
        unsigned s = 0;
        if ((u << f) < BM_NB)
            for (int k = 0; k < (1 << f); k++) s += row[(u << f) + k];
        unitcnt[(int64_t)grp * BM_NB + u] = s;
    }
}

typedef __attribute__((address_space(3))) const unsigned short *lds_u16_p;
typedef int sl_v4a4 __attribute__((ext_vector_type(4), aligned(4)));

// #{keys of the slice below x}: the directory narrows to one cell, `steps` halvings finish (steps covers the fullest
// cell of the unit, so every lane runs the same straight-line code).
__device__ __forceinline__ int sl_rank(lds_u16_p low, lds_u16_p dir, unsigned x, int dshift, int steps)
{
    const unsigned c = x >> dshift, xl = x & ((1u << dshift) - 1u);
    unsigned lo = dir[c], hi = dir[c + 1];
    for (int i = 0; i < steps; i++) {
        const unsigned mid = (lo + hi) >> 1;
        const bool go = lo < hi && (unsigned)low[mid] < xl;
        lo = go ? mid + 1 : lo;
        hi = go ? hi : mid;
    }
    return (int)lo;
}

// What a search workgroup holds for its unit.
struct SlUnit {
    lds_u16_p lowS, lowE, dirS, dirE;
    int sLo, eLo;   // global ranks of the unit's first coordinate
    int nS, nE;
    int steps;
};

// Stage the unit's keys and build the directories.  dyn = [lowS nS_pad][lowE nE_pad][dirS ncs + 2][dirE nce + 2] (u16).
__device__ __forceinline__ SlUnit sl_stage_unit(const BmSeg &sg, int unit, int32_t *dyn, int *s_tmp /* [20] shared */)
{
    const BmGeom g = sg.g;
    const int b0 = unit << g.f, b1 = (b0 + (1 << g.f)) < BM_NB ? b0 + (1 << g.f) : BM_NB;
    const int4 m0 = sg.smeta[b0], m1 = sg.smeta[b1];
    SlUnit U;
    U.sLo = m0.x, U.eLo = m0.y;
    U.nS = m1.z - m0.x, U.nE = m1.w - m0.y;
    unsigned short *lowS = reinterpret_cast<unsigned short *>(dyn);
    unsigned short *lowE = lowS + ((U.nS + 8) & ~7);
    unsigned short *dirS = lowE + ((U.nE + 8) & ~7);
    unsigned short *dirE = dirS + ((g.ncs + 2 + 7) & ~7);
    const long long lo_u = (long long)g.cmin + ((long long)b0 << g.shift);
    const int dshift = g.dshift;
    const unsigned dmask = (1u << dshift) - 1u;
    for (int c = threadIdx.x; c <= g.ncs; c += SL_THREADS) dirS[c] = 0xFFFFu;
    for (int c = threadIdx.x; c <= g.nce; c += SL_THREADS) dirE[c] = 0xFFFFu;
    if (threadIdx.x == 0) s_tmp[16] = 0;
    __syncthreads();
    // four keys per 16-byte load (the slices start at arbitrary ranks: dword-aligned vector loads), every thread's loads
    // of both slices issued before the first is used -- staging is the fixed cost of a work item
    constexpr int SL_STAGE_V = 2;  // vector loads in flight per thread and slice (4 cost the search kernel its 64-register budget)
    const int n_max = U.nS > U.nE ? U.nS : U.nE;
    if (n_max <= 2 * 4 * SL_THREADS) {  // small slices (sparse index): a key per thread and step, nothing issued in vain
        for (int arr = 0; arr < 2; arr++) {
            const int32_t BX_GLOBAL *__restrict__ A = arr == 0 ? as_global(sg.ix.s_ord) + U.sLo : as_global(sg.e_sorted) + U.eLo;  // (as_global: common.hpp)
            const int n = arr == 0 ? U.nS : U.nE;
            unsigned short *low = arr == 0 ? lowS : lowE, *dir = arr == 0 ? dirS : dirE;
            for (int i = threadIdx.x; i < n; i += SL_THREADS) {
                const unsigned rel = (unsigned)((long long)A[i] - lo_u);
                const unsigned prev = i > 0 ? (unsigned)((long long)A[i - 1] - lo_u) >> dshift : 0xFFFFFFFFu;
                low[i] = (unsigned short)(rel & dmask);
                if ((rel >> dshift) != prev) dir[rel >> dshift] = (unsigned short)i;
            }
        }
    } else
    for (int base = 0; base < n_max; base += SL_STAGE_V * 4 * SL_THREADS) {
        sl_v4a4 v[2][SL_STAGE_V];
        int pv[2][SL_STAGE_V];
#pragma unroll
        for (int arr = 0; arr < 2; arr++) {
            const int32_t BX_GLOBAL *__restrict__ A = arr == 0 ? as_global(sg.ix.s_ord) + U.sLo : as_global(sg.e_sorted) + U.eLo;  // (as_global: common.hpp)
            const int n = arr == 0 ? U.nS : U.nE;
#pragma unroll
            for (int k = 0; k < SL_STAGE_V; k++) {
                const int i = base + (k * SL_THREADS + (int)threadIdx.x) * 4;
                // (the index arrays are padded past n: a vector that starts below n may read up to 3 keys beyond it)
                v[arr][k] = *reinterpret_cast<const sl_v4a4 BX_GLOBAL *>(A + (i < n ? i : 0));
                pv[arr][k] = A[i > 0 && i < n ? i - 1 : 0];
            }
        }
#pragma unroll
        for (int arr = 0; arr < 2; arr++) {
            const int n = arr == 0 ? U.nS : U.nE;
            unsigned short *low = arr == 0 ? lowS : lowE, *dir = arr == 0 ? dirS : dirE;
#pragma unroll
            for (int k = 0; k < SL_STAGE_V; k++) {
                const int i = base + (k * SL_THREADS + (int)threadIdx.x) * 4;
                if (i >= n) continue;
                const int key[4] = {v[arr][k].x, v[arr][k].y, v[arr][k].z, v[arr][k].w};
                unsigned prev = i > 0 ? (unsigned)((long long)pv[arr][k] - lo_u) >> dshift : 0xFFFFFFFFu;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (i + j >= n) break;
                    const unsigned rel = (unsigned)((long long)key[j] - lo_u);
                    low[i + j] = (unsigned short)(rel & dmask);
                    if ((rel >> dshift) != prev) dir[rel >> dshift] = (unsigned short)(i + j);  // first key of its cell
                    prev = rel >> dshift;
                }
            }
        }
    }
    __syncthreads();
    // empty cells take the next occupied cell's first key (suffix minimum, the sentinel = n); two cells per thread,
    // thread 0 owns the LAST two, so that an exclusive min-scan over the threads brings "everything to my right".
    unsigned maxocc = 0;
    for (int arr = 0; arr < 2; arr++) {
        unsigned short *dir = arr == 0 ? dirS : dirE;
        const int nc = arr == 0 ? g.ncs : g.nce;
        const unsigned n = (unsigned)(arr == 0 ? U.nS : U.nE);
        const int c0 = 2 * (SL_THREADS - 1 - (int)threadIdx.x), c1 = c0 + 1;
        unsigned a = c1 < nc ? (unsigned)dir[c1] : n, b = c0 < nc ? (unsigned)dir[c0] : n;
        a = a == 0xFFFFu ? n : a, b = b == 0xFFFFu ? n : b;  // (n <= SL_CAP < 0xFFFF)
        unsigned tot;
        const unsigned right = block_exclusive_scan(a < b ? a : b, OpMin(), n, reinterpret_cast<unsigned *>(s_tmp), &tot);
        a = a < right ? a : right;
        b = b < a ? b : a;
        if (c1 <= nc) dir[c1] = (unsigned short)a;
        if (c0 <= nc) dir[c0] = (unsigned short)b;
        if (c0 < nc) {
            const unsigned occ0 = a - b, occ1 = (c1 < nc ? right : a) - a;  // keys of cells c0 and c1
            maxocc = occ0 > maxocc ? occ0 : maxocc;
            maxocc = occ1 > maxocc ? occ1 : maxocc;
        }
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = __shfl_down(maxocc, off, 64);
        maxocc = o > maxocc ? o : maxocc;
    }
    if (lane_id() == 0) atomicMax(&s_tmp[16], (int)maxocc);
    __syncthreads();
    U.steps = 32 - __clz(s_tmp[16]);
    U.lowS = (lds_u16_p)lowS, U.lowE = (lds_u16_p)lowE, U.dirS = (lds_u16_p)dirS, U.dirE = (lds_u16_p)dirE;
    return U;
}

__device__ __forceinline__ unsigned sl_count_record(const SlUnit &U, const BmGeom &g, unsigned rec)
{
    const unsigned len = rec >> g.rshift, off = rec & ((1u << g.rshift) - 1u);
    if (len == bm_len_esc(g)) return BM_REC_ESC;
    const int rE = sl_rank(U.lowE, U.dirE, off + 1u, g.dshift, U.steps);
    const int rS = sl_rank(U.lowS, U.dirS, off + len, g.dshift, U.steps);
    return (unsigned)((U.sLo - U.eLo) + (rS - rE));
}

// find() through the exchange (find_exchange.hpp): the count and, packed for the fill half, `hc` = min(count, 0xFFFF) |
// (rank of qe among the unit's staged starts) << 16 -- the fill walks down from hi = sLo + that rank and never looks a key up.
__device__ __forceinline__ unsigned sl_count_record_hc(const SlUnit &U, const BmGeom &g, unsigned rec, unsigned &hc)
{
    const unsigned len = rec >> g.rshift, off = rec & ((1u << g.rshift) - 1u);
    hc = 0u;
    if (len == bm_len_esc(g)) return BM_REC_ESC;
    const int rE = sl_rank(U.lowE, U.dirE, off + 1u, g.dshift, U.steps);
    const int rS = sl_rank(U.lowS, U.dirS, off + len, g.dshift, U.steps);
    const unsigned c = (unsigned)((U.sLo - U.eLo) + (rS - rE));
    hc = (c < 0xFFFFu ? c : 0xFFFFu) | ((unsigned)rS << 16);
    return c;
}

// N records of a lane at once for find(): their 2 N ranks advance in step (sl_count_slot's shape) -- one after the other they are
// 2 N chains of 2 + steps dependent LDS reads, and with one workgroup per CU (configs[4]: 120 KB of keys) nothing else hides them.
// Same answers as sl_count_record_hc record by record.
#ifndef SL_HC_JOINT
#define SL_HC_JOINT 1  // 0: record by record (round 5's first version; A/B: see DESIGN.md 3.2)
#endif
template <int N>
__device__ __forceinline__ void sl_count_records_hc(const SlUnit &U, const BmGeom &g, const unsigned (&rec)[N], unsigned (&c)[N], unsigned (&hc)[N])
{
    const unsigned omask = (1u << g.rshift) - 1u, dmask = (1u << g.dshift) - 1u, esc_len = bm_len_esc(g);
    constexpr int K = 2 * N;
    unsigned lo[K], hi[K], xl[K];
#pragma unroll
    for (int j = 0; j < N; j++) {
        const unsigned len = rec[j] >> g.rshift, off = rec[j] & omask;
        const bool esc = len == esc_len;
        const unsigned xe = esc ? 0u : off + 1u, xs = esc ? 0u : off + len;  // (an escape record's look-ups stay inside the unit)
        const unsigned ce = xe >> g.dshift, cs = xs >> g.dshift;
        xl[2 * j] = xe & dmask, xl[2 * j + 1] = xs & dmask;
        lo[2 * j] = U.dirE[ce], hi[2 * j] = U.dirE[ce + 1];
        lo[2 * j + 1] = U.dirS[cs], hi[2 * j + 1] = U.dirS[cs + 1];
    }
    for (int i = 0; i < U.steps; i++) {
        unsigned mid[K], key[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            mid[k] = (lo[k] + hi[k]) >> 1;
            key[k] = (k & 1) ? (unsigned)U.lowS[mid[k]] : (unsigned)U.lowE[mid[k]];
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            const bool go = lo[k] < hi[k] && key[k] < xl[k];
            lo[k] = go ? mid[k] + 1 : lo[k];
            hi[k] = go ? hi[k] : mid[k];
        }
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        const bool esc = (rec[j] >> g.rshift) == esc_len;
        const unsigned x = (unsigned)((U.sLo - U.eLo) + ((int)lo[2 * j + 1] - (int)lo[2 * j]));
        c[j] = esc ? BM_REC_ESC : x;
        hc[j] = esc ? 0u : ((x < 0xFFFFu ? x : 0xFFFFu) | (lo[2 * j + 1] << 16));
    }
}

// The four records of a 16-byte slot at once (the flat walk of count_dense.hpp): the eight ranks advance in step -- their
// directory reads, then every halving's eight key reads, are in flight together -- instead of one rank after the other,
// each a chain of 2 + steps dependent LDS reads (round 4: the slice search of a genome share spent its time waiting for
// them: 126 -> see DESIGN.md).  c[j] = the count (BM_REC_ESC for an escape record).
// (Tried in round 4: a rank in TWO LDS reads where the cells are small -- the directory pair as one 4-byte read, the cell's first
// eight keys as one 16-byte read at a 2-byte boundary (gfx950's LDS returns the right bytes: tools/micro/lds_unaligned.hip),
// key < x for all eight by four v_pk_sub_i16 and a popcount.  Exact, and slower: genome pass 1.35 against 0.97 ms, the
// 16-byte reads move eight times the bytes of the halving search through the LDS.  Not kept.)
#ifndef SL_SLOT_RECS
#define SL_SLOT_RECS 2  // records of a slot whose ranks advance together (4: 73 registers in the flat walk -- one workgroup per CU: genome pass 1.18 ms; 2: 0.96; 1: 0.96; one rank after the other: 1.04)
#endif
__device__ __forceinline__ void sl_count_slot(const SlUnit &U, const BmGeom &g, const unsigned (&rec)[4], unsigned (&c)[4], unsigned *hc = nullptr /* [4]: find()'s packed word, see sl_count_record_hc */)
{
    const unsigned omask = (1u << g.rshift) - 1u, dmask = (1u << g.dshift) - 1u, esc_len = bm_len_esc(g);
    constexpr int R = SL_SLOT_RECS, K = 2 * R;
#pragma unroll
    for (int j0 = 0; j0 < 4; j0 += R) {
        unsigned lo[K], hi[K], xl[K];
#pragma unroll
        for (int j = 0; j < R; j++) {
            const unsigned len = rec[j0 + j] >> g.rshift, off = rec[j0 + j] & omask;
            const bool esc = len == esc_len;
            const unsigned xe = esc ? 0u : off + 1u, xs = esc ? 0u : off + len;  // (an escape record's look-ups stay inside the unit)
            const unsigned ce = xe >> g.dshift, cs = xs >> g.dshift;
            xl[2 * j] = xe & dmask, xl[2 * j + 1] = xs & dmask;
            lo[2 * j] = U.dirE[ce], hi[2 * j] = U.dirE[ce + 1];
            lo[2 * j + 1] = U.dirS[cs], hi[2 * j + 1] = U.dirS[cs + 1];
        }
        for (int i = 0; i < U.steps; i++) {
            unsigned mid[K], key[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                mid[k] = (lo[k] + hi[k]) >> 1;
                key[k] = (k & 1) ? (unsigned)U.lowS[mid[k]] : (unsigned)U.lowE[mid[k]];
            }
#pragma unroll
            for (int k = 0; k < K; k++) {
                const bool go = lo[k] < hi[k] && key[k] < xl[k];
                lo[k] = go ? mid[k] + 1 : lo[k];
                hi[k] = go ? hi[k] : mid[k];
            }
        }
#pragma unroll
        for (int j = 0; j < R; j++) {
            const unsigned x = (unsigned)((U.sLo - U.eLo) + ((int)lo[2 * j + 1] - (int)lo[2 * j]));
            const bool esc = (rec[j0 + j] >> g.rshift) == esc_len;
            c[j0 + j] = esc ? BM_REC_ESC : x;
            if (hc) hc[j0 + j] = esc ? 0u : ((x < 0xFFFFu ? x : 0xFFFFu) | (lo[2 * j + 1] << 16));
        }
    }
}

// The walk of bm_search_pipe_kernel (rounds of U runs per L-lane group, the next round's records requested before
// this round is computed, long runs finished by the whole workgroup) over the runs of one UNIT: the run of unit u in
// tile t is what lies between the first slots of buckets u << f and (u + 1) << f in the tile-sorted order.
// APART: the counts go to `out_apart` and the records stay (find() needs them again); otherwise every count is written
// over its record.  Two instantiations rather than two possibly-equal pointers: with may-alias loads and stores the
// compiler drains the memory pipe between a round's stores and the next round's loads (genome search 0.68 -> 0.80 ms).
// (8 waves per SIMD = at most 64 VGPRs: two workgroups per CU when the unit's keys leave room, so that one stages its
// unit while the other searches -- sparse indexes have small units and many work items)
// (Round 6, after the fills: the run table's words kept as loaded and a COUNTED number of stores per round -- six straight-line stores,
// lanes without a record writing to a word nobody reads, so that the waits for the next round's runs and records are vmcnt(6..9)
// instead of drains, HISTORY 11 -- measured 474 -> 489 us on configs[4]: this kernel is bound by its vector instructions, not by
// those waits.  Not kept.)
template <int L, int U, bool APART>
__global__ __launch_bounds__(SL_THREADS) __attribute__((amdgpu_waves_per_eu(APART ? 4 : 8, APART ? 4 : 8))) void sl_search_pipe_kernel(const BmSeg *__restrict__ segs, const int4 *__restrict__ items,
                                                                    const int *__restrict__ n_items, const unsigned *__restrict__ runT, int64_t ntp,
                                                                    unsigned *__restrict__ recs, unsigned *__restrict__ out_apart, int tile_log2,
                                                                    const unsigned *__restrict__ gate, unsigned *__restrict__ hc_out = nullptr /* APART: see sl_count_record_hc */)
{
    unsigned *const out = APART ? out_apart : recs;  // (based on one of the two restrict parameters)
    if (gate && *gate == 0) return;
    constexpr int NG = SL_THREADS / L;
    constexpr unsigned LONG_RUN = 4 * L;
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    __shared__ uint2 s_long[BM_LONG_CAP];
    __shared__ int s_nlong;
    __shared__ int s_tmp[20];
    const int nit = *n_items;
    const int per_xcd = (nit + 7) >> 3;
    const int slot = (int)(blockIdx.x >> 3);
    const int it = (int)(blockIdx.x & 7) * per_xcd + slot;
    if (slot >= per_xcd || it >= nit) return;
    const int4 item = items[it];
    const int unit = item.x & 0xffff, t0 = item.y, t1 = item.z;
    const BmSeg &sg = segs[item.x >> 16];
    const BmGeom g = sg.g;
    const int b0 = unit << g.f, b1 = b0 + (1 << g.f);
    const bool open_end = b1 >= BM_NB;  // the unit reaches the end of the grid: its runs end where the tiles end
    const unsigned *__restrict__ runs0 = runT + (int64_t)b0 * ntp;
    const unsigned *__restrict__ runs1 = runT + (int64_t)(open_end ? b0 : b1) * ntp;
    const int64_t seg_t0 = sg.tile0, seg_nq = sg.nq;
    const int gid = threadIdx.x / L, sub = threadIdx.x % L;
    unsigned run[U];  // first slot | length << 16
    auto load_runs = [&](int tb) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = tb + u * NG + gid;
            const int tc = t < t1 ? t : t0;  // a valid address: no branch around the loads
            const unsigned a = runs0[tc] & 0xffffu;
            unsigned e = runs1[tc] & 0xffffu;
            if (open_end) {
                const int64_t left = seg_nq - (((int64_t)tc - seg_t0) << tile_log2);
                e = left < ((int64_t)1 << tile_log2) ? (unsigned)left : 1u << tile_log2;
            }
            run[u] = t < t1 ? (a | ((e - a) << 16)) : 0u;
        }
    };
    load_runs(t0);
    const SlUnit UN = sl_stage_unit(sg, unit, dyn, s_tmp);
    if (threadIdx.x == 0) s_nlong = 0;
    __syncthreads();
#if defined(SL_EXP) && (SL_EXP & 1)  // diagnostics (wrong results): staging only
    if (UN.steps < 100) return;
#endif
    auto answer = [&](unsigned at, unsigned rec) {
        if (APART && hc_out) {
            unsigned hc;
            out[(size_t)at] = sl_count_record_hc(UN, g, rec, hc);
            hc_out[(size_t)at] = hc;
        } else
            out[(size_t)at] = sl_count_record(UN, g, rec);
    };
    auto prep = [&](BmRound<U> &R, int tb) {
        unsigned cum = 0, lf_at = ~0u, listed_mask = 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = tb + u * NG + gid;
            const unsigned first = ((unsigned)t << tile_log2) + (run[u] & 0xffffu);
            const unsigned len = run[u] >> 16;
            R.first[u] = first;
            R.lens[u] = len;
            R.rec[u] = recs[(size_t)((unsigned)sub < len ? first + (unsigned)sub : 0u)];
            unsigned rem = len > (unsigned)L ? len - (unsigned)L : 0u;
            if (len > LONG_RUN) {  // left to the whole workgroup
                bool listed = false;
                if (sub == 0) {
                    const int k = atomicAdd(&s_nlong, 1);
                    if (k < BM_LONG_CAP) {
                        s_long[k] = make_uint2(first, len);
                        listed = true;
                    }
                }
                listed = __shfl(listed, (int)(threadIdx.x & 63) - sub, 64);
                if (listed) {
                    rem = 0;
                    listed_mask |= 1u << u;
                }
            }
            const unsigned i = (unsigned)sub - cum;
            if ((unsigned)sub >= cum && i < rem) lf_at = first + (unsigned)L + i;
            cum += rem;
        }
        R.lf_total = cum;
        R.listed = listed_mask;
        R.lf_at = lf_at;
        R.lf_second = false;
        R.lf_rec = recs[(size_t)(lf_at != ~0u ? lf_at : 0u)];
    };
    auto finish = [&](BmRound<U> &R) {
        if (SL_HC_JOINT && APART && hc_out) {  // find(): the round's records of a lane together (every lane holds valid records: prep)
            unsigned rr[U + 1], cc[U + 1], hh[U + 1];
#pragma unroll
            for (int u = 0; u < U; u++) rr[u] = R.rec[u];
            rr[U] = R.lf_rec;
#if defined(SL_EXP) && (SL_EXP & 2)  // diagnostics (wrong results): no look-ups
#pragma unroll
            for (int u = 0; u <= U; u++) cc[u] = rr[u] & 0xffu, hh[u] = rr[u] >> 8;
#else
            sl_count_records_hc<U + 1>(UN, g, rr, cc, hh);
#endif
#pragma unroll
            for (int u = 0; u < U; u++)
                if ((unsigned)sub < R.lens[u]) {
                    const size_t at = (size_t)R.first[u] + (unsigned)sub;
                    out[at] = cc[u], hc_out[at] = hh[u];
                }
            if (R.lf_at != ~0u) out[(size_t)R.lf_at] = cc[U], hc_out[(size_t)R.lf_at] = hh[U];
        } else {
#pragma unroll
        for (int u = 0; u < U; u++)
            if ((unsigned)sub < R.lens[u]) answer(R.first[u] + (unsigned)sub, R.rec[u]);
        if (R.lf_at != ~0u) answer(R.lf_at, R.lf_rec);
        }
        for (unsigned base = L; __any(base < R.lf_total); base += L) {  // further leftover passes
            const unsigned i = base + (unsigned)sub;
            unsigned cum = 0, at = ~0u;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const unsigned len = R.lens[u];
                const unsigned rem = len > (unsigned)L && !((R.listed >> u) & 1u) ? len - (unsigned)L : 0u;
                const unsigned j = i - cum;
                if (i >= cum && j < rem) at = R.first[u] + (unsigned)L + j;
                cum += rem;
            }
            if (at != ~0u) answer(at, recs[(size_t)at]);
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
            for (unsigned p = (unsigned)L + threadIdx.x; p < e.y; p += SL_THREADS) answer(e.x + p, recs[(size_t)e.x + p]);
        }
    }
}

// The same search for LONG runs (a unit of many buckets on a sparse index: hundreds of records per tile).  L lanes per
// run waste their passes there; instead the item's runs are laid end to end -- a prefix sum over the run lengths of up
// to SL_FLAT_TILES tiles in LDS -- and the workgroup strides over that flat sequence, every thread keeping DEPTH
// records in flight and advancing its own run pointer as its positions grow.
constexpr int SL_FLAT_TILES = 2048;

template <int DEPTH>
__global__ __launch_bounds__(SL_THREADS) void sl_search_flat_kernel(const BmSeg *__restrict__ segs, const int4 *__restrict__ items,
                                                                    const int *__restrict__ n_items, const unsigned *__restrict__ runT, int64_t ntp,
                                                                    unsigned *__restrict__ recs /* records in, counts out */, int tile_log2,
                                                                    const unsigned *__restrict__ gate)
{
    unsigned *const out = recs;
    if (gate && *gate == 0) return;
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    __shared__ unsigned s_first[SL_FLAT_TILES], s_cum[SL_FLAT_TILES + 1];
    __shared__ int s_tmp[20];
    const int nit = *n_items;
    const int per_xcd = (nit + 7) >> 3;
    const int slot = (int)(blockIdx.x >> 3);
    const int it = (int)(blockIdx.x & 7) * per_xcd + slot;
    if (slot >= per_xcd || it >= nit) return;
    const int4 item = items[it];
    const int unit = item.x & 0xffff, t0 = item.y, t1 = item.z;
    const BmSeg &sg = segs[item.x >> 16];
    const BmGeom g = sg.g;
    const int b0 = unit << g.f, b1 = b0 + (1 << g.f);
    const bool open_end = b1 >= BM_NB;
    const unsigned *__restrict__ runs0 = runT + (int64_t)b0 * ntp;
    const unsigned *__restrict__ runs1 = runT + (int64_t)(open_end ? b0 : b1) * ntp;
    const int64_t seg_t0 = sg.tile0, seg_nq = sg.nq;
    const SlUnit UN = sl_stage_unit(sg, unit, dyn, s_tmp);
    for (int tbase = t0; tbase < t1; tbase += SL_FLAT_TILES) {
        const int nt = t1 - tbase < SL_FLAT_TILES ? t1 - tbase : SL_FLAT_TILES;
        unsigned len[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int j = 2 * (int)threadIdx.x + k;
            const int t = tbase + (j < nt ? j : 0);
            const unsigned a = runs0[t] & 0xffffu;
            unsigned e = runs1[t] & 0xffffu;
            if (open_end) {
                const int64_t left = seg_nq - (((int64_t)t - seg_t0) << tile_log2);
                e = left < ((int64_t)1 << tile_log2) ? (unsigned)left : 1u << tile_log2;
            }
            len[k] = j < nt ? e - a : 0u;
            s_first[j] = a;
        }
        unsigned total;
        const unsigned exc = block_exclusive_scan(len[0] + len[1], OpSum(), 0u, reinterpret_cast<unsigned *>(s_tmp), &total);
        s_cum[2 * threadIdx.x] = exc;
        s_cum[2 * threadIdx.x + 1] = exc + len[0];
        if (threadIdx.x == SL_THREADS - 1) s_cum[SL_FLAT_TILES] = total;
        __syncthreads();
        int r = 0;
        for (unsigned i0 = threadIdx.x; i0 < total; i0 += DEPTH * SL_THREADS) {
            unsigned at[DEPTH], rec[DEPTH];
#pragma unroll
            for (int k = 0; k < DEPTH; k++) {
                const unsigned i = i0 + (unsigned)k * SL_THREADS;
                const bool live = i < total;
                if (live)
                    while (i >= s_cum[r + 1]) r++;  // (runs of length 0 are stepped over)
                at[k] = live ? ((unsigned)(tbase + r) << tile_log2) + s_first[r] + (i - s_cum[r]) : ~0u;
                rec[k] = recs[(size_t)(live ? at[k] : 0u)];
            }
#pragma unroll
            for (int k = 0; k < DEPTH; k++)
                if (at[k] != ~0u) out[(size_t)at[k]] = sl_count_record(UN, g, rec[k]);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// find(): what the fill half of find_exchange.hpp shares with the sorted-batch fill
// ---------------------------------------------------------------------------
// (Round 2's fill and copy on this file's lane groups -- sl_fill_pipe_kernel, sl_hits_unpermute_kernel / sl_hits_copy_kernel:
// configs[4] 9.1 -> 4.7 ms against the bucketed find -- were replaced by find_exchange.hpp in round 5 (2.9 ms) and removed in round 6.)
constexpr int SL_WALK = 8;

// (end, insertion index) of every target in start order, interleaved: one candidate = one 8-byte read, a step of the walk
// = 64 contiguous bytes.  SL_WALK dummy pairs (end = INT_MIN: never a hit) lie in front, so a walk may step past index 0.
typedef int sl_v4a8 __attribute__((ext_vector_type(4), aligned(8)));

__global__ void sl_pack_eid_kernel(const int32_t *__restrict__ e_ord, const int32_t *__restrict__ idx, int n, int2 *__restrict__ eid)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - SL_WALK;
    if (i >= n) return;
    eid[i + SL_WALK] = i < 0 ? make_int2(INT_MIN, 0) : make_int2(e_ord[i], idx[i]);
}

}  // namespace bxmi
