// find_exchange.hpp -- the fill half of find() through the exchange, second generation ("fx_*" kernels).  Included by
// intervals.hip after count_slices.hpp, whose count half (tile sort, run table, plan, slice search, un-permute) it keeps.
//
// IntervalTree.find for a large unsorted batch (intersection.pyx:400-406 -> :180-189; scripts/interval_join.py:16-30), hits
// as CSR in query order.  Round 2's fill half (count_slices.hpp: sl_fill_pipe_kernel, sl_hits_copy_kernel) is bound by the
// number of requests the CUs send to the L2, not by bytes: measured on configs[4] (profiles/r04_find_pmc.txt) the fill sends
// 93 M read and 139 M write requests for 50 M records -- every record drags its own 128-byte line of (end, index) pairs from
// HBM (5.9 GB for 0.4 GB of index) and leaves its ~5 hits as single 4-byte stores -- and the copy another 118 M; both run at the
// same ~135 G requests per second.  What changes here:
//   1. The tile sort orders a tile by HALF buckets (bm_tile_sort_kernel<.., SUB = 2>): a (tile, bucket) run is itself sorted
//      by half, and the table of half-bucket runs is transposed like the bucket table (runT2[half][tile]).
//   2. The index is cut, once per sealed index and unit size, into PIECES: consecutive half buckets of one unit whose
//      candidates -- every rank a record of the piece can reach walking down from hi = #{start < qe} -- fit one LDS WINDOW of
//      FX_CAPW (end, index) pairs.  A fill workgroup stages the piece's window once (coalesced), then walks the piece's
//      records of ALL tiles as one flat sequence (a wave takes 64 tiles' runs at a time, laid end to end under a prefix sum):
//      no index line is fetched twice, no lookup: `hi` and the count come from the count half (`hc`, sl_count_record_hc).
//   3. A wave's hits are collected in LDS and leave as 16-byte stores per record (20 bytes per record on configs[4]: two
//      stores instead of five, and neighbouring lanes = neighbouring records of a run write neighbouring bytes).
//   4. The CSR offsets need no scan over the queries: the un-permute kernel leaves the sum of every 1024 queries' counts and of
//      every tile (bm_unpermute_kernel<.., FIND = 2>), one workgroup scans the TILE sums (fx_tile_scan_kernel), and the copy
//      kernel finishes the offsets of its 1024 queries itself while it moves their hits (fx_hits_copy2_kernel).  It also reads
//      the scratch offset of a query from a query-order array the un-permute kernel wrote, instead of gathering it by slot.
//      Two consecutive queries per lane (fx_hits_copy2_kernel): twice the gathers in flight per wave.
//   5. Lists that fit the memory-side cache skip the scratch and the copy (ivl.fx_direct): the un-permute kernel leaves prefixes
//      in QUERY order (FIND = 3), the fill writes straight into the CSR list, fx_offsets_kernel turns the prefixes into offsets.
//   6. (round 6) The passes of the fill are a software pipeline with a COUNTED number of stores: every pass of 64 records issues
//      exactly FX_SQ 16-byte and three 4-byte store instructions (lanes with nothing to store write to words nobody reads), so the
//      wait for a pass's records is `vmcnt(6)` = "all but the last pass's stores" where round 5's loop drained everything the
//      wave had sent, the acknowledgements of its own stores included, at the top of every pass; the pass loop is unrolled by two
//      with swapped register sets (no register copied while a load is on its way to it), a batch's run words are used as loaded
//      (round 5 computed from them where they were requested: a wait), the rare global accesses of the walk are hand-issued (no
//      FLAT instructions), and the walk reads four candidates per round trip to the LDS: 1.13-1.15 -> 1.04 ms on configs[4].
//      (Measured on top and not kept, round 6: a record's hits in whole 16-byte SLOTS of the scratch list -- the un-permute kernel
//      scanning the counts rounded up to four, the tiles' stretches from a scan of the padded totals, every store of the fill one
//      aligned 16-byte request, -28 % write requests: fill 1092 -> 1025 us, copy 807 -> 836 (its gathers meet sparser lines), find
//      2.93-2.96 ms either way.)
// A record whose walk leaves the staged window (long targets far below, piles larger than the window) reads the pairs from HBM
// as before: exact either way.
#pragma once

namespace bxmi {

constexpr int FX_NBK = 2 * BM_NB;     // half buckets
constexpr int FX_CAPW = 15360;        // (end, index) pairs of one piece's window in LDS (120 KB)
constexpr int FX_BACK = 512;          // pairs staged below the lowest `hi` of a piece: the walk of an ordinary record ends inside
constexpr int FX_HCAP = 512;          // hits a wave collects in LDS per pass of 64 records (mean 320 on configs[4])
#ifndef FX_BT_V
#define FX_BT_V 32
#endif
constexpr int FX_BT = FX_BT_V;       // tiles of a wave's batch (one run per lane; 64 left 16 waves with 24 batches on configs[4])
constexpr int FX_THREADS = 1024;
constexpr int FX_NW = FX_THREADS / 64;
constexpr size_t FX_LDS_BYTES = (size_t)FX_CAPW * 8 + (size_t)FX_NW * (FX_HCAP * 4 + 64 * 4);
// (round 5's diagnostic builds on configs[4]: 1.16 ms whole, 0.78 without the stores, 0.36 without the walk as well, 0.32 without the
// window staging too)
#ifndef FX_SQ
#define FX_SQ 3       // 16-byte stores every pass issues per lane (records of up to 4 FX_SQ + 3 hits; longer ones: a rare loop behind)
#endif
#ifndef FX_XCD
#define FX_XCD 1      // 1: neighbouring pieces are handed out on ONE XCD (eight counters), 0: one counter for the chip
#endif

// ranks at half-bucket boundary sb (first coordinate x = cmin + sb * W / 2), sb = 0 .. FX_NBK:
//   x = #{start < x}, y = #{start < x + SL_MARGIN} (no record's qe reaches further: its length is below 2^(32 - rshift) <= SL_MARGIN)
__global__ __launch_bounds__(256) void fx_meta_kernel(const int32_t *__restrict__ s_ord, int n, int32_t cmin, int shift, int2 *__restrict__ meta2)
{
    const int sb = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (sb > FX_NBK) return;
    const long long x = (long long)cmin + ((long long)sb << (shift - 1));
    meta2[sb] = make_int2(bm_rank_lt64(s_ord, n, x), bm_rank_lt64(s_ord, n, x + SL_MARGIN));
}

// A piece: half buckets [sb0, sb1) of one unit, its window = the ranks [wlo, whi) of the start-ordered index.
struct FxPiece {
    int sb0, sb1, wlo, whi;
};

// Exclusive scan of the tiles' hit totals (one workgroup): tile_base[t] = hits of the tiles before t, tile_base[ntp] = all
// hits, also written to the caller's offsets[nq].
__global__ __launch_bounds__(1024) void fx_tile_scan_kernel(const unsigned long long *__restrict__ tile_tot, int64_t ntp, long long *__restrict__ tile_base,
                                                            long long *__restrict__ grand_total, long long *__restrict__ max_tile = nullptr,
                                                            const unsigned *__restrict__ gate = nullptr)
{
    if (gate && *gate != 0) return;  // (the sorted find's chain on a batch that turned out not to be sorted)
    __shared__ long long lds[16];
    __shared__ unsigned long long s_max;
    if (threadIdx.x == 0) s_max = 0ull;
    long long carry = 0, mine = 0;
    // eight consecutive totals per thread and turn (the sorted find scans 12 207 chunk totals: twelve turns of one total per thread,
    // two barriers each, were 25 us of a 1.2 ms call)
    constexpr int PER = 8;
    for (int64_t base = 0; base < ntp; base += 1024 * PER) {
        const int64_t i0 = base + (int64_t)threadIdx.x * PER;
        long long v[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            v[k] = i0 + k < ntp ? (long long)tile_tot[i0 + k] : 0ll;
            mine = v[k] > mine ? v[k] : mine;
            sum += v[k];
        }
        long long total;
        long long run = carry + block_exclusive_scan(sum, OpSum(), 0ll, lds, &total);
#pragma unroll
        for (int k = 0; k < PER; k++) {
            if (i0 + k < ntp) tile_base[i0 + k] = run;
            run += v[k];
        }
        carry += total;
    }
    // the largest tile total rides behind the grand total: the prefixes INSIDE a tile are 31-bit (bit 31 marks an escape), so a
    // tile with 2^31 hits or more sends the batch to the bucketed find (ivl_find_fx)
    if (max_tile && mine) atomicMax(&s_max, (unsigned long long)mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_base[ntp] = carry;
        if (grand_total) *grand_total = carry;
        if (max_tile) *max_tile = (long long)s_max;
    }
}

typedef int fx_v4a4 __attribute__((ext_vector_type(4), aligned(4)));

// Hand-issued global accesses of the fill's rare paths : not seen by the compiler's count of outstanding operations.
typedef int fxp_v2i __attribute__((ext_vector_type(2)));
typedef int fxp_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int2 fxp_global_pair(const int2 *p)
{
    fxp_v2i v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_int2(v.x, v.y);
}
__device__ __forceinline__ unsigned fxp_global_u32(const unsigned *p)
{
    unsigned v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void fxp_global_store(int32_t *p, int v)
{
    asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void fxp_global_store4(int32_t *p, fx_v4a4 v)
{
    fxp_v4i w;
    w.x = v.x, w.y = v.y, w.z = v.z, w.w = v.w;
    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(w) : "memory");
}

// #{i in [0, 64) : ends[i] <= s}, ends non-decreasing (a wave's inclusive prefix sums in LDS): six halvings.
__device__ __forceinline__ unsigned fx_locate(const unsigned *ends, unsigned s)
{
    unsigned r = 0u;
#pragma unroll
    for (unsigned step = 32u; step >= 1u; step >>= 1)
        if (ends[r + step - 1u] <= s) r += step;
    return r < 63u ? r : 63u;
}

// The fill.  Persistent workgroups (one per CU: the window takes 120 KB of its LDS) draw (piece, tile chunk) pairs from a
// counter.  recs / hc / cnt / loff are in tile-sorted order (tile t's slots at t << tile_log2); tile_base[t] = first hit of
// tile t's region in tmp_hits.
__global__ __launch_bounds__(FX_THREADS) void fx_fill_kernel(const BmSeg *__restrict__ segs, const FxPiece *__restrict__ pieces, int npieces, int nchunks,
                                                            int tiles_per_chunk, const unsigned *__restrict__ runT2 /* [FX_NBK][ntp] */, int64_t ntp,
                                                            const unsigned *__restrict__ recs, const unsigned *__restrict__ hc,
                                                            const unsigned *__restrict__ cnt, const unsigned *__restrict__ loff,
                                                            const long long *__restrict__ tile_base, const int2 *__restrict__ eid /* at index 0 */,
                                                            const int2 *__restrict__ meta2, int32_t *__restrict__ tmp_hits, int tile_log2,
                                                            unsigned *__restrict__ work_counter, int32_t *__restrict__ nobody /* 16 bytes nobody reads */)
{
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    __shared__ int s_work;
    int2 *const s_win = reinterpret_cast<int2 *>(dyn);                                  // [FX_CAPW]
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    int32_t *const st = dyn + 2 * FX_CAPW + wave * (FX_HCAP + 64);                       // [FX_HCAP] this wave's hits
    unsigned *const re = reinterpret_cast<unsigned *>(st + FX_HCAP);                    // [64] where each run of the wave's batch ends
    const BmSeg &sg = segs[0];
    const BmGeom g = sg.g;
    const int ntiles = (int)sg.ntiles;
    const unsigned omask = (1u << g.rshift) - 1u;
    const int nwork = npieces * nchunks;
    // neighbouring pieces write neighbouring bytes of every tile's region: they are handed out on ONE XCD (blockIdx -> XCD is
    // round robin), so that a line shared by two pieces is merged in one L2
    const int xcd = FX_XCD ? (int)(blockIdx.x & 7u) : 0;
    const int per_xcd = FX_XCD ? (nwork + 7) >> 3 : nwork;
    const int w_lo = xcd * per_xcd, w_hi = w_lo + per_xcd < nwork ? w_lo + per_xcd : nwork;
    for (;;) {
        if (threadIdx.x == 0) s_work = w_lo + (int)atomicAdd(work_counter + xcd, 1u);
        __syncthreads();
        const int work = s_work;
        if (work >= w_hi) break;
        const FxPiece pc = pieces[work / nchunks];
        const int t0 = (work % nchunks) * tiles_per_chunk;
        const int t1 = t0 + tiles_per_chunk < ntiles ? t0 + tiles_per_chunk : ntiles;
        const int wlo = pc.wlo, whi = pc.whi;
        // the piece's unit: its first coordinate and the rank of that coordinate among the starts
        const int unit = pc.sb0 >> (g.f + 1);
        const long long lo_u = (long long)g.cmin + ((long long)unit << (g.shift + g.f));
        const int sLo = meta2[unit << (g.f + 1)].x;
        // stage the window: two pairs per 16-byte load, ALL of a thread's loads requested before the first is written to LDS
        // (one load, one wait, one LDS store per round made the staging eight dependent round trips to HBM per piece)
        {
            const int nw = whi - wlo;
            const int2 *__restrict__ src = eid + wlo;
            constexpr int ROUNDS = (FX_CAPW + 2 * FX_THREADS - 1) / (2 * FX_THREADS);
            sl_v4a8 v[ROUNDS];
#pragma unroll
            for (int r = 0; r < ROUNDS; r++) {
                const int i = 2 * ((int)threadIdx.x + r * FX_THREADS);
                // (the pair behind the window's last one is read with it when the window's length is odd: the pair array ends
                // with SL_WALK spare entries... in front; behind, a valid address is all that is needed -- clamp)
                const int ia = i + 1 < nw ? i : (nw >= 2 ? nw - 2 : 0);
                v[r] = *reinterpret_cast<const sl_v4a8 *>(src + ia);
            }
#pragma unroll
            for (int r = 0; r < ROUNDS; r++) {
                const int i = 2 * ((int)threadIdx.x + r * FX_THREADS);
                if (i + 1 < nw)
                    *reinterpret_cast<int4 *>(s_win + i) = make_int4(v[r].x, v[r].y, v[r].z, v[r].w);
                else if (i < nw)  // the last pair of an odd window: it is the SECOND pair of the clamped load (pairs nw - 2, nw - 1)
                    s_win[i] = nw >= 2 ? make_int2(v[r].z, v[r].w) : src[i];
            }
        }
        __syncthreads();
        const unsigned *__restrict__ runs0 = runT2 + (int64_t)pc.sb0 * ntp;
        const unsigned *__restrict__ runs1 = runT2 + (int64_t)(pc.sb1 - 1) * ntp;
        // A wave takes FX_BT tiles at a time (one run per lane), and the kernel is a chain of dependent loads -- run table ->
        // records -> stores -- on ONE workgroup per CU: the next batch's runs and the next pass's records are requested before
        // the current ones are worked on.
        struct Run {  // (as loaded: nothing is computed from a run's words before its batch starts -- that would be a wait where they are requested)
            unsigned r0, r1;
            long long tbase;
        };
        auto load_run = [&](int tb, Run &R) {  // the piece's records of tile tb + lane (past t1: a valid address, the words unused)
            const int t = tb + lane;
            const int ta = lane < FX_BT && t < t1 ? t : t0;
            R.r0 = runs0[ta], R.r1 = runs1[ta], R.tbase = tile_base[ta];
        };
        Run cur_run;
        load_run(t0 + FX_BT * wave, cur_run);
        for (int tb = t0 + FX_BT * wave; tb < t1; tb += FX_BT * FX_NW) {
            Run next_run;
            load_run(tb + FX_BT * FX_NW, next_run);  // (past t1: nothing is loaded)
            const int t = tb + lane;
            const bool live_run = lane < FX_BT && t < t1;
            const unsigned a = live_run ? cur_run.r0 & 0xffffu : 0u;
            const unsigned rlen = live_run ? (cur_run.r1 & 0xffffu) + (cur_run.r1 >> 16) - a : 0u;
            const long long tbase = cur_run.tbase;
            const unsigned rincl = wave_inclusive_sum_dpp(rlen);
            const unsigned T = (unsigned)__builtin_amdgcn_readlane((int)rincl, 63);
            const unsigned rdelta = ((unsigned)t << tile_log2) + a - (rincl - rlen);  // + s = the tile-sorted position of the batch's record s
            re[lane] = rincl;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            struct Rec {
                unsigned at, rec, h, lo;
                long long tb_r;
                bool act;
            };
            auto fetch = [&](unsigned p0, Rec &R) {
                const unsigned s = p0 + (unsigned)lane;
                R.act = s < T;
                const unsigned r = fx_locate(re, s);
                // (every lane takes part in the shuffles: a lane that is switched off answers a ds_bpermute with ZERO, and in the
                // batch's last pass the lane that owns the last run is usually beyond the pass's records)
                const unsigned rd_r = (unsigned)__shfl((int)rdelta, (int)r, 64);
                R.tb_r = __shfl(tbase, (int)r, 64);
                R.at = R.act ? rd_r + s : ((unsigned)tb << tile_log2);
                R.rec = recs[(size_t)R.at], R.h = hc[(size_t)R.at], R.lo = loff[(size_t)R.at];
            };
            // One pass of 64 records.  R: its records (requested a pass ago), N: where the next pass's records go -- the loop below is
            // unrolled by two with the sets swapped, so no register is copied while a load is on its way to it.  Every pass issues
            // EXACTLY FX_SQ 16-byte and three 4-byte store instructions, straight line (a lane with nothing to store writes to words
            // nobody reads), so the compiler can count them: the wait for a pass's records is "all but the last pass's stores", not
            // a drain of everything the wave has sent (memory operations retire in order, stores included; round 5's loop of
            // `while any lane has hits left` stores made every pass wait for the acknowledgement of the pass before).  The rare
            // global accesses of the walk are hand-issued: where an LDS access and a global one meet in one value the compiler makes
            // FLAT instructions of both and drains both counters around every one.
            auto static_stores = [&](const unsigned nn, const unsigned my_off, int32_t *dst) {
                const unsigned quads = nn >> 2, tail0 = nn & ~3u;
#pragma unroll
                for (unsigned q = 0; q < (unsigned)FX_SQ; q++) {
                    const bool ok = q < quads;
                    const unsigned s = ok ? my_off + 4u * q : 0u;
                    fx_v4a4 v;
                    v.x = st[s], v.y = st[s + 1u], v.z = st[s + 2u], v.w = st[s + 3u];
                    fx_v4a4 *d = ok ? reinterpret_cast<fx_v4a4 *>(dst + 4u * q) : reinterpret_cast<fx_v4a4 *>(nobody);
                    *d = v;
                }
#pragma unroll
                for (unsigned u = 0; u < 3u; u++) {
                    const bool ok = tail0 + u < nn;
                    int32_t *d = ok ? dst + tail0 + u : nobody;
                    *d = st[ok ? my_off + tail0 + u : 0u];
                }
                if (__any(quads > (unsigned)FX_SQ)) {  // records with more hits than the static stores cover: rare, not counted (never waited for)
                    for (unsigned q = (unsigned)FX_SQ; __any(q < quads); q++)
                        if (q < quads) {
                            const unsigned s = my_off + 4u * q;
                            fx_v4a4 v;
                            v.x = st[s], v.y = st[s + 1u], v.z = st[s + 2u], v.w = st[s + 3u];
                            fxp_global_store4(dst + 4u * q, v);
                        }
                }
            };
            auto pair_far = [&](int k) -> int2 { return (k >= wlo && k < whi) ? s_win[k - wlo] : fxp_global_pair(eid + k); };
            auto pass = [&](Rec &R, Rec &N, const unsigned p0) {
                const bool act = R.act;
                const unsigned at = R.at, rec = R.rec, h = R.h, lo = R.lo;
                const long long tb_r = R.tb_r;
                fetch(p0 + 64u, N);  // (past the batch's records: valid addresses, the words unused -- every pass issues the same operations)
                const bool esc = (lo >> 31) != 0u;
                unsigned n = act && !esc ? (h & 0xffffu) : 0u;
                if (__any(n == 0xffffu)) {  // (a count that did not fit the packed word: rare)
                    if (n == 0xffffu) n = fxp_global_u32(cnt + (size_t)at);
                }
                const int hi = sLo + (int)(h >> 16);
                const int qs = (int)(lo_u + (long long)(rec & omask));
                int32_t *dst = tmp_hits + tb_r + (long long)(lo & 0x7FFFFFFFu);
                // (prefix sums of the counts clamped to "does not fit": a pile's counts cannot overflow them, and what fits is exact)
                const unsigned nc = n <= (unsigned)FX_HCAP ? n : (unsigned)FX_HCAP + 1u;
                const unsigned hincl = wave_inclusive_sum_dpp(nc);
                // sub-batches of lanes whose hits fit the wave's LDS image together (normally: all 64 at once)
                int first = 0;
                unsigned hbase = 0u;
                do {
                    const bool fits = lane >= first && hincl - hbase <= (unsigned)FX_HCAP;
                    const int k = __popcll(__ballot(fits));  // (hincl is monotone: the lanes that fit are first .. first + k - 1)
                    const bool direct = k == 0;              // lane `first` alone has more hits than the image holds: straight to HBM
                    const int cntl = direct ? 1 : k;
                    const bool in = lane >= first && lane < first + cntl;
                    const unsigned my_off = hincl - nc - hbase;
                    int c = in ? (int)n : 0;
                    int kk = hi - 1;
                    if (!direct && __any(c > 0)) {
                        auto take = [&](const int2 p) {
                            if (p.x > qs) {
                                --c;
                                st[my_off + (unsigned)c] = p.y;
                            }
                        };
                        if (__all(c <= 0 || (hi <= whi && hi - LANE_WINDOW >= wlo))) {
                            // (four candidates per round trip to the LDS: one read and one wait per candidate made the walk a chain of
                            // ~8 dependent LDS latencies per pass on four waves per SIMD; what a lane reads beyond its last hit is inside
                            // the window by the condition above)
                            static_assert(LANE_WINDOW % 4 == 0, "the walk reads four candidates at a time");
                            const int2 *wp = s_win + (c > 0 ? kk - wlo : LANE_WINDOW);
                            int used = 0;
                            for (int step = 0; step < LANE_WINDOW && __any(c > 0); step += 4) {
                                const int2 pa = wp[-step], pb = wp[-step - 1], pc = wp[-step - 2], pd = wp[-step - 3];
                                if (c > 0) take(pa), used++;
                                if (c > 0) take(pb), used++;
                                if (c > 0) take(pc), used++;
                                if (c > 0) take(pd), used++;
                            }
                            kk -= used;
                        } else {
                            for (int step = 0; step < LANE_WINDOW && c > 0 && kk >= 0; step++, kk--) {
                                if (__all(kk >= wlo && kk < whi))
                                    take(s_win[kk - wlo]);
                                else
                                    take(pair_far(kk));
                            }
                        }
                    }
                    unsigned long long m = __ballot(c > 0);  // long walks, and the record that goes straight to HBM: the wave takes them one by one
                    while (m) {
                        const int src = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        int C = __shfl(c, src, 64), K = __shfl(kk, src, 64);
                        const int S = __shfl(qs, src, 64);
                        const unsigned Rr = (unsigned)__shfl((int)my_off, src, 64);
                        int32_t *D = reinterpret_cast<int32_t *>(__shfl((long long)reinterpret_cast<uintptr_t>(dst), src, 64));
                        while (C > 0 && K >= 0) {
                            const int kx = K - lane;
                            int2 p = make_int2(INT_MIN, 0);
                            if (kx >= 0) p = pair_far(kx);
                            const bool f = kx >= 0 && p.x > S;
                            const unsigned long long fm = __ballot(f);
                            // hits at higher ranks come later in the list: lane 0 (the highest rank of the step) takes the last free slot
                            const int before = __popcll(fm & ((1ull << lane) - 1ull));
                            if (f && before < C) {
                                if (direct)
                                    fxp_global_store(D + (C - 1 - before), p.y);
                                else
                                    st[Rr + (unsigned)(C - 1 - before)] = p.y;
                            }
                            C -= __popcll(fm);
                            K -= 64;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    static_stores(in && !direct ? n : 0u, my_off, dst);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the next sub-batch / pass overwrites the image)
                    __builtin_amdgcn_wave_barrier();
                    first += cntl;
                    hbase = (unsigned)__builtin_amdgcn_readlane((int)hincl, first - 1);
                } while (first < 64 && hbase < (unsigned)__builtin_amdgcn_readlane((int)hincl, 63));
            };
            Rec RA, RB;
            {
                fetch(0u, RA);
                int zero;  // (not to the compiler: as many stores behind the first pass's records as behind every other pass's)
                asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
                static_stores((unsigned)zero, 0u, nobody);
            }
            for (unsigned p0 = 0; p0 < T; p0 += 128u) {
                pass(RA, RB, p0);
                if (p0 + 64u >= T) break;
                pass(RB, RA, p0 + 64u);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (`re` is rewritten by the next batch)
            __builtin_amdgcn_wave_barrier();
            cur_run = next_run;
        }
        __syncthreads();  // the next piece's window replaces this one; thread 0 draws the next number
    }
}

// The hit copy: a workgroup takes BM_PART_Q consecutive queries of a tile, the workgroups of a tile follow each other on one XCD,
// and the CSR offsets are finished on the way: offset(query) = tile_base[tile] + the parts of the tile before this one + an
// exclusive scan of the workgroup's own 1024 counts.  The scratch offset of a query comes from the query-order array `svq`
// (bit 31 = escape record: answered here from the sealed index).  QPL consecutive queries per lane: a wave owns the stretch of
// 64 QPL queries (~320 QPL hits on configs[4]) and has QPL times the gathers in flight -- the kernel is a chain of three round
// trips to memory per workgroup (counts, gathers, stores) at the CU's full complement of waves either way, so the work per round
// trip is what counts (one query per lane 0.91 ms, two 0.78, four 0.85: two is what ships).  1024 / QPL threads per part.
#ifndef FX_FL2_V
#define FX_FL2_V 10
#endif
template <int TILE, int QPL>
__global__ __launch_bounds__(BM_PART_Q / QPL) __attribute__((amdgpu_waves_per_eu(8))) void fx_hits_copy2_kernel(
    const BmSeg *__restrict__ segs, const unsigned *__restrict__ svq, const long long *__restrict__ tile_base, const unsigned long long *__restrict__ parts,
    const int32_t *__restrict__ tmp_hits, long long *__restrict__ offsets, int32_t *__restrict__ hits, int64_t ntp)
{
    static_assert(QPL == 2 || QPL == 4, "a lane's queries are one 8- or 16-byte load");
    constexpr int PARTS = TILE / BM_PART_Q, THREADS = BM_PART_Q / QPL, NW = THREADS / 64, WQ = 64 * QPL;
    __shared__ unsigned s_ends[NW][WQ];  // per wave: where each query's hits end in the wave's stretch
    __shared__ unsigned s_dsrc[NW][WQ], s_ddst[NW][WQ];
    __shared__ long long s_scan[16];
    const int xcd = (int)(blockIdx.x & 7u);
    const int64_t unit = (int64_t)(blockIdx.x >> 3), units = ((ntp + 7 - xcd) >> 3) * PARTS;  // of this XCD's tiles
    if (unit >= units) return;
    const BmSeg &sg = segs[0];
    const int64_t tile = (unit / PARTS) * 8 + xcd;
    const int part = (int)(unit % PARTS);
    if (tile >= sg.ntiles) return;  // padding up to the next plan group
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int64_t q0 = tile * TILE;
    const int64_t left = sg.nq - q0;
    const int n = (int)(left < TILE ? left : TILE);
    const int k = part * BM_PART_Q + QPL * (int)threadIdx.x;  // this lane's queries: k .. k + QPL - 1
    if (part * BM_PART_Q >= n) return;  // (uniform)
    const int64_t q = q0 + k;
    long long part_base = tile_base[tile];
    {
        long long v = lane < part ? (long long)parts[tile * PARTS + lane] : 0ll;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        part_base += v;
    }
    // (q is a multiple of QPL and the arrays are aligned: a lane whose last query is live reads its counts and scratch offsets as one load)
    typedef unsigned fx_vqu __attribute__((ext_vector_type(QPL)));
    unsigned c[QPL], sv[QPL];
#pragma unroll
    for (int u = 0; u < QPL; u++) c[u] = 0u, sv[u] = 0u;
    if (k + QPL <= n) {
        const fx_vqu cv = __builtin_nontemporal_load(reinterpret_cast<const fx_vqu BX_GLOBAL *>(as_global(sg.counts) + q));
        const fx_vqu sq = __builtin_nontemporal_load(reinterpret_cast<const fx_vqu *>(svq + q));
#pragma unroll
        for (int u = 0; u < QPL; u++) c[u] = cv[u], sv[u] = sq[u];
    } else {
#pragma unroll
        for (int u = 0; u < QPL; u++)
            if (k + u < n) c[u] = (unsigned)as_global(sg.counts)[q + u], sv[u] = svq[q + u];
    }
    long long mine = 0;
#pragma unroll
    for (int u = 0; u < QPL; u++) mine += (long long)c[u];
    long long total;
    long long o[QPL + 1];
    o[0] = part_base + block_exclusive_scan(mine, OpSum(), 0ll, s_scan, &total);
#pragma unroll
    for (int u = 0; u < QPL; u++) o[u + 1] = o[u] + (long long)c[u];
    if (k + QPL <= n) {
        typedef long long fx_v2ll __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int u = 0; u < QPL; u += 2) __builtin_nontemporal_store(fx_v2ll{o[u], o[u + 1]}, reinterpret_cast<fx_v2ll *>(offsets + q + u));
    } else {
#pragma unroll
        for (int u = 0; u < QPL; u++)
            if (k + u < n) offsets[q + u] = o[u];
    }
    const int32_t *__restrict__ region = tmp_hits + tile_base[tile];
    unsigned nn[QPL], mine_n = 0u;  // (escape records leave holes in the stretch, filled by their own lanes)
#pragma unroll
    for (int u = 0; u < QPL; u++) nn[u] = (sv[u] >> 31) ? 0u : c[u], mine_n += nn[u];
    const unsigned incl = wave_inclusive_sum_dpp(mine_n);
    const unsigned wtotal = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    const long long o_first = __shfl(o[0], 0, 64);
    int32_t *__restrict__ out = hits + o_first;
    unsigned *ends = s_ends[wave], *dsrc = s_dsrc[wave], *ddst = s_ddst[wave];
    {
        unsigned b = incl - mine_n;  // hits of the stretch before the lane's first query
#pragma unroll
        for (int u = 0; u < QPL; u++) {
            dsrc[QPL * lane + u] = sv[u] - b;                            // + s = the hit's place in the tile's region
            ddst[QPL * lane + u] = (unsigned)(o[u] - o_first) - b;       // + s = its place behind the wave's first CSR offset
            b += nn[u];
            ends[QPL * lane + u] = b;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // (Round 6 made the gathers and stores of a turn unconditional -- a fixed number per turn, so that the compiler counts them and
    // every store waits for its own gather only, vmcnt(9), where behind these wave-uniform branches each one waits for everything
    // before it: 811 -> 902 us.  The kernel is bound by the requests it sends (one per query's stretch of the scratch list), not by
    // the chain; the dummy accesses of the rows beyond the wave's hits are requests too.  Not kept.)
    constexpr int FL = FX_FL2_V;
    for (unsigned s0 = 0; s0 < wtotal; s0 += 64u * FL) {
        unsigned dst[FL];
        bool act[FL];
        int v[FL];
#pragma unroll
        for (int j = 0; j < FL; j++)
            if (s0 + 64u * j < wtotal) {  // (wave-uniform)
                const unsigned s = s0 + 64u * j + (unsigned)lane;
                unsigned r = 0u;  // #{i in [0, WQ) : ends[i] <= s}, at most WQ - 1
#pragma unroll
                for (unsigned step = WQ / 2; step >= 1u; step >>= 1)
                    if (ends[r + step - 1u] <= s) r += step;
                r = r < (unsigned)(WQ - 1) ? r : (unsigned)(WQ - 1);
                act[j] = s < wtotal;
                dst[j] = ddst[r] + s;
                v[j] = region[act[j] ? dsrc[r] + s : 0u];
            }
#pragma unroll
        for (int j = 0; j < FL; j++)
            if (s0 + 64u * j < wtotal && act[j]) out[dst[j]] = v[j];
    }
    bool any_esc = false;
#pragma unroll
    for (int u = 0; u < QPL; u++) any_esc |= (sv[u] >> 31) != 0u;
    if (any_esc) {  // escape records: rare, answered from the sealed index by their own lane
        const IndexDev ix = sg.ix;
#pragma unroll
        for (int u = 0; u < QPL; u++) {
            int cc = (int)c[u];
            if (!(sv[u] >> 31) || cc <= 0) continue;
            const int qs = sg.qs[q + u], qe = sg.qe[q + u];
            int32_t *__restrict__ d = hits + o[u];
            for (int j = global_rank_lt(ix.s_ord, 0, ix.n, qe) - 1; cc > 0 && j >= 0; j--)
                if (ix.e_ord[j] > qs) d[--cc] = ix.idx[j];
        }
    }
}

// The fill straight into the CSR list (bm_unpermute_kernel<.., FIND = 3>): what is left of the copy -- the int64 offsets
// (tile base + the query-order prefix inside the tile) and the hits of escape queries, from the sealed index by their own lanes.
// The escapes' hits are only written when the whole list fits the caller's buffer (tile_base[ntiles] = all hits): BXMI_ERANGE
// leaves the buffer alone.
__global__ __launch_bounds__(256) void fx_offsets_kernel(const BmSeg *__restrict__ segs, const unsigned *__restrict__ svq, const long long *__restrict__ tile_base,
                                                         int64_t ntiles, int tile_log2, long long *__restrict__ offsets, int32_t *__restrict__ hits, int64_t cap)
{
    const BmSeg &sg = segs[0];
    const int64_t nq = sg.nq;
    const int64_t q0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (q0 >= nq) return;
    const bm_v4i v4 = __builtin_nontemporal_load(reinterpret_cast<const bm_v4i *>(svq + q0));  // (padded to whole tiles)
    const long long tb = tile_base[q0 >> tile_log2];
    const unsigned v[4] = {(unsigned)v4.x, (unsigned)v4.y, (unsigned)v4.z, (unsigned)v4.w};
    long long o[4];
#pragma unroll
    for (int u = 0; u < 4; u++) o[u] = tb + (long long)(v[u] & 0x7FFFFFFFu);
    if (q0 + 4 <= nq) {
        typedef long long fx_v2ll __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(fx_v2ll{o[0], o[1]}, reinterpret_cast<fx_v2ll *>(offsets + q0));
        __builtin_nontemporal_store(fx_v2ll{o[2], o[3]}, reinterpret_cast<fx_v2ll *>(offsets + q0 + 2));
    } else {
        for (int u = 0; u < 4 && q0 + u < nq; u++) offsets[q0 + u] = o[u];
    }
    if (((v[0] | v[1] | v[2] | v[3]) >> 31) && hits && tile_base[ntiles] <= cap) {  // escape records: rare
        const IndexDev ix = sg.ix;
        for (int u = 0; u < 4 && q0 + u < nq; u++) {
            if (!(v[u] >> 31)) continue;
            int cc = sg.counts[q0 + u];
            if (cc <= 0) continue;
            const int qs = sg.qs[q0 + u], qe = sg.qe[q0 + u];
            int32_t *__restrict__ dst = hits + o[u];
            for (int j = global_rank_lt(ix.s_ord, 0, ix.n, qe) - 1; cc > 0 && j >= 0; j--)
                if (ix.e_ord[j] > qs) dst[--cc] = ix.idx[j];
        }
    }
}

}  // namespace bxmi
