// primitives.hpp -- hand-written device-wide primitives for gfx950 (wave64):
//   * three-phase scan (sum / max), int32 or int64 accumulators
//   * LSD radix sort of 32/64-bit keys, 8 bits per pass, stable, with
//     wave-ballot digit matching for the in-tile ranks and pass skipping
// They serve the index build (sort by the treap's in-order key, sorted ends,
// prefix max of ends) and the CSR offsets of batched find().
// All are HBM-bandwidth-bound integer kernels; no MFMA anywhere.
#pragma once
#include "common.hpp"

namespace bxmi {

// ============================================================================
// scan
// ============================================================================
struct OpSum {
    template <typename T>
    __device__ __forceinline__ T operator()(T a, T b) const
    {
        return a + b;
    }
};
struct OpMax {
    template <typename T>
    __device__ __forceinline__ T operator()(T a, T b) const
    {
        return a > b ? a : b;
    }
};

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

template <typename T, typename Op>
__device__ __forceinline__ T wave_inclusive_scan(T v, Op op)
{
    int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d, 64);
        if (lane >= d) v = op(o, v);
    }
    return v;
}

// The same for 32-bit sums through DPP (gfx9 family: row_shr inside the 16-lane rows, then row_bcast:15 / row_bcast:31 carry the
// rows' totals): six VALU instructions, where the shuffle version is a chain of six dependent ds_bpermute round trips through the
// LDS pipe -- for kernels that scan once per pass of 64 items (find_exchange.hpp).
__device__ __forceinline__ unsigned wave_inclusive_sum_dpp(unsigned v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return (unsigned)x;
}

// Exclusive prefix of per-thread values across a 256-thread block.
// Returns the exclusive prefix for this thread; *block_total gets the total.
template <typename T, typename Op>
__device__ __forceinline__ T block_exclusive_scan(T v, Op op, T identity, T *lds /* >= 8 slots */, T *block_total)
{
    int lane = lane_id(), w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    T inc = wave_inclusive_scan(v, op);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    T wave_off = identity, total = identity;
    for (int i = 0; i < nw; i++) {
        T x = lds[i];
        if (i < w) wave_off = op(wave_off, x);
        total = op(total, x);
    }
    T prev = __shfl_up(inc, 1, 64);
    T exc = lane == 0 ? identity : prev;
    __syncthreads();  // lds reusable by the caller afterwards
    *block_total = total;
    return op(wave_off, exc);
}

template <typename In, typename Acc, typename Op>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const In *__restrict__ in, int64_t n, Acc identity,
                                                                  Acc *__restrict__ block_sums)
{
    __shared__ Acc lds[8];
    Op op;
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    Acc acc = identity;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        int64_t i = base + (int64_t)j * SCAN_THREADS + threadIdx.x;
        if (i < n) acc = op(acc, (Acc)in[i]);
    }
    Acc total;
    (void)block_exclusive_scan(acc, op, identity, lds, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// Single workgroup: exclusive scan of the per-tile sums, in place.
template <typename Acc, typename Op>
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(Acc *__restrict__ sums, int64_t nblocks, Acc identity,
                                                              Acc *__restrict__ grand_total)
{
    __shared__ Acc lds[16];
    Op op;
    Acc carry = identity;
    for (int64_t base = 0; base < nblocks; base += blockDim.x) {
        int64_t i = base + threadIdx.x;
        Acc v = i < nblocks ? sums[i] : identity;
        Acc total;
        Acc exc = block_exclusive_scan(v, op, identity, lds, &total);
        if (i < nblocks) sums[i] = op(carry, exc);
        carry = op(carry, total);
    }
    if (threadIdx.x == 0 && grand_total) *grand_total = carry;
}

template <typename In, typename Acc, typename Op, bool INCLUSIVE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const In *in, Acc *out, int64_t n, Acc identity,
                                                                 const Acc *__restrict__ block_sums)
{
    __shared__ Acc lds[8];
    Op op;
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    Acc v[SCAN_ITEMS];
    Acc run = identity;
    // a thread's SCAN_ITEMS consecutive inputs as 16-byte loads, its outputs as 16-byte stores, when the types and the
    // alignment allow (the CSR offsets of find(): int32 counts in, int64 offsets out)
    const bool wide = sizeof(In) == 4 && (sizeof(Acc) == 8 || sizeof(Acc) == 4) && SCAN_ITEMS % 4 == 0 && base + SCAN_ITEMS <= n &&
                      ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (wide) {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j += 4) {
            const int4 w = *reinterpret_cast<const int4 *>(reinterpret_cast<const int32_t *>(in) + base + j);
            const int32_t q[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                In x;
                __builtin_memcpy(&x, &q[u], sizeof(In) == 4 ? 4 : 0);
                v[j + u] = (Acc)x;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) v[j] = (base + j < n) ? (Acc)in[base + j] : identity;
    }
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) run = op(run, v[j]);
    Acc total;
    Acc off = op(block_sums[blockIdx.x], block_exclusive_scan(run, op, identity, lds, &total));
    Acc r[SCAN_ITEMS];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        Acc inc = op(off, v[j]);
        r[j] = INCLUSIVE ? inc : off;
        off = inc;
    }
    if (wide) {
        constexpr int PER = 16 / (int)sizeof(Acc);  // outputs per 16-byte store
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j += PER) {
            int4 w;
            __builtin_memcpy(&w, &r[j], 16);
            *reinterpret_cast<int4 *>(out + base + j) = w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++)
            if (base + j < n) out[base + j] = r[j];
    }
}

// out[i] = scan(in[0..i]) (inclusive) or scan(in[0..i)) (exclusive).  `grand_total`
// (device, optional) receives the reduction of everything; for the exclusive sum
// this is what CSR callers store at offsets[n].  `scratch` holds the tile sums.
template <typename In, typename Acc, typename Op, bool INCLUSIVE>
int device_scan(const In *in, Acc *out, int64_t n, Acc identity, Acc *grand_total, DevBuf &scratch, hipStream_t st)
{
    if (n <= 0) {
        if (grand_total) BXMI_HIP(hipMemcpyAsync(grand_total, &identity, sizeof(Acc), hipMemcpyHostToDevice, st));
        return BXMI_OK;
    }
    int64_t nblocks = div_up(n, SCAN_TILE);
    BXMI_TRY(scratch.reserve((size_t)nblocks * sizeof(Acc)));
    Acc *sums = scratch.as<Acc>();
    hipLaunchKernelGGL((scan_reduce_kernel<In, Acc, Op>), dim3((unsigned)nblocks), dim3(SCAN_THREADS), 0, st, in, n, identity, sums);
    hipLaunchKernelGGL((scan_block_sums_kernel<Acc, Op>), dim3(1), dim3(1024), 0, st, sums, nblocks, identity, grand_total);
    hipLaunchKernelGGL((scan_apply_kernel<In, Acc, Op, INCLUSIVE>), dim3((unsigned)nblocks), dim3(SCAN_THREADS), 0, st, in, out, n,
                       identity, sums);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// ============================================================================
// radix sort (keys only)
// ============================================================================
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = 16;                      // keys per lane
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;    // 4096 keys per workgroup
constexpr int RS_RADIX = 256;

template <typename K>
__device__ __forceinline__ unsigned rs_digit(K k, int shift)
{
    return (unsigned)(k >> shift) & 0xFFu;
}

// Histogram of every 8-bit digit of every key in one pass over HBM (decides
// which passes can be skipped because all keys share that digit).
template <typename K>
__global__ __launch_bounds__(RS_THREADS) void rs_all_digits_kernel(const K *__restrict__ keys, int64_t n,
                                                                  unsigned *__restrict__ global_hist /* [sizeof(K)][256] */)
{
    constexpr int ND = (int)sizeof(K);
    __shared__ unsigned h[ND * RS_RADIX];
    for (int i = threadIdx.x; i < ND * RS_RADIX; i += RS_THREADS) h[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * RS_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * RS_THREADS) {
        K k = keys[i];
#pragma unroll
        for (int d = 0; d < ND; d++) atomicAdd(&h[d * RS_RADIX + rs_digit(k, 8 * d)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ND * RS_RADIX; i += RS_THREADS)
        if (h[i]) atomicAdd(&global_hist[i], h[i]);
}

// Per-tile digit counts for one pass, stored digit-major: hist[d * ntiles + tile].
template <typename K>
__global__ __launch_bounds__(RS_THREADS) void rs_tile_hist_kernel(const K *__restrict__ keys, int64_t n, int shift,
                                                                 unsigned *__restrict__ hist, int64_t ntiles)
{
    __shared__ unsigned h[RS_RADIX];
    h[threadIdx.x] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int j = 0; j < RS_ITEMS; j++) {
        int64_t i = base + (int64_t)j * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[rs_digit(keys[i], shift)], 1u);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// Stable scatter of one tile.  A wave owns RS_ITEMS*64 consecutive keys and
// walks them 64 at a time; lanes holding the same digit find each other with
// eight ballots, so the in-round rank is a popcount and only one lane per
// digit touches the wave's LDS counter.
template <typename K>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const K *__restrict__ in, K *__restrict__ out, int64_t n,
                                                               int shift, const unsigned *__restrict__ scanned_hist,
                                                               int64_t ntiles)
{
    __shared__ unsigned wcnt[RS_WAVES][RS_RADIX];
    __shared__ unsigned wbase[RS_WAVES][RS_RADIX];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RS_WAVES * RS_RADIX; i += RS_THREADS) (&wcnt[0][0])[i] = 0;
    __syncthreads();

    const int64_t wbeg = (int64_t)blockIdx.x * RS_TILE + (int64_t)wave * (RS_ITEMS * 64);
    const unsigned long long lt = lanemask_lt();
    K key[RS_ITEMS];
    unsigned rank[RS_ITEMS];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        int64_t i = wbeg + r * 64 + lane;
        bool valid = i < n;
        key[r] = valid ? in[i] : (K)0;
        unsigned d = valid ? rs_digit(key[r], shift) : 0u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            bool bit = (d >> b) & 1u;
            unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        unsigned before = (unsigned)__popcll(peers & lt);
        unsigned prev = wcnt[wave][d];  // every peer reads the same counter (LDS broadcast)
        rank[r] = prev + before;
        if (valid && before == 0) wcnt[wave][d] = prev + (unsigned)__popcll(peers);
    }
    __syncthreads();
    {
        // thread t owns digit t: global base of (digit, tile) then a prefix over the waves
        unsigned run = scanned_hist[(int64_t)threadIdx.x * ntiles + blockIdx.x];
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) {
            wbase[w][threadIdx.x] = run;
            run += wcnt[w][threadIdx.x];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        int64_t i = wbeg + r * 64 + lane;
        if (i < n) out[wbase[wave][rs_digit(key[r], shift)] + rank[r]] = key[r];
    }
}

struct SortScratch {
    DevBuf hist;       // per-tile histograms of the current pass
    DevBuf scan_tmp;   // tile sums of the scan over hist
    DevBuf digits;     // all-digit histogram (sizeof(K) * 256 counters)
};

// Sorts keys[0..n) ascending (unsigned compare).  `keys` and `tmp` ping-pong;
// *result points at whichever holds the sorted data.  Passes whose digit is
// constant over the whole input are skipped (one tiny D2H read decides).
template <typename K>
int radix_sort_keys(K *keys, K *tmp, int64_t n, K **result, SortScratch &sc, hipStream_t st)
{
    *result = keys;
    if (n <= 1) return BXMI_OK;
    if (n >= (int64_t)1 << 31) return fail(BXMI_EINVAL, "radix_sort_keys: n=%lld too large", (long long)n);
    constexpr int ND = (int)sizeof(K);
    int64_t ntiles = div_up(n, RS_TILE);
    BXMI_TRY(sc.digits.reserve(ND * RS_RADIX * sizeof(unsigned)));
    BXMI_TRY(sc.hist.reserve((size_t)ntiles * RS_RADIX * sizeof(unsigned)));
    unsigned *dig = sc.digits.as<unsigned>();
    BXMI_HIP(hipMemsetAsync(dig, 0, ND * RS_RADIX * sizeof(unsigned), st));
    hipLaunchKernelGGL((rs_all_digits_kernel<K>), dim3(stream_grid(n, RS_THREADS * 8)), dim3(RS_THREADS), 0, st, keys, n, dig);
    BXMI_LAUNCH_CHECK();
    unsigned host_dig[8 * RS_RADIX];
    BXMI_HIP(hipMemcpyAsync(host_dig, dig, ND * RS_RADIX * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));

    K *src = keys, *dst = tmp;
    unsigned *hist = sc.hist.as<unsigned>();
    for (int d = 0; d < ND; d++) {
        bool trivial = false;
        for (int b = 0; b < RS_RADIX; b++)
            if (host_dig[d * RS_RADIX + b] == (unsigned)n) trivial = true;
        if (trivial) continue;
        int shift = 8 * d;
        hipLaunchKernelGGL((rs_tile_hist_kernel<K>), dim3((unsigned)ntiles), dim3(RS_THREADS), 0, st, src, n, shift, hist, ntiles);
        BXMI_LAUNCH_CHECK();
        BXMI_TRY((device_scan<unsigned, unsigned, OpSum, false>(hist, hist, ntiles * RS_RADIX, 0u, nullptr, sc.scan_tmp, st)));
        hipLaunchKernelGGL((rs_scatter_kernel<K>), dim3((unsigned)ntiles), dim3(RS_THREADS), 0, st, src, dst, n, shift, hist, ntiles);
        BXMI_LAUNCH_CHECK();
        K *t = src;
        src = dst;
        dst = t;
    }
    *result = src;
    return BXMI_OK;
}

}  // namespace bxmi
