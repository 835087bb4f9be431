// bitset.hip -- the basewise bitset behind bx.bitset.BinnedBitSet.
//
// Reference (src/binBits.c over src/kent/bits.c): 1024 lazily allocated bins,
// each a byte array or one of two sentinels (ALL_ZERO / ALL_ONE); every op is a
// byte-at-a-time loop.
//
// MI355X design: the whole chromosome is ONE dense array of 64-bit words in
// HBM (512 Mi bits = 64 MiB; a 24-chromosome genome = 1.5 GiB of 288 GB), bit p
// at word p>>6, bit p&63.  AND / OR / NOT / popcount are word-parallel streams
// with 16-byte lane loads and a wave64 reduction; set_range / count_range are
// batched (one lane per BED line, a whole wave for long ranges).  The
// reference's per-bin tri-state survives as a tiny side array of tags: it is
// unobservable except through two quirks we reproduce bit-exactly --
// count_range on an ALL_ONE first bin subtracts the start offset
// (binBits.c:155,161) and invert() turns untouched bins into ALL_ONE.
// Everything here is HBM-bandwidth bound integer work: no MFMA.
#include <climits>
#include <cmath>
#include <new>
#include <vector>

#include "primitives.hpp"

namespace bxmi {

enum : uint8_t { TAG_ZERO = 0, TAG_ONE = 1, TAG_DATA = 2 };

constexpr int BITS_THREADS = 256;
constexpr int LONG_WORDS = 32;  // ranges spanning more words than this are filled / counted by a whole wave

__device__ __forceinline__ unsigned long long mask_from(int bit) { return ~0ull << bit; }          // bits >= bit
__device__ __forceinline__ unsigned long long mask_upto(int bit) { return ~0ull >> (63 - bit); }   // bits <= bit

// ---------------------------------------------------------------------------
// set_range (binBits.c:98-128), batched
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BITS_THREADS) void bits_set_ranges_kernel(unsigned long long *__restrict__ words,
                                                                      uint8_t *__restrict__ tags, int bin_size,
                                                                      const int32_t *__restrict__ start,
                                                                      const int32_t *__restrict__ len, int64_t n)
{
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * BITS_THREADS + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BITS_THREADS) >> 6;
    for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
        int64_t i = base + lane;
        int64_t s = 0, e = 0;
        if (i < n) {
            s = start[i];
            e = s + (int64_t)len[i];
        }
        bool active = e > s;
        int64_t w0 = s >> 6, w1 = (e - 1) >> 6;
        int64_t b0 = s / bin_size, b1 = (e - 1) / bin_size;
        bool is_long = active && (w1 - w0 > LONG_WORDS || b1 - b0 > LONG_WORDS);
        if (active) {
            unsigned long long m0 = mask_from((int)(s & 63)), m1 = mask_upto((int)((e - 1) & 63));
            if (w0 == w1) {
                atomicOr(&words[w0], m0 & m1);
            } else {
                atomicOr(&words[w0], m0);
                atomicOr(&words[w1], m1);
            }
            if (!is_long) {
                for (int64_t w = w0 + 1; w < w1; w++) words[w] = ~0ull;
                // ALL_ZERO bins become allocated; ALL_ONE bins stay as they are (binBits.c:106-113)
                for (int64_t b = b0; b <= b1; b++)
                    if (tags[b] == TAG_ZERO) tags[b] = TAG_DATA;
            }
        }
        // long ranges: the whole wave streams the interior words and the bin tags
        unsigned long long todo = __ballot(is_long);
        while (todo) {
            int src = (int)__ffsll((long long)todo) - 1;
            todo &= todo - 1;
            int64_t a = __shfl(w0, src, 64) + 1, b = __shfl(w1, src, 64);
            for (int64_t w = a + lane; w < b; w += 64) words[w] = ~0ull;
            int64_t ba = __shfl(b0, src, 64), bb = __shfl(b1, src, 64);
            for (int64_t t = ba + lane; t <= bb; t += 64)
                if (tags[t] == TAG_ZERO) tags[t] = TAG_DATA;
        }
    }
}

// ---------------------------------------------------------------------------
// count_range (binBits.c:130-178), batched
// ---------------------------------------------------------------------------
// The dense words already hold the logical bits of every bin, so the count is
// a masked popcount -- except that the reference, when the FIRST bin of the
// range is ALL_ONE, adds (piece - offset) instead of piece (binBits.c:155,161).
// Later bins start at offset 0, so the whole quirk is "- start % bin_size".
__global__ __launch_bounds__(BITS_THREADS) void bits_count_ranges_kernel(const unsigned long long *__restrict__ words,
                                                                        const uint8_t *__restrict__ tags /* may be NULL */,
                                                                        int bin_size, const int32_t *__restrict__ start,
                                                                        const int32_t *__restrict__ len, int64_t n,
                                                                        int32_t *__restrict__ out)
{
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * BITS_THREADS + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BITS_THREADS) >> 6;
    for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
        int64_t i = base + lane;
        int64_t s = 0, e = 0;
        if (i < n) {
            s = start[i];
            e = s + (int64_t)len[i];
        }
        bool active = e > s;
        int64_t w0 = s >> 6, w1 = (e - 1) >> 6;
        bool is_long = active && (w1 - w0 > LONG_WORDS);
        long long c = 0;
        unsigned long long m0 = mask_from((int)(s & 63)), m1 = mask_upto((int)((e - 1) & 63));
        if (active) {
            if (w0 == w1) {
                c = __popcll(words[w0] & m0 & m1);
            } else {
                c = __popcll(words[w0] & m0) + __popcll(words[w1] & m1);
                if (!is_long)
                    for (int64_t w = w0 + 1; w < w1; w++) c += __popcll(words[w]);
            }
            if (tags && tags[s / bin_size] == TAG_ONE) c -= s % bin_size;
        }
        unsigned long long todo = __ballot(is_long);
        while (todo) {
            int src = (int)__ffsll((long long)todo) - 1;
            todo &= todo - 1;
            int64_t a = __shfl(w0, src, 64) + 1, b = __shfl(w1, src, 64);
            long long part = 0;
            for (int64_t w = a + lane; w < b; w += 64) part += __popcll(words[w]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            if (lane == src) c += part;
        }
        if (i < n) out[i] = (int32_t)c;
    }
}


// One (short) range, one launch: the latency path behind the per-call count_range() of the drop-in class.
// Arguments travel as kernel arguments and the answer lands in host-visible memory: launch + one stream sync.
__global__ __launch_bounds__(BITS_THREADS) void bits_count_one_kernel(const unsigned long long *__restrict__ words,
                                                                     const uint8_t *__restrict__ tags, int bin_size, int64_t s,
                                                                     int64_t e, long long *__restrict__ out_host, unsigned long long seq)
{
    __shared__ long long red[BITS_THREADS / 64];
    __shared__ unsigned long long total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const int64_t w0 = s >> 6, w1 = (e - 1) >> 6;
    long long c = 0;
    for (int64_t w = w0 + threadIdx.x; w <= w1; w += BITS_THREADS) {
        unsigned long long x = words[w];
        if (w == w0) x &= mask_from((int)(s & 63));
        if (w == w1) x &= mask_upto((int)((e - 1) & 63));
        c += __popcll(x);
    }
    block_accumulate_i64(c, red, &total);
    __syncthreads();
    if (threadIdx.x == 0) {
        long long r = (long long)total;
        if (tags && tags[s / bin_size] == TAG_ONE) r -= s % bin_size;  // binBits.c:155,161
        *out_host = r;
        publish_to_host(reinterpret_cast<unsigned long long *>(out_host) + 1, seq);
    }
}

// One long range, whole grid: popcount of words [w0, w1] with edge masks.
__global__ __launch_bounds__(BITS_THREADS) void bits_popcount_span_kernel(const unsigned long long *__restrict__ words,
                                                                         int64_t w0, int64_t w1, unsigned long long m0,
                                                                         unsigned long long m1,
                                                                         unsigned long long *__restrict__ acc)
{
    __shared__ long long red[BITS_THREADS / 64];
    long long c = 0;
    // interior words [a, b): 16-byte lane loads over the even-aligned middle, scalars at the fringes
    int64_t tid = (int64_t)blockIdx.x * BITS_THREADS + threadIdx.x, nth = (int64_t)gridDim.x * BITS_THREADS;
    int64_t a = w0 + 1, b = w1 > w0 ? w1 : w0 + 1;
    int64_t a2 = (a + 1) & ~1ll;
    if (a2 > b) a2 = b;
    int64_t b2 = b & ~1ll;
    if (b2 < a2) b2 = a2;
    const ulonglong2 *v = reinterpret_cast<const ulonglong2 *>(words);
    for (int64_t p = (a2 >> 1) + tid; p < (b2 >> 1); p += nth) {
        ulonglong2 x = v[p];
        c += __popcll(x.x) + __popcll(x.y);
    }
    if (tid == 0) {
        for (int64_t w = a; w < a2; w++) c += __popcll(words[w]);
        for (int64_t w = b2; w < b; w++) c += __popcll(words[w]);
        if (w0 == w1)
            c += __popcll(words[w0] & m0 & m1);
        else
            c += __popcll(words[w0] & m0) + __popcll(words[w1] & m1);
    }
    block_accumulate_i64(c, red, acc);
}

// ---------------------------------------------------------------------------
// iand / ior / invert (binBits.c:230-317) on the dense words
// ---------------------------------------------------------------------------
// OP: 0 = and, 1 = or, 2 = xor.  COUNT: also accumulate popcount of the result inside [0, size).
// One launch per call of the drop-in classes (bed_intersect_basewise.py:25-28 does one iand per chromosome), so the
// launch itself is most of the cost: the per-bin tags are updated by the FIRST workgroup of the same launch instead of
// a second kernel, a lane keeps four 16-byte loads per operand in flight, and the counting variants run at most one
// workgroup per CU (their single atomic per workgroup serialises on one counter, ~12 ns each).
constexpr int BITS_UNROLL = 4;

__device__ __forceinline__ long long pair_popcount(ulonglong2 x, int64_t w, int64_t full_words, unsigned long long tail_mask)
{
    if (w + 1 < full_words) return __popcll(x.x) + __popcll(x.y);
    long long c = w < full_words ? __popcll(x.x) : (w == full_words ? __popcll(x.x & tail_mask) : 0);
    return c + ((w + 1) < full_words ? __popcll(x.y) : ((w + 1) == full_words ? __popcll(x.y & tail_mask) : 0));
}

__device__ __forceinline__ void tags_binary(int op, uint8_t *__restrict__ ta, const uint8_t *__restrict__ tb, int64_t nbins)
{
    for (int64_t i = threadIdx.x; i < nbins; i += blockDim.x) {
        const uint8_t x = ta[i], y = tb[i];
        if (op == 0) {                                 // binBitsAnd, binBits.c:237-256
            if (x == TAG_ZERO) continue;
            if (y == TAG_ZERO) ta[i] = TAG_ZERO;
            else if (y == TAG_ONE) continue;
            else if (x == TAG_ONE) ta[i] = TAG_DATA;   // clone of other
        } else {                                       // binBitsOr, binBits.c:271-290
            if (x == TAG_ONE) continue;
            if (y == TAG_ONE) ta[i] = TAG_ONE;
            else if (y == TAG_ZERO) continue;
            else if (x == TAG_ZERO) ta[i] = TAG_DATA;  // clone of other
        }
    }
}

// (these two run with 256 threads per workgroup when they only stream, with BITS_COUNT_THREADS when they also count:
// a counting launch ends in one atomic per workgroup on the caller's int64 -- ~85 ns each on the same address, and no
// cheaper spread over 64 partial sums with per-slot tickets: the fences cost more than they save (measured: popcount
// 0.17 -> 0.23 ms at the same grid, 0.52 ms at 8 workgroups per CU) -- which keeps the grid at one workgroup per CU, and 256
// threads per CU cannot keep enough loads in flight: 2 TB/s on a chromosome-sized set, 2.2-2.4 with 1024)
constexpr int BITS_COUNT_THREADS = 1024;
constexpr int64_t BITS_WIDE_BELOW = (int64_t)1 << 21;  // 16-byte pairs (32 MiB of words): larger sets stream well enough at 256 threads (5.4 TB/s at 64 MiB)

template <int OP, bool COUNT>
__global__ __launch_bounds__(BITS_COUNT_THREADS) void bits_binary_kernel(unsigned long long *__restrict__ a,
                                                                  const unsigned long long *__restrict__ b,
                                                                  int64_t npairs /* 16-byte pairs */, int64_t size_bits,
                                                                  unsigned long long *__restrict__ acc, uint8_t *__restrict__ ta,
                                                                  const uint8_t *__restrict__ tb, int64_t nbins /* 0: no tags (flat set) */)
{
    __shared__ long long red[BITS_COUNT_THREADS / 64];
    ulonglong2 *va = reinterpret_cast<ulonglong2 *>(a);
    const ulonglong2 *vb = reinterpret_cast<const ulonglong2 *>(b);
    const int64_t full_words = size_bits >> 6;  // words entirely inside [0,size)
    const unsigned long long tail_mask = (size_bits & 63) ? ~(~0ull << (size_bits & 63)) : 0ull;
    long long c = 0;
    // the per-bin tags (an array of their own): by the FIRST workgroup, before its share of the words, so that the three
    // dependent round trips hide behind everybody else's streaming instead of trailing the launch
    if (nbins > 0 && blockIdx.x == 0) tags_binary(OP, ta, tb, nbins);
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; p + (BITS_UNROLL - 1) * nth < npairs; p += BITS_UNROLL * nth) {
        ulonglong2 x[BITS_UNROLL], y[BITS_UNROLL];
#pragma unroll
        for (int k = 0; k < BITS_UNROLL; k++) x[k] = va[p + k * nth], y[k] = vb[p + k * nth];
#pragma unroll
        for (int k = 0; k < BITS_UNROLL; k++) {
            if (OP == 0) x[k].x &= y[k].x, x[k].y &= y[k].y;
            else if (OP == 1) x[k].x |= y[k].x, x[k].y |= y[k].y;
            else x[k].x ^= y[k].x, x[k].y ^= y[k].y;
            va[p + k * nth] = x[k];
            if (COUNT) c += pair_popcount(x[k], (p + k * nth) * 2, full_words, tail_mask);
        }
    }
    for (; p < npairs; p += nth) {
        ulonglong2 x = va[p], y = vb[p];
        if (OP == 0) x.x &= y.x, x.y &= y.y;
        else if (OP == 1) x.x |= y.x, x.y |= y.y;
        else x.x ^= y.x, x.y ^= y.y;
        va[p] = x;
        if (COUNT) c += pair_popcount(x, p * 2, full_words, tail_mask);
    }
    if (COUNT) block_accumulate_i64(c, red, acc);
}

__global__ __launch_bounds__(BITS_COUNT_THREADS) void bits_popcount_kernel(const unsigned long long *__restrict__ a, int64_t npairs,
                                                                          int64_t size_bits, unsigned long long *__restrict__ acc)
{
    __shared__ long long red[BITS_COUNT_THREADS / 64];
    const ulonglong2 *va = reinterpret_cast<const ulonglong2 *>(a);
    const int64_t full_words = size_bits >> 6;
    const unsigned long long tail_mask = (size_bits & 63) ? ~(~0ull << (size_bits & 63)) : 0ull;
    long long c = 0;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; p + (BITS_UNROLL - 1) * nth < npairs; p += BITS_UNROLL * nth) {
        ulonglong2 x[BITS_UNROLL];
#pragma unroll
        for (int k = 0; k < BITS_UNROLL; k++) x[k] = va[p + k * nth];
#pragma unroll
        for (int k = 0; k < BITS_UNROLL; k++) c += pair_popcount(x[k], (p + k * nth) * 2, full_words, tail_mask);
    }
    for (; p < npairs; p += nth) c += pair_popcount(va[p], p * 2, full_words, tail_mask);
    block_accumulate_i64(c, red, acc);
}

// invert: flip every bit of [0, total_bits); words past it stay zero.
__global__ __launch_bounds__(BITS_THREADS) void bits_not_kernel(unsigned long long *__restrict__ a, int64_t nwords,
                                                               int64_t total_bits)
{
    int64_t nth = (int64_t)gridDim.x * BITS_THREADS;
    const int64_t full = total_bits >> 6;
    const unsigned long long tail_mask = (total_bits & 63) ? ~(~0ull << (total_bits & 63)) : 0ull;
    for (int64_t w = (int64_t)blockIdx.x * BITS_THREADS + threadIdx.x; w < nwords; w += nth) {
        unsigned long long x = ~a[w];
        a[w] = w < full ? x : (w == full ? (x & tail_mask) : 0ull);
    }
}

// Per-bin state tables of binBitsAnd / binBitsOr / binBitsNot.
__global__ void tags_not_kernel(uint8_t *__restrict__ a, int64_t nbins)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbins; i += (int64_t)gridDim.x * blockDim.x) {
        uint8_t x = a[i];
        a[i] = x == TAG_ONE ? TAG_ZERO : (x == TAG_ZERO ? TAG_ONE : TAG_DATA);  // binBits.c:304-315
    }
}

// set / clear of one position (binBits.c:67-96)
__global__ void bits_point_kernel(unsigned long long *words, uint8_t *tags, int64_t pos, int64_t bin, int set)
{
    if (threadIdx.x | blockIdx.x) return;
    uint8_t t = tags[bin];
    unsigned long long m = 1ull << (pos & 63);
    if (set) {
        if (t == TAG_ONE) return;
        if (t == TAG_ZERO) tags[bin] = TAG_DATA;
        words[pos >> 6] |= m;
    } else {
        if (t == TAG_ZERO) return;
        if (t == TAG_ONE) tags[bin] = TAG_DATA;  // the dense words already hold the ones
        words[pos >> 6] &= ~m;
    }
}

// ---------------------------------------------------------------------------
// next_set / next_clear (binBits.c:180-228): first bit == val in [start, limit)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BITS_THREADS) void bits_find_kernel(const unsigned long long *__restrict__ words,
                                                                int64_t start, int64_t limit, int val, int64_t w_begin,
                                                                int64_t w_end /* exclusive */,
                                                                unsigned long long *__restrict__ result)
{
    int64_t nth = (int64_t)gridDim.x * BITS_THREADS;
    const unsigned long long flip = val ? 0ull : ~0ull;
    for (int64_t w = w_begin + (int64_t)blockIdx.x * BITS_THREADS + threadIdx.x; w < w_end; w += nth) {
        if ((unsigned long long)(w << 6) >= __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        unsigned long long x = words[w] ^ flip;
        if ((w << 6) < start) x &= mask_from((int)(start & 63));
        if ((w << 6) + 64 > limit) x &= ~(~0ull << (limit - (w << 6)));
        if (x) {
            atomicMin(result, (unsigned long long)((w << 6) + __ffsll((long long)x) - 1));
            break;
        }
    }
}

// ---------------------------------------------------------------------------
// maximal runs of set bits in [from, size)
// ---------------------------------------------------------------------------
constexpr int RUN_WPT = 4;                              // words per thread
constexpr int RUN_TILE = BITS_THREADS * RUN_WPT;        // words per workgroup

__device__ __forceinline__ unsigned long long run_word(const unsigned long long *__restrict__ words, int64_t w,
                                                       int64_t from, int64_t size, int64_t w_first, int64_t w_last)
{
    if (w < w_first || w > w_last) return 0ull;
    unsigned long long x = words[w];
    if ((w << 6) < from) x &= mask_from((int)(from & 63));
    if ((w << 6) + 64 > size) x &= ~(~0ull << (size - (w << 6)));
    return x;
}

// starts: set bits whose predecessor is clear; lasts: set bits whose successor is clear
__device__ __forceinline__ void run_edges(const unsigned long long *__restrict__ words, int64_t w, int64_t from,
                                          int64_t size, int64_t w_first, int64_t w_last, unsigned long long &starts,
                                          unsigned long long &lasts)
{
    unsigned long long x = run_word(words, w, from, size, w_first, w_last);
    if (!x) {
        starts = lasts = 0;
        return;
    }
    unsigned long long prev = run_word(words, w - 1, from, size, w_first, w_last) >> 63;
    unsigned long long next = run_word(words, w + 1, from, size, w_first, w_last) & 1ull;
    starts = x & ~((x << 1) | prev);
    lasts = x & ~((x >> 1) | (next << 63));
}

__global__ __launch_bounds__(BITS_THREADS) void bits_runs_count_kernel(const unsigned long long *__restrict__ words,
                                                                      int64_t from, int64_t size, int64_t w_first,
                                                                      int64_t w_last, int32_t *__restrict__ tile_starts,
                                                                      int32_t *__restrict__ tile_lasts)
{
    __shared__ int lds[8];
    int64_t wb = w_first + (int64_t)blockIdx.x * RUN_TILE + (int64_t)threadIdx.x * RUN_WPT;
    int cs = 0, cl = 0;
#pragma unroll
    for (int j = 0; j < RUN_WPT; j++) {
        unsigned long long s, l;
        run_edges(words, wb + j, from, size, w_first, w_last, s, l);
        cs += __popcll(s);
        cl += __popcll(l);
    }
    int ts, tl;
    (void)block_exclusive_scan(cs, OpSum(), 0, lds, &ts);
    (void)block_exclusive_scan(cl, OpSum(), 0, lds, &tl);
    if (threadIdx.x == 0) {
        tile_starts[blockIdx.x] = ts;
        tile_lasts[blockIdx.x] = tl;
    }
}

__global__ __launch_bounds__(BITS_THREADS) void bits_runs_fill_kernel(const unsigned long long *__restrict__ words,
                                                                     int64_t from, int64_t size, int64_t w_first,
                                                                     int64_t w_last, const int32_t *__restrict__ tile_starts,
                                                                     const int32_t *__restrict__ tile_lasts,
                                                                     int32_t *__restrict__ run_start,
                                                                     int32_t *__restrict__ run_end)
{
    __shared__ int lds[8];
    int64_t wb = w_first + (int64_t)blockIdx.x * RUN_TILE + (int64_t)threadIdx.x * RUN_WPT;
    unsigned long long s[RUN_WPT], l[RUN_WPT];
    int cs = 0, cl = 0;
#pragma unroll
    for (int j = 0; j < RUN_WPT; j++) {
        run_edges(words, wb + j, from, size, w_first, w_last, s[j], l[j]);
        cs += __popcll(s[j]);
        cl += __popcll(l[j]);
    }
    int ts, tl;
    int ps = tile_starts[blockIdx.x] + block_exclusive_scan(cs, OpSum(), 0, lds, &ts);
    int pl = tile_lasts[blockIdx.x] + block_exclusive_scan(cl, OpSum(), 0, lds, &tl);
#pragma unroll
    for (int j = 0; j < RUN_WPT; j++) {
        int64_t bit0 = (wb + j) << 6;
        unsigned long long m = s[j];
        while (m) {
            run_start[ps++] = (int32_t)(bit0 + __ffsll((long long)m) - 1);
            m &= m - 1;
        }
        m = l[j];
        while (m) {
            run_end[pl++] = (int32_t)(bit0 + __ffsll((long long)m));  // exclusive end
            m &= m - 1;
        }
    }
}


// ---------------------------------------------------------------------------
// genome-scale batches: AND / OR / popcount over MANY bitsets in one launch
// ---------------------------------------------------------------------------
// bed_intersect_basewise.py:25-28 and bed_coverage.py:27-29 loop over the
// chromosomes; 24 small launches (16 MB each) leave the GPU mostly idle and pay
// one contended atomic per workgroup.  A group keeps a device table of its
// members so the whole genome is one launch: workgroups walk 64 KiB chunks of a
// flattened chunk space, and per-member counts land on per-member counters.
constexpr int MB_CHUNK_PAIRS = 4096;  // 16-byte pairs per chunk = 64 KiB

struct GroupSeg {
    unsigned long long *words;
    uint8_t *tags;
    int64_t npairs;
    int64_t size_bits;
    int64_t chunk_first;  // first chunk of this member in the flattened chunk space
    int64_t tag_first;    // first tag of this member in the flattened tag space
    int64_t ntags;
};

__device__ __forceinline__ int group_find(const GroupSeg *__restrict__ segs, int nseg, int64_t chunk)
{
    int lo = 0, hi = nseg;  // last member whose first chunk is <= chunk
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (segs[mid].chunk_first <= chunk)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
// 16-byte non-temporal store: the result of a streaming AND/OR is not read again by this kernel
__device__ __forceinline__ void store_pair_nt(ulonglong2 *dst, ulonglong2 v)
{
    u64x2 t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<u64x2 *>(dst));
}

// OP: 0 = and, 1 = or, 3 = read-only popcount.  COUNT: accumulate popcount of the result inside [0,size).
template <int OP, bool COUNT>
__global__ __launch_bounds__(BITS_THREADS) void bits_group_kernel(const GroupSeg *__restrict__ sa, const GroupSeg *__restrict__ sb,
                                                                 int nseg, int64_t total_chunks,
                                                                 unsigned long long *__restrict__ counts)
{
    __shared__ long long red[BITS_THREADS / 64];
    const int64_t per = (total_chunks + gridDim.x - 1) / gridDim.x;
    int64_t c = (int64_t)blockIdx.x * per;
    const int64_t c_end = c + per < total_chunks ? c + per : total_chunks;
    if (c >= c_end) return;
    int seg = group_find(sa, nseg, c);
    long long acc = 0;
    for (; c < c_end; c++) {
        while (seg + 1 < nseg && sa[seg + 1].chunk_first <= c) {  // next member: flush this one's count
            if (COUNT) {
                block_accumulate_i64(acc, red, counts + seg);
                __syncthreads();
                acc = 0;
            }
            seg++;
        }
        const GroupSeg A = sa[seg];
        // (the members' word arrays come out of a table in memory: as_global, common.hpp -- global_load / global_store with counted
        // waits instead of FLAT instructions)
        u64x2 BX_GLOBAL *va = reinterpret_cast<u64x2 BX_GLOBAL *>(as_global(A.words));
        const u64x2 BX_GLOBAL *vb = OP == 3 ? nullptr : reinterpret_cast<const u64x2 BX_GLOBAL *>(as_global(sb[seg].words));
        const int64_t p0 = (c - A.chunk_first) * MB_CHUNK_PAIRS;
        const int64_t p1 = p0 + MB_CHUNK_PAIRS < A.npairs ? p0 + MB_CHUNK_PAIRS : A.npairs;
        const int64_t full_words = A.size_bits >> 6;
        const unsigned long long tail_mask = (A.size_bits & 63) ? ~(~0ull << (A.size_bits & 63)) : 0ull;
        // 4 x 16-byte loads per operand in flight per lane before the first store (a chunk is 16 pairs per lane)
        constexpr int U = 4;
        for (int64_t pb = p0 + threadIdx.x; pb < p1; pb += (int64_t)BITS_THREADS * U) {
            u64x2 x[U], y[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t p = pb + (int64_t)u * BITS_THREADS;
                x[u] = p < p1 ? va[p] : u64x2{0, 0};
                if (OP != 3) y[u] = p < p1 ? vb[p] : u64x2{0, 0};
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t p = pb + (int64_t)u * BITS_THREADS;
                if (p >= p1) break;
                if (OP != 3) {
                    if (OP == 0) {
                        x[u].x &= y[u].x;
                        x[u].y &= y[u].y;
                    } else {
                        x[u].x |= y[u].x;
                        x[u].y |= y[u].y;
                    }
                    __builtin_nontemporal_store(x[u], va + p);
                }
                if (COUNT) {
                    int64_t w = p * 2;
                    if (w + 1 < full_words) {
                        acc += __popcll(x[u].x) + __popcll(x[u].y);
                    } else {
                        acc += w < full_words ? __popcll(x[u].x) : (w == full_words ? __popcll(x[u].x & tail_mask) : 0);
                        acc += (w + 1) < full_words ? __popcll(x[u].y) : ((w + 1) == full_words ? __popcll(x[u].y & tail_mask) : 0);
                    }
                }
            }
        }
    }
    if (COUNT) block_accumulate_i64(acc, red, counts + seg);
}

// Per-bin state tables of binBitsAnd / binBitsOr over the flattened tag space of a group.
template <int OP>
__global__ void tags_group_kernel(const GroupSeg *__restrict__ sa, const GroupSeg *__restrict__ sb, int nseg, int64_t total_tags)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_tags; i += (int64_t)gridDim.x * blockDim.x) {
        int lo = 0, hi = nseg;
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (sa[mid].tag_first <= i)
                lo = mid;
            else
                hi = mid;
        }
        int64_t k = i - sa[lo].tag_first;
        uint8_t *a = sa[lo].tags;
        const uint8_t *b = sb[lo].tags;
        uint8_t x = a[k], y = b[k];
        if (OP == 0) {  // binBits.c:237-256
            if (x == TAG_ZERO || y == TAG_ONE) continue;
            if (y == TAG_ZERO) a[k] = TAG_ZERO;
            else if (x == TAG_ONE) a[k] = TAG_DATA;
        } else {  // binBits.c:271-290
            if (x == TAG_ONE || y == TAG_ZERO) continue;
            if (y == TAG_ONE) a[k] = TAG_ONE;
            else if (x == TAG_ZERO) a[k] = TAG_DATA;
        }
    }
}

static int64_t g_opt_bits_grid = 0;
int64_t bits_get_grid() { return g_opt_bits_grid; }

int bits_set_option(const char *key, int64_t value)
{
    if (!strcmp(key, "bits.grid")) {
        g_opt_bits_grid = value;
        return 1;
    }
    return 0;
}

static int bits_grid(int64_t items, int per_block)
{
    if (g_opt_bits_grid > 0) {
        int64_t need = div_up(items, per_block);
        return (int)(need < g_opt_bits_grid ? (need < 1 ? 1 : need) : g_opt_bits_grid);
    }
    return stream_grid(items, per_block);
}

}  // namespace bxmi

using namespace bxmi;

struct bxmi_bits {
    int32_t size = 0, bin_size = 0, nbins = 0;
    int64_t total_bits = 0;  // bins cover [0, nbins*bin_size) >= size
    int64_t nwords = 0;      // even, so the words can be read as 16-byte pairs
    int64_t cap_words = 0;   // words actually allocated: [0, cap_words) hold bits, everything beyond is zero (see bits_need)
    bool bounded_call = false;  // a host entry point has already sized / clamped for the `_dev` form it is about to call
    DevBuf words, tags;
    bool maybe_one = false;  // some bin may be ALL_ONE (only after invert / ior with such a set)
    bool flat = false;       // plain BitSet (bitset.pyx:107-173): no bins, no tri-state quirks
    DevBuf q_a, q_b, q_out, acc, tiles_s, tiles_l, run_s, run_e, scan_tmp;
    long long *one_buf = nullptr;  // host-visible result of the one-range latency path: [0] the count, [1] completion word
    unsigned long long one_seq = 0;
    hipStream_t stream = nullptr;
};

static int bits_stream(bxmi_bits *h)
{
    if (!h->stream) BXMI_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    return BXMI_OK;
}

// The reference allocates a bin (64 KiB at the default granularity) the first time a bit in it is set
// (binBits.c:98-128), which is what lets bitset_builders.py create one MAX-sized set per sequence name of a
// scaffold-level assembly.  Here the words are one dense array, allocated LAZILY up to the highest bit any operation has
// needed so far: a fresh set owns no words at all, `set_range` grows the array to the end of the batch (rounded up to whole
// bins, at least doubling), reads treat everything past the allocated part as the zeros it logically is.  Operations
// that cannot bound what they touch (the stream-ordered `_dev` forms, groups, invert, the raw word view) ask for the whole array.
constexpr int64_t BITS_ALL = -1;
static int64_t cap_bits(const bxmi_bits *h) { return h->cap_words * 64; }

static int bits_need(bxmi_bits *h, int64_t upto_bits /* exclusive; BITS_ALL = the whole set */)
{
    int64_t want = h->nwords;
    if (upto_bits != BITS_ALL) {
        if (upto_bits <= cap_bits(h)) return BXMI_OK;
        const int64_t bins = div_up(upto_bits, h->bin_size);
        want = (div_up(bins * (int64_t)h->bin_size, 64) + 1) & ~1ll;
        if (want < 2 * h->cap_words) want = 2 * h->cap_words;
        if (want > h->nwords) want = h->nwords;
    }
    if (want <= h->cap_words) return BXMI_OK;
    void *np = nullptr;
    const size_t bytes = (size_t)want * 8 + 64;
    hipError_t e = hipMalloc(&np, bytes);
    if (e != hipSuccess)
        return fail(BXMI_ENOMEM, "BinnedBitSet of %d bits: no room for %zu MiB of words on the device (%s); every set holds dense words up "
                    "to its highest set bit -- size the sets with `lens` instead of the 512 Mi default", h->size, bytes >> 20, hipGetErrorString(e));
    // The old words may still be written by stream-ordered work on a caller's non-blocking stream (a `_dev` call queued
    // before this one): growing is rare, so the whole device drains before the words move.
    if (h->cap_words) e = hipDeviceSynchronize();
    if (e == hipSuccess && h->cap_words) e = hipMemcpy(np, h->words.p, (size_t)h->cap_words * 8, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipMemset(static_cast<char *>(np) + (size_t)h->cap_words * 8, 0, bytes - (size_t)h->cap_words * 8);
    if (e != hipSuccess) {
        (void)hipFree(np);
        return fail(BXMI_EHIP, "bits_need: %s", hipGetErrorString(e));
    }
    h->words.release();
    h->words.p = np;
    h->words.cap = bytes;
    h->cap_words = want;
    return BXMI_OK;
}

extern "C" int bxmi_bits_create(int64_t size, int64_t granularity, bxmi_bits_t **out)
{
    if (!out) return fail(BXMI_EINVAL, "bxmi_bits_create: out is NULL");
    if (size > 2147483647ll) return fail(BXMI_EINVAL, "bxmi_bits_create: size %lld exceeds 2^31-1", (long long)size);
    if (size < 1 || granularity < 0 || granularity > 2147483647ll)
        return fail(BXMI_EINVAL, "bxmi_bits_create: size must be >= 1 and granularity >= 1 (0 = flat BitSet)");
    bxmi_bits *h = new (std::nothrow) bxmi_bits();
    if (!h) return fail(BXMI_ENOMEM, "bxmi_bits_create: host allocation failed");
    h->size = (int32_t)size;
    if (granularity == 0) {  // flat BitSet: one pseudo-bin, tags never consulted
        h->flat = true;
        h->bin_size = (int32_t)size;
        h->nbins = 1;
    } else {
        // binBits.c:13-14 -- both quotients are formed in single precision
        h->bin_size = (int32_t)std::ceil((double)((float)(int32_t)size / (float)(int32_t)granularity));
        h->nbins = (int32_t)std::ceil((double)((float)(int32_t)size / (float)h->bin_size));
    }
    h->total_bits = (int64_t)h->nbins * h->bin_size;
    // float32 rounding can leave nbins*bin_size < size (e.g. 16777217/1024): the reference then
    // indexes past its bin table; we keep the tail addressable instead of reproducing that UB.
    int64_t tag_count = h->nbins;
    if (h->total_bits < size) {
        tag_count = div_up(size, h->bin_size);
        h->total_bits = tag_count * h->bin_size;
    }
    h->nwords = (div_up(h->total_bits, 64) + 1) & ~1ll;
    int rc = h->tags.reserve((size_t)tag_count + 64);  // the words come with the first bit (bits_need)
    if (rc == BXMI_OK && hipMemset(h->tags.p, 0, h->tags.cap) != hipSuccess) rc = fail(BXMI_EHIP, "hipMemset failed");
    if (rc != BXMI_OK) {
        delete h;
        return rc;
    }
    *out = h;
    return BXMI_OK;
}

extern "C" int bxmi_bits_destroy(bxmi_bits_t *h)
{
    if (!h) return BXMI_OK;
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->one_buf) (void)hipHostFree(h->one_buf);
    delete h;
    return BXMI_OK;
}

extern "C" int bxmi_bits_info(const bxmi_bits_t *h, int32_t *size, int32_t *bin_size, int32_t *nbins)
{
    if (!h) return fail(BXMI_EINVAL, "bxmi_bits_info: NULL handle");
    if (size) *size = h->size;
    if (bin_size) *bin_size = h->bin_size;
    if (nbins) *nbins = h->nbins;
    return BXMI_OK;
}

extern "C" int bxmi_bits_words_dev(bxmi_bits_t *h, uint64_t **words_dev, int64_t *nwords)
{
    if (!h) return fail(BXMI_EINVAL, "bxmi_bits_words_dev: NULL handle");
    BXMI_TRY(bits_need(h, BITS_ALL));
    if (words_dev) *words_dev = h->words.as<uint64_t>();
    if (nwords) *nwords = h->nwords;
    return BXMI_OK;
}

extern "C" int bxmi_bits_bin_states(bxmi_bits_t *h, uint8_t *out)
{
    if (!h || !out) return fail(BXMI_EINVAL, "bxmi_bits_bin_states: bad arguments");
    BXMI_HIP(hipMemcpy(out, h->tags.p, (size_t)h->nbins, hipMemcpyDeviceToHost));
    return BXMI_OK;
}

static int check_pos(const bxmi_bits *h, int64_t pos, const char *who)
{
    if (!h) return fail(BXMI_EINVAL, "%s: NULL handle", who);
    if (pos < 0 || pos >= h->size) return fail(BXMI_EINVAL, "%s: position %lld outside [0,%d)", who, (long long)pos, h->size);
    return BXMI_OK;
}

extern "C" int bxmi_bits_get(bxmi_bits_t *h, int32_t pos, int *bit)
{
    BXMI_TRY(check_pos(h, pos, "bxmi_bits_get"));
    if (!bit) return fail(BXMI_EINVAL, "bxmi_bits_get: bit is NULL");
    unsigned long long w = 0;
    if (pos < cap_bits(h)) BXMI_HIP(hipMemcpy(&w, h->words.as<unsigned long long>() + (pos >> 6), 8, hipMemcpyDeviceToHost));
    *bit = (int)((w >> (pos & 63)) & 1ull);
    return BXMI_OK;
}

static int bits_point(bxmi_bits *h, int32_t pos, int set)
{
    BXMI_TRY(bits_stream(h));
    BXMI_TRY(bits_need(h, h->maybe_one ? BITS_ALL : (int64_t)pos + 1));
    hipLaunchKernelGGL(bits_point_kernel, dim3(1), dim3(64), 0, h->stream, h->words.as<unsigned long long>(), h->tags.as<uint8_t>(),
                       (int64_t)pos, (int64_t)(pos / h->bin_size), set);
    BXMI_LAUNCH_CHECK();
    BXMI_HIP(hipStreamSynchronize(h->stream));
    return BXMI_OK;
}

extern "C" int bxmi_bits_set(bxmi_bits_t *h, int32_t pos)
{
    BXMI_TRY(check_pos(h, pos, "bxmi_bits_set"));
    return bits_point(h, pos, 1);
}

extern "C" int bxmi_bits_clear(bxmi_bits_t *h, int32_t pos)
{
    BXMI_TRY(check_pos(h, pos, "bxmi_bits_clear"));
    return bits_point(h, pos, 0);
}

extern "C" int bxmi_bits_set_ranges_dev(bxmi_bits_t *h, const int32_t *start, const int32_t *len, int64_t n, void *stream)
{
    if (!h || n < 0 || (n > 0 && (!start || !len))) return fail(BXMI_EINVAL, "bxmi_bits_set_ranges_dev: bad arguments");
    if (n == 0) return BXMI_OK;
    if (!h->bounded_call) BXMI_TRY(bits_need(h, BITS_ALL));  // device-resident ranges: their extent is not known here
    hipLaunchKernelGGL(bits_set_ranges_kernel, dim3(bits_grid(n, BITS_THREADS)), dim3(BITS_THREADS), 0, as_stream(stream),
                       h->words.as<unsigned long long>(), h->tags.as<uint8_t>(), h->bin_size, start, len, n);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// Host-side validation of a batch (the Python wrapper raises the exact
// bitset.pyx exception first; this is the C ABI's own guard against writes
// outside the array).
static int validate_ranges(const bxmi_bits *h, const int32_t *start, const int32_t *len, int64_t n, const char *who)
{
    for (int64_t i = 0; i < n; i++) {
        int64_t s = start[i], c = len[i];
        if (s < 0 || c < 0 || s + c > h->size || (s >= h->size))
            return fail(BXMI_EINVAL, "%s: range %lld (start=%lld, len=%lld) outside [0,%d]", who, (long long)i, (long long)s, (long long)c,
                        h->size);
    }
    return BXMI_OK;
}

static int upload_ranges(bxmi_bits *h, const int32_t *start, const int32_t *len, int64_t n)
{
    BXMI_TRY(bits_stream(h));
    BXMI_TRY(h->q_a.reserve((size_t)(n + 4) * 4));
    BXMI_TRY(h->q_b.reserve((size_t)(n + 4) * 4));
    BXMI_HIP(hipMemcpyAsync(h->q_a.p, start, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    BXMI_HIP(hipMemcpyAsync(h->q_b.p, len, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    return BXMI_OK;
}

extern "C" int bxmi_bits_set_ranges(bxmi_bits_t *h, const int32_t *start, const int32_t *len, int64_t n)
{
    if (!h || n < 0 || (n > 0 && (!start || !len))) return fail(BXMI_EINVAL, "bxmi_bits_set_ranges: bad arguments");
    if (n == 0) return BXMI_OK;
    BXMI_TRY(validate_ranges(h, start, len, n, "bxmi_bits_set_ranges"));
    int64_t top = 0;
    for (int64_t i = 0; i < n; i++)
        if (len[i] > 0 && (int64_t)start[i] + len[i] > top) top = (int64_t)start[i] + len[i];
    if (top == 0) return BXMI_OK;  // nothing but empty ranges: no bin is touched (binBits.c:101)
    BXMI_TRY(bits_need(h, h->maybe_one ? BITS_ALL : top));
    BXMI_TRY(upload_ranges(h, start, len, n));
    h->bounded_call = true;
    const int rc = bxmi_bits_set_ranges_dev(h, h->q_a.as<int32_t>(), h->q_b.as<int32_t>(), n, h->stream);
    h->bounded_call = false;
    BXMI_TRY(rc);
    BXMI_HIP(hipStreamSynchronize(h->stream));
    return BXMI_OK;
}

extern "C" int bxmi_bits_count_ranges_dev(bxmi_bits_t *h, const int32_t *start, const int32_t *len, int64_t n, int32_t *out,
                                          void *stream)
{
    if (!h || n < 0 || (n > 0 && (!start || !len || !out))) return fail(BXMI_EINVAL, "bxmi_bits_count_ranges_dev: bad arguments");
    if (n == 0) return BXMI_OK;
    if (!h->bounded_call) BXMI_TRY(bits_need(h, BITS_ALL));
    hipLaunchKernelGGL(bits_count_ranges_kernel, dim3(bits_grid(n, BITS_THREADS)), dim3(BITS_THREADS), 0, as_stream(stream),
                       h->words.as<unsigned long long>(), h->maybe_one ? h->tags.as<uint8_t>() : nullptr, h->bin_size, start, len, n, out);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

extern "C" int bxmi_bits_count_ranges(bxmi_bits_t *h, const int32_t *start, const int32_t *len, int64_t n, int32_t *out)
{
    if (!h || n < 0 || (n > 0 && (!start || !len || !out))) return fail(BXMI_EINVAL, "bxmi_bits_count_ranges: bad arguments");
    if (n == 0) return BXMI_OK;
    BXMI_TRY(validate_ranges(h, start, len, n, "bxmi_bits_count_ranges"));
    if (h->cap_words < h->nwords) {
        // bits past the allocated words are zero (and no bin there is ALL_ONE: invert allocates everything): count what lies
        // inside, on clamped copies of the ranges
        const int64_t cap = cap_bits(h);
        if (cap == 0) {
            memset(out, 0, (size_t)n * 4);
            return BXMI_OK;
        }
        std::vector<int32_t> cs((size_t)n), cl((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            const int64_t s0 = start[i], e0 = s0 + len[i];
            const bool in = s0 < cap;
            cs[(size_t)i] = in ? (int32_t)s0 : 0;
            cl[(size_t)i] = in ? (int32_t)((e0 < cap ? e0 : cap) - s0) : 0;
        }
        BXMI_TRY(upload_ranges(h, cs.data(), cl.data(), n));
        BXMI_HIP(hipStreamSynchronize(h->stream));  // the staging vectors die with this scope
    } else {
        BXMI_TRY(upload_ranges(h, start, len, n));
    }
    BXMI_TRY(h->q_out.reserve((size_t)(n + 4) * 4));
    h->bounded_call = true;
    const int rc = bxmi_bits_count_ranges_dev(h, h->q_a.as<int32_t>(), h->q_b.as<int32_t>(), n, h->q_out.as<int32_t>(), h->stream);
    h->bounded_call = false;
    BXMI_TRY(rc);
    BXMI_HIP(hipMemcpyAsync(out, h->q_out.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    BXMI_HIP(hipStreamSynchronize(h->stream));
    return BXMI_OK;
}

extern "C" int bxmi_bits_count_range(bxmi_bits_t *h, int32_t start, int32_t len, int32_t *out)
{
    if (!h || !out) return fail(BXMI_EINVAL, "bxmi_bits_count_range: bad arguments");
    BXMI_TRY(validate_ranges(h, &start, &len, 1, "bxmi_bits_count_range"));
    *out = 0;
    if (len == 0) return BXMI_OK;
    if (h->cap_words < h->nwords) {  // past the allocated words everything is zero
        const int64_t cap = cap_bits(h);
        if (start >= cap) return BXMI_OK;
        if ((int64_t)start + len > cap) len = (int32_t)(cap - start);
    }
    if (len < (1 << 20)) {  // up to 16 Ki words: one workgroup, result straight into host-visible memory
        BXMI_TRY(bits_stream(h));
        if (!h->one_buf) {
            BXMI_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->one_buf), 64, hipHostMallocDefault));
            memset(h->one_buf, 0, 64);
        }
        const unsigned long long seq = ++h->one_seq;
        hipLaunchKernelGGL(bits_count_one_kernel, dim3(1), dim3(BITS_THREADS), 0, h->stream, h->words.as<unsigned long long>(),
                           h->maybe_one ? h->tags.as<uint8_t>() : nullptr, h->bin_size, (int64_t)start, (int64_t)start + len, h->one_buf, seq);
        BXMI_LAUNCH_CHECK();
        BXMI_TRY(wait_for_host_flag(reinterpret_cast<const unsigned long long *>(h->one_buf) + 1, seq, h->stream));
        *out = (int32_t)*reinterpret_cast<volatile long long *>(h->one_buf);
        return BXMI_OK;
    }
    // chromosome-scale range: every CU takes part
    BXMI_TRY(bits_stream(h));
    BXMI_TRY(h->acc.reserve(64));
    BXMI_HIP(hipMemsetAsync(h->acc.p, 0, 8, h->stream));
    int64_t s = start, e = s + len;
    int64_t w0 = s >> 6, w1 = (e - 1) >> 6;
    unsigned long long m0 = ~0ull << (s & 63), m1 = ~0ull >> (63 - ((e - 1) & 63));
    hipLaunchKernelGGL(bits_popcount_span_kernel, dim3(bits_grid((w1 - w0 + 1) / 2 + 1, BITS_THREADS * 4)), dim3(BITS_THREADS), 0, h->stream,
                       h->words.as<unsigned long long>(), w0, w1, m0, m1, h->acc.as<unsigned long long>());
    BXMI_LAUNCH_CHECK();
    long long c = 0;
    BXMI_HIP(hipMemcpyAsync(&c, h->acc.p, 8, hipMemcpyDeviceToHost, h->stream));
    uint8_t tag = TAG_ZERO;
    if (h->maybe_one) BXMI_HIP(hipMemcpyAsync(&tag, h->tags.as<uint8_t>() + s / h->bin_size, 1, hipMemcpyDeviceToHost, h->stream));
    BXMI_HIP(hipStreamSynchronize(h->stream));
    if (tag == TAG_ONE) c -= s % h->bin_size;  // binBits.c:155,161
    *out = (int32_t)c;
    return BXMI_OK;
}

extern "C" int bxmi_bits_next(bxmi_bits_t *h, int32_t start, int val, int32_t *out)
{
    BXMI_TRY(check_pos(h, start, "bxmi_bits_next"));
    if (!out) return fail(BXMI_EINVAL, "bxmi_bits_next: out is NULL");
    BXMI_TRY(bits_stream(h));
    BXMI_TRY(h->acc.reserve(64));
    // The reference scans whole bins, i.e. up to nbins*bin_size, and a hit in the padding of the
    // last bin is reported as-is; the first such index is `size`, so clamping is exact.
    int64_t limit = h->total_bits;
    if (h->cap_words < h->nwords) {
        // zeros from the end of the allocated words on: a set bit can only be found before it, a clear one is there at the latest
        const int64_t cap = cap_bits(h);
        if (start >= cap) {
            *out = val ? h->size : start;
            return BXMI_OK;
        }
        limit = cap;
    }
    const int64_t scanned_to = limit;
    const unsigned long long none = ~0ull;
    int64_t w_begin = start >> 6, w_all = div_up(limit, 64);
    // stage 1: the next 64 Ki bits with one workgroup; stage 2: the rest of the chromosome, whole grid
    int64_t w_stage1 = w_begin + 1024 < w_all ? w_begin + 1024 : w_all;
    unsigned long long res = none;
    for (int stage = 0; stage < 2 && res == none; stage++) {
        int64_t a = stage == 0 ? w_begin : w_stage1, b = stage == 0 ? w_stage1 : w_all;
        if (a >= b) continue;
        BXMI_HIP(hipMemcpyAsync(h->acc.p, &none, 8, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(bits_find_kernel, dim3(bits_grid(b - a, BITS_THREADS)), dim3(BITS_THREADS), 0, h->stream,
                           h->words.as<unsigned long long>(), (int64_t)start, limit, val, a, b, h->acc.as<unsigned long long>());
        BXMI_LAUNCH_CHECK();
        BXMI_HIP(hipMemcpyAsync(&res, h->acc.p, 8, hipMemcpyDeviceToHost, h->stream));
        BXMI_HIP(hipStreamSynchronize(h->stream));
    }
    if (res == none && !val && scanned_to < h->size) res = (unsigned long long)scanned_to;  // the first bit past the allocated words is clear
    *out = (res == none || res >= (unsigned long long)h->size) ? h->size : (int32_t)res;
    return BXMI_OK;
}

static int same_shape(const bxmi_bits *a, const bxmi_bits *b, const char *who)
{
    if (!a || !b) return fail(BXMI_EINVAL, "%s: NULL handle", who);
    if (a->size != b->size) return fail(BXMI_EINVAL, "%s: BitSets must have the same size", who);
    if (a->flat != b->flat) return fail(BXMI_EINVAL, "%s: cannot mix BitSet and BinnedBitSet", who);
    if (a->bin_size != b->bin_size || a->nbins != b->nbins)
        return fail(BXMI_EINVAL, "%s: BitSets must have the same granularity (the reference's behaviour is undefined here)", who);
    return BXMI_OK;
}

template <int OP, bool COUNT>
static int bits_binary(bxmi_bits *h, const bxmi_bits *other, unsigned long long *acc_dev, hipStream_t st)
{
    // Lazily allocated operands (bits_need): past its allocated words a set is all zeros.  AND leaves nothing of `h` beyond
    // `other`'s words; OR / XOR need `h` to reach as far as `other` does.
    if (OP == 0) {
        if (other->cap_words < h->cap_words)
            BXMI_HIP(hipMemsetAsync(h->words.as<unsigned long long>() + other->cap_words, 0, (size_t)(h->cap_words - other->cap_words) * 8, st));
    } else if (h->cap_words < other->cap_words) {
        BXMI_TRY(bits_need(h, other->cap_words == other->nwords ? BITS_ALL : other->cap_words * 64));
    }
    const int64_t nw = h->cap_words < other->cap_words ? h->cap_words : other->cap_words;  // words both operands own
    const int64_t npairs = nw >> 1;
    const int64_t nb = h->flat || OP == 2 ? 0 : div_up(h->total_bits, h->bin_size);
    if (npairs == 0 && nb == 0) return BXMI_OK;
    // a workgroup moves BITS_UNROLL x 4 KiB per operand and sweep; counting variants stay at one workgroup per CU
    const int threads = COUNT && npairs <= BITS_WIDE_BELOW ? BITS_COUNT_THREADS : BITS_THREADS;
    int grid = bits_grid(npairs, threads * BITS_UNROLL);
    if (COUNT && grid > device_props().cus) grid = device_props().cus;
    hipLaunchKernelGGL((bits_binary_kernel<OP, COUNT>), dim3(grid), dim3(threads), 0, st,
                       h->words.as<unsigned long long>(), other->words.as<unsigned long long>(), npairs, (int64_t)h->size, acc_dev,
                       h->tags.as<uint8_t>(), other->tags.as<uint8_t>(), nb);
    BXMI_LAUNCH_CHECK();
    if (h->flat) return BXMI_OK;
    if (OP == 0)
        h->maybe_one = h->maybe_one && other->maybe_one;
    else
        h->maybe_one = h->maybe_one || other->maybe_one;
    return BXMI_OK;
}

extern "C" int bxmi_bits_and_dev(bxmi_bits_t *h, const bxmi_bits_t *other, void *stream)
{
    BXMI_TRY(same_shape(h, other, "bxmi_bits_and"));
    return bits_binary<0, false>(h, other, nullptr, as_stream(stream));
}

extern "C" int bxmi_bits_or_dev(bxmi_bits_t *h, const bxmi_bits_t *other, void *stream)
{
    BXMI_TRY(same_shape(h, other, "bxmi_bits_or"));
    return bits_binary<1, false>(h, other, nullptr, as_stream(stream));
}

extern "C" int bxmi_bits_and_count_dev(bxmi_bits_t *h, const bxmi_bits_t *other, int64_t *count_dev, void *stream)
{
    BXMI_TRY(same_shape(h, other, "bxmi_bits_and_count"));
    if (!count_dev) return fail(BXMI_EINVAL, "bxmi_bits_and_count_dev: count_dev is NULL");
    return bits_binary<0, true>(h, other, reinterpret_cast<unsigned long long *>(count_dev), as_stream(stream));
}

extern "C" int bxmi_bits_popcount_dev(bxmi_bits_t *h, int64_t *count_dev, void *stream)
{
    if (!h || !count_dev) return fail(BXMI_EINVAL, "bxmi_bits_popcount_dev: bad arguments");
    int64_t npairs = h->cap_words >> 1;  // (nothing is set past the allocated words)
    if (npairs == 0) return BXMI_OK;
    const int threads = npairs <= BITS_WIDE_BELOW ? BITS_COUNT_THREADS : BITS_THREADS;
    int grid = bits_grid(npairs, threads * BITS_UNROLL);
    if (grid > device_props().cus) grid = device_props().cus;
    hipLaunchKernelGGL(bits_popcount_kernel, dim3(grid), dim3(threads), 0, as_stream(stream),
                       h->words.as<unsigned long long>(), npairs, (int64_t)h->size, reinterpret_cast<unsigned long long *>(count_dev));
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

extern "C" int bxmi_bits_and(bxmi_bits_t *h, const bxmi_bits_t *other)
{
    BXMI_TRY(same_shape(h, other, "bxmi_bits_and"));
    BXMI_TRY(bits_stream(h));
    BXMI_TRY(bxmi_bits_and_dev(h, other, h->stream));
    BXMI_HIP(hipStreamSynchronize(h->stream));
    return BXMI_OK;
}

extern "C" int bxmi_bits_or(bxmi_bits_t *h, const bxmi_bits_t *other)
{
    BXMI_TRY(same_shape(h, other, "bxmi_bits_or"));
    BXMI_TRY(bits_stream(h));
    BXMI_TRY(bxmi_bits_or_dev(h, other, h->stream));
    BXMI_HIP(hipStreamSynchronize(h->stream));
    return BXMI_OK;
}

// BitSet.ixor (bitset.pyx:160-162, bits.c:244-253); only defined for flat sets, like the reference.
extern "C" int bxmi_bits_xor(bxmi_bits_t *h, const bxmi_bits_t *other)
{
    BXMI_TRY(same_shape(h, other, "bxmi_bits_xor"));
    if (!h->flat || !other->flat) return fail(BXMI_EINVAL, "bxmi_bits_xor: only flat BitSets support xor");
    BXMI_TRY(bits_stream(h));
    BXMI_TRY((bits_binary<2, false>(h, other, nullptr, h->stream)));
    BXMI_HIP(hipStreamSynchronize(h->stream));
    return BXMI_OK;
}

extern "C" int bxmi_bits_and_count(bxmi_bits_t *h, const bxmi_bits_t *other, int64_t *count)
{
    BXMI_TRY(same_shape(h, other, "bxmi_bits_and_count"));
    if (!count) return fail(BXMI_EINVAL, "bxmi_bits_and_count: count is NULL");
    BXMI_TRY(bits_stream(h));
    BXMI_TRY(h->acc.reserve(64));
    BXMI_HIP(hipMemsetAsync(h->acc.p, 0, 8, h->stream));
    BXMI_TRY(bxmi_bits_and_count_dev(h, other, h->acc.as<int64_t>(), h->stream));
    BXMI_HIP(hipMemcpyAsync(count, h->acc.p, 8, hipMemcpyDeviceToHost, h->stream));
    BXMI_HIP(hipStreamSynchronize(h->stream));
    return BXMI_OK;
}

extern "C" int bxmi_bits_not(bxmi_bits_t *h)
{
    if (!h) return fail(BXMI_EINVAL, "bxmi_bits_not: NULL handle");
    BXMI_TRY(bits_stream(h));
    BXMI_TRY(bits_need(h, BITS_ALL));  // every untouched bit becomes a one
    hipLaunchKernelGGL(bits_not_kernel, dim3(bits_grid(h->nwords, BITS_THREADS * 2)), dim3(BITS_THREADS), 0, h->stream,
                       h->words.as<unsigned long long>(), h->nwords, h->total_bits);
    int64_t nb = div_up(h->total_bits, h->bin_size);
    hipLaunchKernelGGL(tags_not_kernel, dim3(stream_grid(nb, 256)), dim3(256), 0, h->stream, h->tags.as<uint8_t>(), nb);
    BXMI_LAUNCH_CHECK();
    BXMI_HIP(hipStreamSynchronize(h->stream));
    h->maybe_one = !h->flat;  // every untouched bin is now ALL_ONE
    return BXMI_OK;
}

extern "C" int bxmi_bits_runs(bxmi_bits_t *h, int32_t from, int32_t *run_start, int32_t *run_end, int64_t cap, int64_t *n_runs)
{
    if (!h || !n_runs || cap < 0 || (cap > 0 && (!run_start || !run_end))) return fail(BXMI_EINVAL, "bxmi_bits_runs: bad arguments");
    if (from < 0 || from > h->size) return fail(BXMI_EINVAL, "bxmi_bits_runs: from=%d outside [0,%d]", from, h->size);
    *n_runs = 0;
    if (from == h->size) return BXMI_OK;
    BXMI_TRY(bits_stream(h));
    hipStream_t st = h->stream;
    // past the allocated words there is no set bit, hence no run: scan [from, min(size, allocated bits)) -- a run that
    // reaches the end of the allocated words ends there, as the next bit is a zero
    const int64_t size = h->cap_words < h->nwords && cap_bits(h) < h->size ? cap_bits(h) : h->size;
    if (from >= size) return BXMI_OK;
    int64_t w_first = from >> 6, w_last = (size - 1) >> 6;
    int64_t ntiles = div_up(w_last - w_first + 1, RUN_TILE);
    BXMI_TRY(h->tiles_s.reserve((size_t)(ntiles + 2) * 4));
    BXMI_TRY(h->tiles_l.reserve((size_t)(ntiles + 2) * 4));
    BXMI_TRY(h->acc.reserve(64));
    const unsigned long long *words = h->words.as<unsigned long long>();
    int32_t *ts = h->tiles_s.as<int32_t>(), *tl = h->tiles_l.as<int32_t>();
    hipLaunchKernelGGL(bits_runs_count_kernel, dim3((unsigned)ntiles), dim3(BITS_THREADS), 0, st, words, (int64_t)from, size, w_first, w_last,
                       ts, tl);
    BXMI_LAUNCH_CHECK();
    int32_t *tot = h->acc.as<int32_t>();
    BXMI_TRY((device_scan<int32_t, int32_t, OpSum, false>(ts, ts, ntiles, 0, tot, h->scan_tmp, st)));
    BXMI_TRY((device_scan<int32_t, int32_t, OpSum, false>(tl, tl, ntiles, 0, tot + 1, h->scan_tmp, st)));
    int32_t totals[2] = {0, 0};
    BXMI_HIP(hipMemcpyAsync(totals, tot, 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    if (totals[0] != totals[1]) return fail(BXMI_EHIP, "bxmi_bits_runs: internal error, %d starts vs %d ends", totals[0], totals[1]);
    *n_runs = totals[0];
    if (totals[0] > cap) return fail(BXMI_ERANGE, "bxmi_bits_runs: %d runs need a larger buffer than cap=%lld", totals[0], (long long)cap);
    if (totals[0] == 0) return BXMI_OK;
    BXMI_TRY(h->run_s.reserve((size_t)(totals[0] + 4) * 4));
    BXMI_TRY(h->run_e.reserve((size_t)(totals[0] + 4) * 4));
    hipLaunchKernelGGL(bits_runs_fill_kernel, dim3((unsigned)ntiles), dim3(BITS_THREADS), 0, st, words, (int64_t)from, size, w_first, w_last, ts,
                       tl, h->run_s.as<int32_t>(), h->run_e.as<int32_t>());
    BXMI_LAUNCH_CHECK();
    BXMI_HIP(hipMemcpyAsync(run_start, h->run_s.p, (size_t)totals[0] * 4, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipMemcpyAsync(run_end, h->run_e.p, (size_t)totals[0] * 4, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    return BXMI_OK;
}

// ---- groups ------------------------------------------------------------------
struct bxmi_bits_group {
    std::vector<bxmi_bits *> members;
    DevBuf table;
    int64_t total_chunks = 0, total_tags = 0;
};

extern "C" int bxmi_bits_group_create(bxmi_bits_t *const *members, int n, bxmi_bits_group_t **out)
{
    if (!members || !out || n < 1) return fail(BXMI_EINVAL, "bxmi_bits_group_create: bad arguments");
    bxmi_bits_group *g = new (std::nothrow) bxmi_bits_group();
    if (!g) return fail(BXMI_ENOMEM, "bxmi_bits_group_create: host allocation failed");
    std::vector<GroupSeg> segs((size_t)n);
    for (int i = 0; i < n; i++) {
        bxmi_bits *m = members[i];
        if (!m) {
            delete g;
            return fail(BXMI_EINVAL, "bxmi_bits_group_create: member %d is NULL", i);
        }
        g->members.push_back(m);
        if (bits_need(m, BITS_ALL) != BXMI_OK) {  // a group addresses whole members
            delete g;
            return BXMI_ENOMEM;
        }
        GroupSeg &s = segs[(size_t)i];
        s.words = m->words.as<unsigned long long>();
        s.tags = m->tags.as<uint8_t>();
        s.npairs = m->nwords >> 1;
        s.size_bits = m->size;
        s.chunk_first = g->total_chunks;
        s.tag_first = g->total_tags;
        s.ntags = div_up(m->total_bits, m->bin_size);
        g->total_chunks += div_up(s.npairs, MB_CHUNK_PAIRS);
        g->total_tags += s.ntags;
    }
    int rc = g->table.reserve(segs.size() * sizeof(GroupSeg));
    if (rc == BXMI_OK && hipMemcpy(g->table.p, segs.data(), segs.size() * sizeof(GroupSeg), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(BXMI_EHIP, "bxmi_bits_group_create: table upload failed");
    if (rc != BXMI_OK) {
        delete g;
        return rc;
    }
    *out = g;
    return BXMI_OK;
}

extern "C" int bxmi_bits_group_destroy(bxmi_bits_group_t *g)
{
    delete g;
    return BXMI_OK;
}

static int group_pair_check(const bxmi_bits_group *a, const bxmi_bits_group *b, const char *who)
{
    if (!a || !b) return fail(BXMI_EINVAL, "%s: NULL group", who);
    if (a->members.size() != b->members.size()) return fail(BXMI_EINVAL, "%s: groups have different member counts", who);
    for (size_t i = 0; i < a->members.size(); i++) BXMI_TRY(same_shape(a->members[i], b->members[i], who));
    return BXMI_OK;
}

template <int OP, bool COUNT>
static int group_launch(bxmi_bits_group *a, const bxmi_bits_group *b, int64_t *counts_dev, hipStream_t st)
{
    const int n = (int)a->members.size();
    int grid = (int)(a->total_chunks < (int64_t)device_props().cus * 8 ? a->total_chunks : (int64_t)device_props().cus * 8);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((bits_group_kernel<OP, COUNT>), dim3(grid), dim3(BITS_THREADS), 0, st, a->table.as<GroupSeg>(),
                       b ? b->table.as<GroupSeg>() : nullptr, n, a->total_chunks, reinterpret_cast<unsigned long long *>(counts_dev));
    if (OP == 0 || OP == 1) {
        hipLaunchKernelGGL((tags_group_kernel<OP>), dim3(stream_grid(a->total_tags, 256)), dim3(256), 0, st, a->table.as<GroupSeg>(),
                           b->table.as<GroupSeg>(), n, a->total_tags);
        for (int i = 0; i < n; i++) {
            bxmi_bits *x = a->members[(size_t)i];
            const bxmi_bits *y = b->members[(size_t)i];
            if (!x->flat) x->maybe_one = OP == 0 ? (x->maybe_one && y->maybe_one) : (x->maybe_one || y->maybe_one);
        }
    }
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

extern "C" int bxmi_bits_group_and_dev(bxmi_bits_group_t *g, const bxmi_bits_group_t *other, int64_t *counts_dev, void *stream)
{
    BXMI_TRY(group_pair_check(g, other, "bxmi_bits_group_and"));
    return counts_dev ? group_launch<0, true>(g, other, counts_dev, as_stream(stream))
                      : group_launch<0, false>(g, other, nullptr, as_stream(stream));
}

extern "C" int bxmi_bits_group_or_dev(bxmi_bits_group_t *g, const bxmi_bits_group_t *other, void *stream)
{
    BXMI_TRY(group_pair_check(g, other, "bxmi_bits_group_or"));
    return group_launch<1, false>(g, other, nullptr, as_stream(stream));
}

extern "C" int bxmi_bits_group_popcount_dev(bxmi_bits_group_t *g, int64_t *counts_dev, void *stream)
{
    if (!g || !counts_dev) return fail(BXMI_EINVAL, "bxmi_bits_group_popcount_dev: bad arguments");
    return group_launch<3, true>(g, nullptr, counts_dev, as_stream(stream));
}
