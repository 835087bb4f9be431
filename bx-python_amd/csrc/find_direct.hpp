// find_direct.hpp -- the direct find kernels (window + count, ballot-compacted fill), the one-query kernel behind IntervalTree.find(start, end) per call, and the neighbour window filter.  intersection.pyx:180-189, :232-260, :400-406.
// Included by intervals.hip (one translation unit; the kernels share its constants and device helpers).
#pragma once

namespace bxmi {

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

}  // namespace bxmi
