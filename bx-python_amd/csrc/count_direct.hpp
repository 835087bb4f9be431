// count_direct.hpp -- index build kernels, the 32-ary search trees (device side + the host-side Tree that builds them) and the direct count kernel (ivl_count_kernel: small batches, indexes with reversed targets).  intersection.pyx:112-116 (order), :169-189 (find).
// Included by intervals.hip (one translation unit; the kernels share its constants and device helpers).
#pragma once

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

}  // namespace bxmi
