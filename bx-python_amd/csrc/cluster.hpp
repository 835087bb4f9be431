// cluster.hpp -- distance clustering (ClusterTree; src/cluster.c:112-147, :226-260) from the sealed index: one comparison per interval.
// Included by intervals.hip (one translation unit; the kernels share its constants and device helpers).
#pragma once

namespace bxmi {

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

}  // namespace bxmi
