// common.hpp -- shared host-side plumbing of libbxmi (error handling, device
// buffers, launch geometry) and the wave64 device helpers every kernel uses.
// gfx950 only: wave = 64 lanes, 256 CUs in 8 XCDs, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/bxmi.h"

namespace bxmi {

// ---- error plumbing --------------------------------------------------------
std::string &last_error();
int fail(int code, const char *fmt, ...);

#define BXMI_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return ::bxmi::fail(_e == hipErrorOutOfMemory ? BXMI_ENOMEM : BXMI_EHIP, "%s: %s (%s:%d)", #expr, \
                                hipGetErrorString(_e), __FILE__, __LINE__);                     \
    } while (0)

#define BXMI_TRY(expr)            \
    do {                          \
        int _s = (expr);          \
        if (_s != BXMI_OK) return _s; \
    } while (0)

#define BXMI_LAUNCH_CHECK() BXMI_HIP(hipGetLastError())

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ---- device buffer ---------------------------------------------------------
// Grow-only HBM buffer.  288 GB per GPU: we never shrink, we keep scratch.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    // Ensure capacity (contents NOT preserved on growth unless keep==true).
    int reserve(size_t bytes, bool keep = false, hipStream_t st = nullptr)
    {
        if (bytes <= cap) return BXMI_OK;
        size_t want = bytes + (bytes >> 3) + 256;  // 12.5% slack, avoids regrow churn
        void *np = nullptr;
        hipError_t e = hipMalloc(&np, want);
        if (e != hipSuccess) return fail(BXMI_ENOMEM, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        if (keep && p && cap) {
            e = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) {
                (void)hipFree(np);
                return fail(BXMI_EHIP, "grow copy: %s", hipGetErrorString(e));
            }
        }
        if (p) (void)hipFree(p);
        p = np;
        cap = want;
        return BXMI_OK;
    }
    template <typename T>
    T *as() const
    {
        return reinterpret_cast<T *>(p);
    }
};

// ---- launch geometry --------------------------------------------------------
struct DeviceProps {
    int cus = 256;
    int device = -1;
};
const DeviceProps &device_props();

// Grid for a bandwidth-bound grid-stride kernel: enough workgroups to fill
// 256 CUs x 8 resident 256-thread blocks, a multiple of the 8 XCDs.
inline int stream_grid(int64_t work_items, int items_per_block)
{
    int64_t need = (work_items + items_per_block - 1) / items_per_block;
    int64_t cap = (int64_t)device_props().cus * 8;
    int64_t g = need < cap ? need : cap;
    if (g < 1) g = 1;
    return (int)g;
}

inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- wave64 device helpers --------------------------------------------------
#define BXMI_WAVE 64

// Pointers the kernels LOAD FROM MEMORY (the members of BmSeg / IndexDev in a segment table) are generic to the compiler: every
// access through them becomes a FLAT instruction, which counts against the vector-memory AND the LDS counter and may complete
// out of order between the two -- the compiler then waits for `vmcnt(0) lgkmcnt(0)` wherever it waits at all, and every LDS wait
// drains the loads and stores in flight.  (Kernel ARGUMENTS are known to be global.)  as_global() types such a pointer as what it
// is -- address space 1 -- and the accesses through the result are global_load / global_store with counted waits.
#define BX_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ T BX_GLOBAL *as_global(T *p)
{
    return (T BX_GLOBAL *)p;
}
// (int4 is a class: it cannot be copied out of or into another address space -- 16-byte accesses go through the built-in vector type)
typedef int bx_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 load_int4(const int32_t BX_GLOBAL *p)
{
    const bx_v4i v = *reinterpret_cast<const bx_v4i BX_GLOBAL *>(p);
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_int4(int32_t BX_GLOBAL *p, int a, int b, int c, int d)
{
    *reinterpret_cast<bx_v4i BX_GLOBAL *>(p) = bx_v4i{a, b, c, d};
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// DPP controls (gfx9 DPP16 encodings).
#define BXMI_DPP_QUAD_XOR1 0xB1   // quad_perm [1,0,3,2]
#define BXMI_DPP_QUAD_XOR2 0x4E   // quad_perm [2,3,0,1]
#define BXMI_DPP_HALF_MIRROR 0x141  // lane i <-> 7-i inside each 8-lane half row

// Sum over each aligned group of 8 lanes; every lane of the group gets the sum.
// Three DPP adds, no LDS traffic.
__device__ __forceinline__ int group8_sum_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, BXMI_DPP_QUAD_XOR1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, BXMI_DPP_QUAD_XOR2, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, BXMI_DPP_HALF_MIRROR, 0xf, 0xf, false);
    return v;
}

// Same through ds_bpermute (kept as the A/B baseline and a safety net).
__device__ __forceinline__ int group8_sum_shfl(int v)
{
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}

__device__ __forceinline__ unsigned long long lanemask_lt()
{
    return (1ull << lane_id()) - 1ull;
}

// Block-wide int64 sum -> one atomicAdd per block.  `red` must hold blockDim/64 slots.
__device__ __forceinline__ void block_accumulate_i64(long long v, long long *red, unsigned long long *global_acc)
{
    v = wave_sum_i64(v);
    int w = threadIdx.x >> 6;
    if (lane_id() == 0) red[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long s = 0;
        int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; i++) s += red[i];
        if (s) atomicAdd(global_acc, (unsigned long long)s);
    }
}

// ---- completion through host memory (the per-call latency paths) ------------
// A one-workgroup kernel that answers ONE call leaves its result in host-visible memory; its last act is to publish a
// sequence number next to it (system-scope release after a workgroup barrier: everything the workgroup wrote before
// is visible to whoever reads the number with acquire).  The host then polls that word instead of going through the
// stream's completion signal -- and falls back to a stream synchronisation if the number does not show up soon
// (queue busy with earlier work, `core.poll` = 0).
extern int64_t g_opt_poll;

__device__ __forceinline__ void publish_to_host(unsigned long long *flag, unsigned long long seq)
{
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int wait_for_host_flag(const unsigned long long *flag, unsigned long long seq, hipStream_t st);

}  // namespace bxmi
