// does gfx950's LDS return the right bytes for 4/8/16-byte reads at 2-byte boundaries?  (hipcc emits ds_read_b32/b64/b128 for them)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) const unsigned short *lds_u16_p;
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) P4 { unsigned v; };
struct __attribute__((packed, aligned(2))) P8 { v2u v; };
struct __attribute__((packed, aligned(2))) P16 { v4u v; };
__global__ void k(unsigned *bad)
{
    __shared__ unsigned short a[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) a[i] = (unsigned short)(i * 40503u + 7u);
    __syncthreads();
    lds_u16_p p = (lds_u16_p)a;
    for (unsigned i = threadIdx.x; i < 8000; i += blockDim.x) {
        const unsigned x4 = ((const __attribute__((address_space(3))) P4 *)(p + i))->v;
        const v2u x8 = ((const __attribute__((address_space(3))) P8 *)(p + i))->v;
        const v4u x16 = ((const __attribute__((address_space(3))) P16 *)(p + i))->v;
        auto w = [&](unsigned j) { return (unsigned)a[i + 2 * j] | ((unsigned)a[i + 2 * j + 1] << 16); };
        if (x4 != w(0)) atomicAdd(&bad[0], 1u);
        if (x8.x != w(0) || x8.y != w(1)) atomicAdd(&bad[1], 1u);
        if (x16.x != w(0) || x16.y != w(1) || x16.z != w(2) || x16.w != w(3)) atomicAdd(&bad[2], 1u);
    }
}
int main()
{
    unsigned *d, h[3] = {0, 0, 0};
    hipMalloc(&d, 12);
    hipMemset(d, 0, 12);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d);
    hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("mismatches: b32 %u  b64 %u  b128 %u (of 8000 offsets each)\n", h[0], h[1], h[2]);
    return 0;
}
