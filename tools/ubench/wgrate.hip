// wgrate.hip — how many workgroups per microsecond does the chip start and retire?  (diagnostic tool, not part of
// libshafa_hip.so.)  rle_decode_kernel and sfe6_kernel both run one tile per workgroup and both retire ~100 workgroups
// per microsecond whatever their residency; this measures the rate of workgroups that do nothing, with the same
// shape (256 lanes, LDS footprint as a parameter), and of workgroups that copy one tile of T bytes.
//   hipcc --offload-arch=gfx950 -O3 -o wgrate wgrate.hip && ./wgrate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;
typedef uint64_t u64;
typedef u32 v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_empty(u32 *out)
{
    extern __shared__ u32 lds[];
    if (threadIdx.x == 0) lds[0] = blockIdx.x;
    __syncthreads();
    if (lds[0] == 0xFFFFFFFFu) out[0] = 1;
}

template <int BPL>      // bytes per lane: 16 * BPL/16 loads
__global__ __launch_bounds__(256) void k_copy(const v4u *__restrict__ in, v4u *__restrict__ out)
{
    extern __shared__ u32 lds[];
    const u64 base = (u64)blockIdx.x * (256 * BPL / 16);
    v4u v[BPL / 16];
#pragma unroll
    for (int i = 0; i < BPL / 16; ++i) v[i] = __builtin_nontemporal_load(in + base + i * 256 + threadIdx.x);
    if (threadIdx.x == 0) lds[0] = v[0].x;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BPL / 16; ++i) __builtin_nontemporal_store(v[i], out + base + i * 256 + threadIdx.x);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const size_t bytes = 2ull << 30;
    v4u *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
    u32 *d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int lds_sizes[] = {1024, 16384, 25600, 40960};
    hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int l : lds_sizes) {
        const int n = 262144;
        hipLaunchKernelGGL(k_empty, dim3(n), dim3(256), l, 0, d);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_empty, dim3(n), dim3(256), l, 0, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("empty workgroups, %5d B LDS: %d in %.3f ms = %.1f per us\n", l, n, ms, n / ms / 1e3);
    }
#define COPY(BPL) { const int n = (int)(bytes / (256 * BPL)); \
        hipLaunchKernelGGL(k_copy<BPL>, dim3(n), dim3(256), 16384, 0, a, b); hipDeviceSynchronize(); hipEventRecord(e0); \
        hipLaunchKernelGGL(k_copy<BPL>, dim3(n), dim3(256), 16384, 0, a, b); hipEventRecord(e1); hipEventSynchronize(e1); \
        float ms; hipEventElapsedTime(&ms, e0, e1); \
        printf("copy %3d B per lane (%6d B tiles, 16 KiB LDS): %7d workgroups in %.3f ms = %.1f per us, %.2f TB/s read+write\n", BPL, 256 * BPL, n, ms, n / ms / 1e3, 2.0 * bytes / ms / 1e9); }
    COPY(16) COPY(32) COPY(64) COPY(128)
    return 0;
}
