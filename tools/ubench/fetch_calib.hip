// fetch_calib.hip — what does FETCH_SIZE count for sfd_scan's loads?  MI355X_MICROARCH.md: FETCH_SIZE reports half of
// the bytes of a coalesced 16 B/lane streaming read on gfx950 ("128-B requests tallied at 64 B": double it) and is
// UNCALIBRATED for other shapes.  sfd_scan reads 32 contiguous bytes per lane pair at a 256-byte stride, a strip's four
// 64-byte phases one after the other.  Three kernels read the same N bytes exactly once:
//   k_stream : 16 B per lane, fully coalesced (the calibrated shape)
//   k_strips : sfd_scan's shape, the four phases back to back
//   k_strips_paused : the same with a pause between phases (the walk: the other half of a 128-byte line is asked for
//                     microseconds later)
// run under  rocprofv3 --pmc FETCH_SIZE --kernel-trace  and compare the counter (KB) with N.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ unsigned sink;
__global__ __launch_bounds__(256) void k_stream(const u32x4_t *in, size_t n16)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc ^= in[i].x;
    if (acc == 0x12345678u) sink = acc;
}
template <int PAUSE>
__global__ __launch_bounds__(256) void k_strips(const unsigned char *in, size_t nwaves)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= nwaves) return;
    const unsigned lane = threadIdx.x & 63, px = lane >> 1, ph = lane & 1;
    const unsigned char *p = in + wave * 16384 + (size_t)px * 256 + 16 * ph;
    unsigned acc = 0;
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc ^= ((const u32x4_t *)(p + (size_t)(t >> 1) * 32 * 256 + 64 * k + 32 * (t & 1)))->x;
        for (int s = 0; s < PAUSE; ++s) __builtin_amdgcn_s_sleep(127);
    }
    if (acc == 0x12345678u) sink = acc;
}
int main()
{
    const size_t N = (size_t)4 << 30;
    unsigned char *d;
    if (hipMalloc(&d, N) != hipSuccess) return 1;
    hipMemset(d, 1, N);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, (const u32x4_t *)d, N / 16);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_strips<0>, dim3((unsigned)(N / 16384 / 4)), dim3(256), 0, 0, d, N / 16384);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_strips<4>, dim3((unsigned)(N / 16384 / 4)), dim3(256), 0, 0, d, N / 16384);
    hipDeviceSynchronize();
    printf("N = %zu bytes per kernel\n", N);
    return 0;
}
