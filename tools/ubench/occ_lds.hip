// Workgroups of 256 per CU by dynamic LDS size: what hipOccupancyMaxActiveBlocksPerMultiprocessor answers, and what a
// launch shows (a kernel that only waits ~50 us, 24 workgroups per CU: its time is the number of rounds the CU needs).
//   hipcc --offload-arch=gfx950 -O3 -o occ_lds occ_lds.hip && ./occ_lds
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned *o, unsigned long long ticks)
{
    extern __shared__ unsigned s[];
    s[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (s[255 - threadIdx.x] == 12345u) o[threadIdx.x] = 1;
}
int main()
{
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    unsigned *d; hipMalloc(&d, 4096);
    const int per_cu = 24;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int lds : {16384, 19840, 20480, 20736, 20992, 21504, 25728, 26112, 26624, 26880, 27136, 27264, 27296, 27392, 27648, 32768, 40448, 40960, 41216, 41472, 54000, 65536, 81920, 82176}) {
        int n = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k, 256, lds);
        hipLaunchKernelGGL(k, dim3(p.multiProcessorCount * per_cu), dim3(256), lds, 0, d, 5000ull);     // 100 MHz clock: 50 us
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(p.multiProcessorCount * per_cu), dim3(256), lds, 0, d, 5000ull);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double rounds = ms * 1e3 / 50.0;
        printf("dynamic LDS %6d B: API says %d per CU; %d workgroups per CU took %.0f us = %.1f rounds -> %.1f resident per CU\n", lds, n, per_cu, ms * 1e3, rounds, per_cu / rounds);
    }
    return 0;
}
