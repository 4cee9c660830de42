#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned *o) { extern __shared__ unsigned s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); o[threadIdx.x] = s[255 - threadIdx.x]; }
int main()
{
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int lds : {16384, 18640, 19840, 20480, 21000, 22000, 22500, 23000, 23400, 23405, 24000, 25728, 26000, 27000, 32768, 40000, 54000, 65536}) {
        int n = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k, 256, lds);
        printf("dynamic LDS %6d B: %d workgroups of 256 per CU\n", lds, n);
    }
    return 0;
}
