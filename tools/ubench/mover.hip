// mover.hip — what a plain data mover of the Shannon-Fano encoder's traffic shape reaches on this part (diagnostic tool, not
// part of libshafa_hip.so): every workgroup reads IN bytes and writes OUT = 13/16 IN bytes (the headline stream's 0.8123),
// nothing is computed.  Variants: one-shot grid or persistent grid-stride loop; fully coalesced 16-byte loads or the
// encoder's two 16-byte loads at a 32-byte lane stride; non-temporal or plain loads / stores; bytes per workgroup; stores
// straight from registers or through an LDS round trip with a barrier (as the encoder's windows).
// Build: hipcc --offload-arch=gfx950 -O3 -o mover mover.hip ; run: ./mover
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

typedef uint32_t u32;
typedef uint64_t u64;
typedef u32 v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <bool NT> __device__ __forceinline__ v4u ld(const v4u *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(v4u *p, v4u v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// U = 16-byte loads per lane (IN = 256 * 16 * U bytes per workgroup and iteration), stores 13 of every 16 loaded pieces.
// STRIDE32: lane l loads pieces 2 l and 2 l + 1 of each 8 KiB chunk (two instructions, half of every line each).
// LDSRT: the pieces go through LDS and a barrier before they are stored (ds_write_b128 / ds_read_b128).
template <int U, bool STRIDE32, bool NTL, bool NTS, bool LDSRT, bool PERSIST>
__global__ __launch_bounds__(256) void k_move(const v4u *__restrict__ in, v4u *__restrict__ out, u64 nwg)
{
    __shared__ v4u lds[LDSRT ? 256 * U : 1];
    const u32 tid = threadIdx.x;
    for (u64 w = blockIdx.x; w < nwg; w += PERSIST ? gridDim.x : nwg) {
        const v4u *ib = in + w * (256ull * U);
        v4u *ob = out + w * (208ull * U);             // 13/16 of the pieces
        v4u v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const u32 idx = STRIDE32 ? (u32)(k >> 1) * 512u + 2u * tid + (u32)(k & 1) : (u32)k * 256u + tid;
            v[k] = ld<NTL>(ib + idx);
        }
        if (LDSRT) {
#pragma unroll
            for (int k = 0; k < U; ++k) lds[k * 256 + tid] = v[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < U; ++k) v[k] = lds[k * 256 + ((tid + 64) & 255)];
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const u32 idx = (u32)k * 256u + tid;      // output piece: the first 208 U of the 256 U
            if (idx < 208u * U) st<NTS>(ob + idx, v[k]);
        }
        if (LDSRT && PERSIST) __syncthreads();
    }
}

template <typename F>
static float time_ms(F f, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main(int argc, char **argv)
{
    // `mover headline`: the two figures bench.py quotes next to the encoder (roofline_encode.mover_*), one line each
    const bool brief = argc > 1 && !strcmp(argv[1], "headline");
    const u64 n = 8ull << 30;                          // input bytes per launch, as the headline's 128 x 64 MiB
    v4u *d_a, *d_b;
    CK(hipMalloc(&d_a, n)); CK(hipMalloc(&d_b, n));
    CK(hipMemset(d_a, 1, n)); CK(hipMemset(d_b, 2, n));
    const double bytes = (double)n * (1.0 + 13.0 / 16.0);
#define RUN(NAME, U, S32, NTL, NTS, LDSRT, PERSIST, GRID) { \
        const u64 nwg = n / (256ull * 16 * U); \
        const u32 grid = PERSIST ? (u32)(GRID) : (u32)nwg; \
        float ms = time_ms([&] { hipLaunchKernelGGL((k_move<U, S32, NTL, NTS, LDSRT, PERSIST>), dim3(grid), dim3(256), 0, 0, d_a, d_b, nwg); }, 5); \
        printf("%-64s %7.3f ms  %5.2f TB/s  (of 8 TB/s: %.3f)\n", NAME, ms, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0); }
    if (brief) {
        RUN("one-shot 32 KiB/WG coalesced, nt ld, nt st", 8, false, true, true, false, false, 0)
        RUN("persistent 2048 WGs, 32 KiB/iter coalesced, plain ld, nt st", 8, false, false, true, false, true, 2048)
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        RUN("one-shot 32 KiB/WG coalesced, plain ld, nt st", 8, false, false, true, false, false, 0)
        RUN("one-shot 32 KiB/WG coalesced, nt ld, nt st", 8, false, true, true, false, false, 0)
        RUN("one-shot 32 KiB/WG coalesced, plain ld, plain st", 8, false, false, false, false, false, 0)
        RUN("one-shot 32 KiB/WG stride-32 loads, plain ld, nt st", 8, true, false, true, false, false, 0)
        RUN("one-shot 32 KiB/WG stride-32 loads, nt ld, nt st", 8, true, true, true, false, false, 0)
        RUN("one-shot 32 KiB/WG stride-32, plain ld, nt st, LDS round trip", 8, true, false, true, true, false, 0)
        RUN("one-shot 16 KiB/WG coalesced, plain ld, nt st", 4, false, false, true, false, false, 0)
        RUN("one-shot 16 KiB/WG stride-32, plain ld, nt st, LDS round trip", 4, true, false, true, true, false, 0)
        RUN("one-shot 64 KiB/WG coalesced, plain ld, nt st", 16, false, false, true, false, false, 0)
        RUN("one-shot 8 KiB/WG coalesced, plain ld, nt st", 2, false, false, true, false, false, 0)
        RUN("persistent 2048 WGs, 32 KiB/iter coalesced, plain ld, nt st", 8, false, false, true, false, true, 2048)
        RUN("persistent 1792 WGs (7/CU), 32 KiB/iter stride-32, LDS round trip", 8, true, false, true, true, true, 1792)
    }
    return 0;
}
