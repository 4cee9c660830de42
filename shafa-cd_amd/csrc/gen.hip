// gen.hip — synthetic byte streams for bench/tests (no reference counterpart).
// byte i = map[r16] (or r16 >> 8), r16 = (splitmix64(seed + (i >> 2)) >> (16 * (i & 3))) & 0xFFFF:
// the very stream oracle/shafa_oracle.c orc_gen_bytes produces, so GPU inputs can be checked on the CPU.
#include "common.hpp"
#include "internal.hpp"

namespace {

__device__ __forceinline__ u64 splitmix64(u64 x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void gen_kernel(u64 seed, u64 first, const u8 *__restrict__ map,
                                                  u8 *__restrict__ out, u64 n)
{
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 p = t * 16;
    if (p >= n) return;
    u32 w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u64 r = splitmix64(seed + ((first + p) >> 2) + q);
        u32 acc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 r16 = (u32)(r >> (16 * j)) & 0xFFFFu;
            const u32 byte = map ? map[r16] : (r16 >> 8);
            acc |= byte << (8 * j);
        }
        w[q] = acc;
    }
    if (p + 16 <= n) {
        *(uint4 *)(out + p) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (p + j < n) out[p + j] = (u8)(w[j >> 2] >> (8 * (j & 3)));
    }
}

}  // namespace

int gen_launch(hipStream_t st, u64 seed, u64 first, const u8 *d_map, u8 *d_out, size_t n)
{
    if ((first & 15) || ((uintptr_t)d_out & 15)) return SHAFA_OUTSIDE_MODULE;
    if (!n) return SHAFA_SUCCESS;
    // at most 8 GiB per launch: a grid holds fewer than 2^32 threads (64 GiB in one launch is an invalid configuration)
    constexpr u64 PART = 8ull << 30;
    for (u64 done = 0; done < (u64)n; done += PART) {
        const u64 m = (u64)n - done < PART ? (u64)n - done : PART;
        const u64 threads = ceil_div_u64(m, 16);
        hipLaunchKernelGGL(gen_kernel, dim3((u32)ceil_div_u64(threads, 256)), dim3(256), 0, st, seed, first + done,
                           d_map, d_out + done, m);
    }
    HIP_TRY(hipGetLastError());
    return SHAFA_SUCCESS;
}
