// scatter_store.hip — what do per-lane sequential 16-byte stores cost when every lane owns its own contiguous run?
// The shape of a symbol pass whose lanes own long strips: lane l of a wave writes RUN bytes (its strip's symbols) as
// 16-byte pieces, RUN / PH of them per phase, PH phases, with an optional pause between phases (the walk).
//   usage: scatter_store <run_bytes> <phases> <pause_sleeps> [coalesced=0|1]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_store(u32x4_t *out, int run16, int phases, int pause, int coalesced)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int per = run16 / phases;
    u32x4_t v = {1u, 2u, 3u, (unsigned)lane};
    u32x4_t *base = out + wave * 64 * run16;
    for (int p = 0; p < phases; ++p) {
        for (int j = 0; j < per; ++j) {
            const size_t idx = coalesced ? (size_t)(p * per + j) * 64 + lane : (size_t)lane * run16 + p * per + j;
            __builtin_nontemporal_store(v, base + idx);
        }
        for (int s = 0; s < pause; ++s) __builtin_amdgcn_s_sleep(127);
    }
}
__global__ __launch_bounds__(256) void k_store_t(u32x4_t *out, int run16, int phases, int pause, int coalesced)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int per = run16 / phases;
    u32x4_t v = {1u, 2u, 3u, (unsigned)lane};
    u32x4_t *base = out + wave * 64 * run16;
    for (int p = 0; p < phases; ++p) {
        for (int j = 0; j < per; ++j) {
            const size_t idx = coalesced ? (size_t)(p * per + j) * 64 + lane : (size_t)lane * run16 + p * per + j;
            base[idx] = v;
        }
        for (int s = 0; s < pause; ++s) __builtin_amdgcn_s_sleep(127);
    }
}
int main(int argc, char **argv)
{
    const int run = argc > 1 ? atoi(argv[1]) : 320, phases = argc > 2 ? atoi(argv[2]) : 4, pause = argc > 3 ? atoi(argv[3]) : 0;
    const size_t total = (size_t)8 << 30;
    const int run16 = run / 16;
    const size_t waves = total / ((size_t)64 * run16 * 16);
    u32x4_t *d;
    if (hipMalloc(&d, total) != hipSuccess) return 1;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int nt = 0; nt < 2; ++nt)
        for (int co = 0; co < 2; ++co) {
            float best = 1e9;
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(a);
                if (nt) hipLaunchKernelGGL(k_store, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, d, run16, phases, pause, co);
                else hipLaunchKernelGGL(k_store_t, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, d, run16, phases, pause, co);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("run %d B phases %d pause %d %s %s: %.3f ms  %.2f TB/s\n", run, phases, pause, nt ? "nt" : "temporal", co ? "coalesced" : "per-lane runs",
                   best, (double)waves * 64 * run16 * 16 / best / 1e9);
        }
    return 0;
}
