// chain_ilp.hip — do dependent LDS table look-ups (the Shannon-Fano decoder's walks: sfd_spec, sfd_wstage) run faster with
// TWO independent chains per lane (instruction-level parallelism) than with one, at the same number of chains per CU?
// (diagnostic tool, not part of libshafa_hip.so.)  A chain is the decoder's step reduced to its dependences:
//   window = (stream >> used) & mask ; e = table[window] (ds_read_u8, 1 KiB table) ; used += e & 15 ; every third look-up the
//   32-bit stream word advances (two ds_read_b32 of the lane's LDS row, rows 17 words apart as in the decoder).
// The CU's LDS bounds the chains in flight in the real kernels (stream rows + image per chain), so the comparison that
// matters is C chains per lane x (32 / C) waves per CU: same chains per CU, fewer waves, more independent work per wave.
//   hipcc --offload-arch=gfx950 -O3 -o chain_ilp chain_ilp.hip && ./chain_ilp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;
typedef uint64_t u64;

constexpr int STEPS = 3000;                            // fetches per chain (three look-ups each)
constexpr int ROW = 17;                                // LDS words per chain row

template <int C>
__global__ __launch_bounds__(256) void k_chain(u32 *out, u32 seed)
{
    extern __shared__ u32 lds[];                       // [table 1 KiB][C * 256 rows of ROW words][padding to set the residency]
    const u32 tid = threadIdx.x;

    unsigned char *tab = (unsigned char *)lds;
    for (u32 i = tid; i < 1024; i += 256) tab[i] = (unsigned char)(5 + ((i * 2654435761u + seed) >> 29));      // 5..12 bits a look-up
    u32 *rows = lds + 256;
    for (u32 i = tid; i < (u32)C * 256u * ROW; i += 256) rows[i] = i * 2246822519u + seed;
    __syncthreads();
    u32 q[C], acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { q[c] = (tid * 7u + (u32)c * 3u) & 31u; acc[c] = 0; }
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const u32 *r = rows + ((u32)c * 256u + tid) * ROW + ((q[c] >> 5) & 15u);
            const u32 w = __builtin_amdgcn_alignbit(r[1], r[0], q[c]);
            u32 used = 0, e = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                e = tab[(w >> used) & 1023u];
                used += e & 15u;
                acc[c] += e;
            }
            q[c] += used;
        }
    }
    u32 x = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) x ^= acc[c] + q[c];
    if (x == 0x12345u) out[tid] = x;
}

template <int C>
static void run(int wgs_per_cu, int cus, u32 *d)
{
    // dynamic LDS chosen so that exactly wgs_per_cu workgroups of 256 fit a CU's 160 KiB (never less than the kernel needs)
    size_t need = 1024 + (size_t)C * 256 * ROW * 4;
    size_t lds = 163840 / wgs_per_cu / 256 * 256;
    if (lds < need) { printf("C=%d: %d workgroups per CU do not fit\n", C, wgs_per_cu); return; }
    if (lds > 65536) hipFuncSetAttribute((const void *)k_chain<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 4;
    hipLaunchKernelGGL(k_chain<C>, dim3(cus * wgs_per_cu * rounds), dim3(256), lds, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_chain<C>, dim3(cus * wgs_per_cu * rounds), dim3(256), lds, 0, d, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double lookups = (double)cus * wgs_per_cu * rounds * 256.0 * C * STEPS * 3.0;
    printf("%d chain(s) per lane, %2d waves per CU (%4d chains per CU): %8.3f ms  %.2f wave-look-ups per ns per CU-cycle-ish: %.3f G lane-look-ups/s per CU, %.1f cycles per wave-look-up per CU at 2.1 GHz\n",
           C, wgs_per_cu * 4, wgs_per_cu * 256 * C, ms, 0.0, lookups / ms / 1e6 / cus, ms * 1e-3 * 2.1e9 / (lookups / 64.0 / cus));
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    u32 *d; hipMalloc(&d, 4096);
    const int cus = p.multiProcessorCount;
    for (int rep = 0; rep < 2; ++rep) {
        run<1>(8, cus, d);     // 32 waves, 2048 chains: sfd_spec today
        run<2>(4, cus, d);     // 16 waves, 2048 chains
        run<4>(2, cus, d);     //  8 waves, 2048 chains
        run<1>(6, cus, d);     // 24 waves, 1536 chains: sfd_wstage today
        run<2>(3, cus, d);     // 12 waves, 1536 chains
        run<2>(8, cus, d);     // 32 waves, 4096 chains (if the LDS allowed it)
        run<1>(4, cus, d);     // 16 waves, 1024 chains
    }
    return 0;
}
