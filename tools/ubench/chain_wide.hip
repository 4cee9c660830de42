// chain_wide.hip — would the Shannon-Fano decoder's walks gain from WIDER table windows if the stream were fetched 64 bits
// at a time?  (diagnostic tool, not part of libshafa_hip.so.)  Today (sfd_spec / sfd_wstage on the headline data): a 10-bit
// window consumes 7.85 bits per look-up on average and three look-ups share one 32-bit fetch of the lane's LDS row; a 12-bit
// window would consume ~9.3 bits but only two fit a 32-bit fetch (measured slower in round 2: more fetches).  With a 64-bit
// fetch five 12-bit look-ups share one.  The kernels below are the walks reduced to their LDS and VALU operations:
//   A: 1 KiB byte table (10-bit window), 32-bit fetch, 3 look-ups per fetch         (today)
//   B: 4 KiB byte table (12-bit window), 64-bit fetch, 5 look-ups per fetch
//   C: 4 KiB dword table (10-bit, the symbol pass's sym3), 32-bit fetch, 3 per fetch (today's sfd_wstage look-up)
//   D: 16 KiB dword table (12-bit), 64-bit fetch, 5 per fetch
// Entries hold "bits consumed" drawn so that the averages are 7.85 (10-bit) and 9.3 (12-bit); the figure of merit is
// nanoseconds per stream bit per CU.    hipcc --offload-arch=gfx950 -O3 -o chain_wide chain_wide.hip && ./chain_wide
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;
typedef uint64_t u64;
constexpr int STEPS = 2000;

// MODE 0: A, 1: B, 2: C, 3: D.  rows: ROWW words per lane (odd: no bank conflicts between lanes)
template <int MODE>
__global__ __launch_bounds__(256) void k_walk(u32 *out, u32 seed, u32 *bits_out)
{
    extern __shared__ u32 lds[];
    constexpr bool WIDE = MODE == 1 || MODE == 3, DW = MODE >= 2;
    constexpr u32 K = WIDE ? 12 : 10, TABN = 1u << K, ROWW = WIDE ? 19 : 17;
    const u32 tid = threadIdx.x;
    unsigned char *tab8 = (unsigned char *)lds;
    u32 *tab32 = lds;
    for (u32 i = tid; i < TABN; i += 256) {
        const u32 r = (i * 2654435761u + seed) >> 8;
        // 10-bit window: 5..10 bits, mean 7.85; 12-bit window: 6..12, mean 9.3
        const u32 used = WIDE ? 6u + (r % 100 < 45 ? 3u + r % 4 : r % 100 < 80 ? 2u + r % 3 : r % 3)
                              : 5u + (r % 100 < 50 ? 3u + r % 3 : r % 100 < 80 ? 2u + r % 2 : r % 3);
        if (DW) tab32[i] = (used << 24) | (r & 0xFFFFFFu); else tab8[i] = (unsigned char)used;
    }
    u32 *rows = lds + (DW ? TABN : TABN / 4);
    for (u32 i = tid; i < 256u * ROWW; i += 256) rows[i] = i * 2246822519u + seed;
    __syncthreads();
    u32 q = (tid * 7u) & 31u, acc = 0, total = 0;
    const u32 *row = rows + tid * ROWW;
    for (int s = 0; s < STEPS; ++s) {
        const u32 wi = (q >> 5) & 15u;
        if (!WIDE) {
            const u32 w = __builtin_amdgcn_alignbit(row[wi + 1], row[wi], q);
            u32 used = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const u32 idx = (w >> used) & (TABN - 1);
                const u32 e = DW ? tab32[idx] : (u32)tab8[idx];
                used += DW ? (e >> 24) : e;
                acc += e;
            }
            q += used; total += used;
        } else {
            const u32 lo = __builtin_amdgcn_alignbit(row[wi + 1], row[wi], q), hi = __builtin_amdgcn_alignbit(row[wi + 2], row[wi + 1], q);
            const u64 w = ((u64)hi << 32) | lo;
            u32 used = 0;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const u32 idx = (u32)(w >> used) & (TABN - 1);
                const u32 e = DW ? tab32[idx] : (u32)tab8[idx];
                used += DW ? (e >> 24) : e;
                acc += e;
            }
            q += used; total += used;
        }
    }
    if (acc == 0x12345u) out[tid] = acc;
    if (tid == 0 && blockIdx.x == 0) *bits_out = total;
}

template <int MODE>
static void run(const char *name, int wgs_per_cu, int cus, u32 *d, u32 *dbits)
{
    constexpr bool WIDE = MODE == 1 || MODE == 3, DW = MODE >= 2;
    const size_t need = (DW ? 4 : 1) * (size_t)(WIDE ? 4096 : 1024) + 256 * (WIDE ? 19 : 17) * 4;
    size_t lds = 163840 / wgs_per_cu / 256 * 256;
    if (lds < need) { printf("%s: %d workgroups per CU do not fit (%zu B each)\n", name, wgs_per_cu, need); return; }
    if (lds > 65536) hipFuncSetAttribute((const void *)k_walk<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 4;
    hipLaunchKernelGGL(k_walk<MODE>, dim3(cus * wgs_per_cu * rounds), dim3(256), lds, 0, d, 1u, dbits);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_walk<MODE>, dim3(cus * wgs_per_cu * rounds), dim3(256), lds, 0, d, 1u, dbits);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    u32 bits = 0; hipMemcpy(&bits, dbits, 4, hipMemcpyDeviceToHost);
    const double per_lane_bits = bits;                  // lane 0's walk: the others' are statistically the same
    const double tot_bits = per_lane_bits * 256.0 * cus * wgs_per_cu * rounds;
    printf("%-58s %2d waves/CU: %7.3f ms, %.2f bits per look-up, %.1f Gbit/s per CU\n", name, wgs_per_cu * 4, ms,
           per_lane_bits / (STEPS * (WIDE ? 5.0 : 3.0)), tot_bits / ms / 1e6 / cus);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    u32 *d, *db; hipMalloc(&d, 4096); hipMalloc(&db, 4);
    const int cus = p.multiProcessorCount;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("A byte table 1 KiB, 10-bit window, 32-bit fetch x3", 8, cus, d, db);
        run<1>("B byte table 4 KiB, 12-bit window, 64-bit fetch x5", 8, cus, d, db);
        run<1>("B byte table 4 KiB, 12-bit window, 64-bit fetch x5", 6, cus, d, db);
        run<2>("C dword table 4 KiB, 10-bit window, 32-bit fetch x3", 6, cus, d, db);
        run<3>("D dword table 16 KiB, 12-bit window, 64-bit fetch x5", 4, cus, d, db);
        run<3>("D dword table 16 KiB, 12-bit window, 64-bit fetch x5", 3, cus, d, db);
    }
    return 0;
}
