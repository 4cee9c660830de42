// What the decoder's LDS operations cost on gfx950, one operation at a time, at the decoder's occupancy.
//   hipcc --offload-arch=gfx950 -O3 -o lds_ops lds_ops.hip && ./lds_ops
// Throughput: 256-lane workgroups, WPS per CU, 8 independent operations between waits; cycles per wave-instruction
// and CU.  Latency: one wave per SIMD, a chain of dependent look-ups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int REPS = 2048;

enum Mode {
    RD32_4K, RD32_1K, RD32_256B, RDU8_1K, RDU8_256B, BPERM, RD32_ROWS, RD2_ROWS, RD64_ROWS2, RD64_8K, RD64_4K,
    OR_ALL, OR_THIRD, WR_ALL, WR_THIRD, OR_LINEAR, WR_LINEAR, RD32_LINEAR, RD32_4K_HALF, RD32_4K_QUARTER, N_MODES
};

template <int MODE>
__global__ __launch_bounds__(256) void k_tp(u32 *out, const u32 *rnd, u32 seed)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const u32 tid = threadIdx.x;
    for (u32 i = tid; i < 10240; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    const u32 sb = (u32)(size_t)lds;
    u32 ad[8], act = 0;
    for (int j = 0; j < 8; ++j) {
        const u32 rv = rnd[(tid * 8 + j + seed) & 4095];
        const u32 jw = (j + (rv % 3u)) % 8u;                // word inside the lane's row: lanes within +-1 word of each other
        const u32 prog = 5 * j + (rv & 7);                  // byte progress inside the lane's output run
        switch (MODE) {
        case RD32_4K: case RD32_4K_HALF: case RD32_4K_QUARTER: ad[j] = sb + ((rv >> 3) % 1024u) * 4; break;
        case RD32_1K: ad[j] = sb + ((rv >> 3) % 256u) * 4; break;
        case RD32_256B: ad[j] = sb + ((rv >> 3) % 64u) * 4; break;
        case RDU8_1K: ad[j] = sb + ((rv >> 3) % 1024u); break;
        case RDU8_256B: ad[j] = sb + ((rv >> 3) % 256u); break;
        case BPERM: ad[j] = ((rv >> 3) % 64u) * 4; break;
        case RD32_ROWS: case RD2_ROWS: ad[j] = sb + (tid * 9 + jw) * 4; break;
        case RD64_ROWS2: ad[j] = sb + (tid * 9 + jw) * 8; break;          // rows of 8 {w[k], w[k+1]} pairs, 9 apart
        case RD64_8K: ad[j] = sb + ((rv >> 3) % 1024u) * 8; break;
        case RD64_4K: ad[j] = sb + ((rv >> 3) % 512u) * 8; break;
        case OR_ALL: case OR_THIRD: case WR_ALL: case WR_THIRD: ad[j] = sb + ((tid * 39 + prog) & ~3u); break;
        case OR_LINEAR: case WR_LINEAR: case RD32_LINEAR: ad[j] = sb + (tid + 256 * j) * 4; break;
        }
        bool a = true;
        if (MODE == OR_THIRD || MODE == WR_THIRD) a = (rv >> 20) % 10u < 3u;
        if (MODE == RD32_4K_HALF) a = (rv >> 20) % 2u == 0;
        if (MODE == RD32_4K_QUARTER) a = (rv >> 20) % 4u == 0;
        act |= (u32)a << j;
    }
    u32 sink = 0;
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (act & (1u << j)) {
                switch (MODE) {
                case RD32_4K: case RD32_1K: case RD32_256B: case RD32_ROWS: case RD32_LINEAR: case RD32_4K_HALF: case RD32_4K_QUARTER:
                    { u32 v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(ad[j])); asm volatile("" :: "v"(v)); } break;
                case RDU8_1K: case RDU8_256B:
                    { u32 v; asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(ad[j])); asm volatile("" :: "v"(v)); } break;
                case BPERM:
                    { u32 v; asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v) : "v"(ad[j]), "v"(seed)); asm volatile("" :: "v"(v)); } break;
                case RD2_ROWS:
                    { u64 v; asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v) : "v"(ad[j])); asm volatile("" :: "v"(v)); } break;
                case RD64_ROWS2: case RD64_8K: case RD64_4K:
                    { u64 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(ad[j])); asm volatile("" :: "v"(v)); } break;
                case OR_ALL: case OR_THIRD: case OR_LINEAR:
                    asm volatile("ds_or_b32 %0, %1" :: "v"(ad[j]), "v"(seed) : "memory"); break;
                case WR_ALL: case WR_THIRD: case WR_LINEAR:
                    asm volatile("ds_write_b32 %0, %1" :: "v"(ad[j]), "v"(seed) : "memory"); break;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (lds[tid] == 77 && seed == 99) out[tid] = sink;
}

// latency: v = table[f(v)] in a chain, one wave per SIMD
template <int KIND>      // 0 ds_read_b32 random 4 KiB, 1 ds_bpermute_b32, 2 ds_read_u8 1 KiB, 3 ds_read_b32 same address
__global__ __launch_bounds__(256) void k_lat(u32 *out, u32 seed)
{
    __shared__ u32 tab[1024];
    const u32 tid = threadIdx.x;
    for (u32 i = tid; i < 1024; i += 256) tab[i] = (i * 2654435761u) >> 7;
    __syncthreads();
    const u32 sb = (u32)(size_t)tab;
    u32 v = tid * 2654435761u + seed;
    u32 reg = (tid * 40503u) >> 3;
    for (int r = 0; r < REPS * 4; ++r) {
        if (KIND == 0) { const u32 a = sb + ((v & 1023u) << 2); asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a)); }
        if (KIND == 1) { const u32 a = (v & 63u) << 2; u32 w; asm volatile("ds_bpermute_b32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(a), "v"(reg)); v = w + r; }
        if (KIND == 2) { const u32 a = sb + (v & 1023u); u32 w; asm volatile("ds_read_u8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(a)); v = v * 5 + w; }
        if (KIND == 3) { const u32 a = sb + ((v & 0u) << 2); u32 w; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(a)); v = w + r; }
    }
    out[blockIdx.x * 256 + tid] = v;
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

int main()
{
    u32 *d_out; CK(hipMalloc(&d_out, 4 << 20));
    std::vector<u32> h_rnd(4096);
    { u64 s = 12345; for (auto &x : h_rnd) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = (u32)(s >> 33); } }
    u32 *d_rnd; CK(hipMalloc(&d_rnd, 4096 * 4));
    CK(hipMemcpy(d_rnd, h_rnd.data(), 4096 * 4, hipMemcpyHostToDevice));
    const double GHZ = 2.4;
    for (int wps : {3, 6}) {
        const int grid = 256 * wps;
        printf("throughput, %d workgroups of 256 per CU (cycles per wave-instruction and CU; a masked instruction counts as one)\n", wps);
#define RUNL(M, NAME) { float ms = time_ms([&] { hipLaunchKernelGGL(k_tp<M>, dim3(grid), dim3(256), 40960, 0, d_out, d_rnd, 1u); }, 3); \
        double inst = (double)REPS * 8 * wps * 4; \
        printf("  %-52s %8.3f ms  %6.2f\n", NAME, ms, ms * 1e-3 * GHZ * 1e9 / inst); }
        RUNL(RD32_LINEAR, "ds_read_b32 lane-linear")
        RUNL(RD32_4K, "ds_read_b32 random 4 KiB")
        RUNL(RD32_4K_HALF, "ds_read_b32 random 4 KiB, half of the lanes")
        RUNL(RD32_4K_QUARTER, "ds_read_b32 random 4 KiB, a quarter of the lanes")
        RUNL(RD32_1K, "ds_read_b32 random 1 KiB")
        RUNL(RD32_256B, "ds_read_b32 random 256 B")
        RUNL(RDU8_1K, "ds_read_u8 random 1 KiB")
        RUNL(RDU8_256B, "ds_read_u8 random 256 B")
        RUNL(BPERM, "ds_bpermute_b32 random lane")
        RUNL(RD64_8K, "ds_read_b64 random 8 KiB")
        RUNL(RD64_4K, "ds_read_b64 random 4 KiB")
        RUNL(RD32_ROWS, "ds_read_b32 rows 9 words apart, word +-1")
        RUNL(RD2_ROWS, "ds_read2_b32 rows 9 words apart, word +-1")
        RUNL(RD64_ROWS2, "ds_read_b64 rows 9 pairs apart, pair +-1")
        RUNL(WR_LINEAR, "ds_write_b32 lane-linear")
        RUNL(OR_LINEAR, "ds_or_b32 lane-linear")
        RUNL(WR_ALL, "ds_write_b32 runs 39 bytes apart, all lanes")
        RUNL(OR_ALL, "ds_or_b32 runs 39 bytes apart, all lanes")
        RUNL(WR_THIRD, "ds_write_b32 runs 39 bytes apart, 30 % of the lanes")
        RUNL(OR_THIRD, "ds_or_b32 runs 39 bytes apart, 30 % of the lanes")
    }
    printf("latency, one wave per SIMD (cycles per dependent look-up, includes the address arithmetic)\n");
#define RUNLAT(K, NAME) { float ms = time_ms([&] { hipLaunchKernelGGL(k_lat<K>, dim3(256), dim3(256), 0, 0, d_out, 1u); }, 3); \
        printf("  %-52s %8.3f ms  %6.1f\n", NAME, ms, ms * 1e-3 * GHZ * 1e9 / (REPS * 4.0)); }
    RUNLAT(0, "ds_read_b32 random 4 KiB") RUNLAT(3, "ds_read_b32 one address") RUNLAT(2, "ds_read_u8 random 1 KiB") RUNLAT(1, "ds_bpermute_b32")
    return 0;
}
