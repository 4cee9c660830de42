// Unaligned LDS accesses on gfx950: are ds_read_b32 / ds_write_b32 at byte addresses correct, and what do they cost?
//   hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip && ./lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
typedef unsigned char u8;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int REPS = 4096;

// correctness: every lane reads 4 bytes at byte offset tid * 5 + k and writes 4 bytes at tid * 7 + 1
__global__ void k_check(u32 *out)
{
    __shared__ __attribute__((aligned(16))) u8 s[8192];
    const u32 tid = threadIdx.x;
    for (u32 i = tid; i < 8192; i += 256) s[i] = (u8)(i * 37 + 11);
    __syncthreads();
    const u32 sb = (u32)(size_t)s;
    u32 v;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(sb + tid * 5 + (tid & 3)) : "memory");
    out[tid] = v;
    __syncthreads();
    if ((tid & 1) == 0) asm volatile("ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(sb + 4096 + tid * 7 + 1), "v"(0xA1B2C3D4u + tid) : "memory");
    __syncthreads();
    for (u32 i = tid; i < 2048; i += 256) out[256 + i] = s[4096 + i];
}

// mode 0: aligned ds_read_b32, rows 17 words apart + word offset; 1: unaligned ds_read_b32 rows 68 bytes apart + byte offset
// 2: aligned ds_read2_b32; 3: unaligned ds_write_b32 (rows ~40 bytes apart); 4: aligned ds_write_b32; 5: ds_or_b32 aligned
// 6: ds_read_u8 random 4 KiB table; 7: ds_read_u16 random 8 KiB table; 8: ds_read_b32 random 16 KiB table
// 9: unaligned ds_read_b64
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(u32 *out, const u32 *rnd, u32 seed)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 12288; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    const u32 sb = (u32)(size_t)lds;
    u32 ad[8];
    for (int j = 0; j < 8; ++j) {
        const u32 rv = rnd[(tid * 8 + j + seed) & 4095];
        const u32 prog = 4 * j + (rv & 7);                  // byte progress inside the row: lanes differ by a few bytes
        if (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 5) ad[j] = sb + tid * 68 + (prog & ~3u);
        if (MODE == 1 || MODE == 9) ad[j] = sb + tid * 68 + prog;
        if (MODE == 3) ad[j] = sb + tid * 41 + prog;
        if (MODE == 6) ad[j] = sb + (rv >> 3) % 4096;
        if (MODE == 7) ad[j] = sb + ((rv >> 3) % 4096) * 2;
        if (MODE == 8) ad[j] = sb + ((rv >> 3) % 4096) * 4;
    }
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0 || MODE == 1 || MODE == 8) { u32 v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(ad[j])); asm volatile("" :: "v"(v)); }
            if (MODE == 2) { unsigned long long v; asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v) : "v"(ad[j])); asm volatile("" :: "v"(v)); }
            if (MODE == 9) { unsigned long long v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(ad[j])); asm volatile("" :: "v"(v)); }
            if (MODE == 3 || MODE == 4) asm volatile("ds_write_b32 %0, %1" :: "v"(ad[j]), "v"(seed) : "memory");
            if (MODE == 5) asm volatile("ds_or_b32 %0, %1" :: "v"(ad[j]), "v"(seed) : "memory");
            if (MODE == 6) { u32 v; asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(ad[j])); asm volatile("" :: "v"(v)); }
            if (MODE == 7) { u32 v; asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(ad[j])); asm volatile("" :: "v"(v)); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (lds[tid] == 77 && seed == 99) out[tid] = 1;
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
    u32 *d_out; CK(hipMalloc(&d_out, 1 << 20));
    {
        hipLaunchKernelGGL(k_check, dim3(1), dim3(256), 0, 0, d_out);
        std::vector<u32> h(256 + 2048);
        CK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
        int bad_r = 0, bad_w = 0;
        std::vector<u8> img(2048);
        for (int i = 0; i < 2048; ++i) img[i] = (u8)((4096 + i) * 37 + 11);
        for (int t = 0; t < 256; ++t) {
            const u32 a = t * 5 + (t & 3);
            u32 want = 0;
            for (int b = 0; b < 4; ++b) want |= (u32)(u8)((a + b) * 37 + 11) << (8 * b);
            if (h[t] != want) { if (bad_r < 4) printf("read tid %d addr %u: got %08x want %08x\n", t, a, h[t], want); ++bad_r; }
            if ((t & 1) == 0) { const u32 v = 0xA1B2C3D4u + t; for (int b = 0; b < 4; ++b) img[t * 7 + 1 + b] = (u8)(v >> (8 * b)); }
        }
        for (int i = 0; i < 2048; ++i) if (h[256 + i] != img[i]) { if (bad_w < 4) printf("write byte %d: got %02x want %02x\n", i, h[256 + i], img[i]); ++bad_w; }
        printf("unaligned ds_read_b32: %s (%d bad)   unaligned ds_write_b32: %s (%d bad)\n", bad_r ? "WRONG" : "ok", bad_r, bad_w ? "WRONG" : "ok", bad_w);
    }
    std::vector<u32> h_rnd(4096);
    { unsigned long long s = 12345; for (auto &x : h_rnd) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = (u32)(s >> 33); } }
    u32 *d_rnd; CK(hipMalloc(&d_rnd, 4096 * 4));
    CK(hipMemcpy(d_rnd, h_rnd.data(), 4096 * 4, hipMemcpyHostToDevice));
    const double GHZ = 2.4;
    const int wps = 2, grid = 256 * wps;
#define RUNL(M, NAME) { float ms = time_ms([&] { hipLaunchKernelGGL(k_lds<M>, dim3(grid), dim3(256), 49152, 0, d_out, d_rnd, 1u); }, 3); \
        double inst = (double)REPS * 8 * wps * 4;  /* wave instructions per CU */ \
        printf("  %-44s %8.3f ms  %.2f cyc/inst/CU\n", NAME, ms, ms * 1e-3 * GHZ * 1e9 / inst); }
    RUNL(0, "ds_read_b32 aligned, rows 17 words") RUNL(1, "ds_read_b32 UNALIGNED, rows 68 bytes") RUNL(2, "ds_read2_b32 aligned, rows 17 words")
    RUNL(9, "ds_read_b64 UNALIGNED, rows 68 bytes")
    RUNL(4, "ds_write_b32 aligned") RUNL(3, "ds_write_b32 UNALIGNED rows 41 bytes") RUNL(5, "ds_or_b32 aligned")
    RUNL(6, "ds_read_u8 random 4 KiB") RUNL(7, "ds_read_u16 random 8 KiB") RUNL(8, "ds_read_b32 random 16 KiB")
    return 0;
}
