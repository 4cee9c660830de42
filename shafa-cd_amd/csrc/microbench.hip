// microbench.hip — calibration kernels for the encoder's front end (diagnostic only; not on the
// product path).  shafa_hip_microbench(mode, d_in, n, d_lut, d_out, items, threads) enqueues one launch:
//   mode 0: stream read (16 B/lane loads, xor-reduce)
//   mode 1: + one LDS LUT look-up per byte, sum of lengths
//   mode 2: + group building (the encoder's 4-symbol concatenation)
//   mode 3: + DPP wave scan of the item totals
#include "common.hpp"
#include "internal.hpp"

namespace {
struct MbBlk { const u8 *in; const u32 *lut; u64 n; const u64 *off; };

//   mode 4: mode 3 + zero a 10 KiB LDS window per workgroup (+ barrier)
//   mode 5: mode 4 + input/LUT pointers fetched from a parameter record in memory (as EncBlk)
//   mode 6: mode 5 + one 8-byte per-tile load (tile offset)
template <int MODE, int ITEMS, int THREADS>
__global__ __launch_bounds__(THREADS) void mb_kernel(const u8 *__restrict__ in, u64 n, const u32 *__restrict__ lutg,
                                                     u32 *__restrict__ out, const MbBlk *__restrict__ rec)
{
    __shared__ u32 lut[256];
    __shared__ u64 stage[MODE >= 7 ? 4096 : 1282];     // mode 7+: 32 KiB of LDS => at most 5 workgroups per CU
    __shared__ u32 wtot[8];
    const int tid = threadIdx.x;
    u64 extra = 0;
    if (MODE >= 5) {
        const MbBlk r = rec[blockIdx.y];
        in = r.in; lutg = r.lut; n = r.n;
        if (MODE >= 6) extra = r.off[blockIdx.x];
    }
    if (tid < 256) lut[tid] = lutg[tid];
    if (MODE >= 4) for (int i = tid; i < 1282; i += THREADS) stage[i] = 0;
    u32 scan_keep = 0;
    if (THREADS > 256 || true) __syncthreads();
    const u64 tile = (u64)blockIdx.x * THREADS * 16 * ITEMS;
    u32 accv = 0;
    u64 accg = 0;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const u64 idx = tile + (u64)it * THREADS * 16 + (u64)tid * 16;
        if (idx + 16 > n) continue;
        const uint4 v = *(const uint4 *)(in + idx);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
        if (MODE == 0) { accv ^= v.x ^ v.y ^ v.z ^ v.w; continue; }
        u32 tot = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32 e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = lut[(w[g] >> (8 * j)) & 0xFFu];
            const u32 l0 = (e[0] >> 16) & 31u, l1 = (e[1] >> 16) & 31u, l2 = (e[2] >> 16) & 31u, l3 = (e[3] >> 16) & 31u;
            tot += l0 + l1 + l2 + l3;
            if (MODE >= 2) {
                const u32 a = ((e[0] & 0xFFFFu) << l1) | (e[1] & 0xFFFFu);
                const u32 c = ((e[2] & 0xFFFFu) << l3) | (e[3] & 0xFFFFu);
                accg ^= ((u64)a << (l2 + l3)) | c;
            }
        }
        if (MODE >= 3) {
            u32 s = tot;
            s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x111, 0xf, 0xf, false);
            s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x112, 0xf, 0xf, false);
            s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xf, false);
            s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xf, 0xf, false);
            s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x142, 0xa, 0xf, false);
            s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x143, 0xc, 0xf, false);
            tot = s;
            if (MODE >= 8 && (tid & 63) == 63) wtot[(it & 1) * 4 + (tid >> 6)] = s;
        }
        accv += tot;
    }
    if (MODE >= 8) {
        __syncthreads();
        for (int w = 0; w < 8; ++w) scan_keep += wtot[w];
    }
    accv ^= (u32)accg ^ (u32)(accg >> 32) ^ (u32)extra ^ scan_keep;
    if (MODE >= 4) accv ^= (u32)stage[tid];
    if (accv == 0x9E3779B9u) out[blockIdx.x * THREADS + tid] = accv;     // practically never: keeps the work alive
}
}  // namespace

extern "C" int shafa_hip_microbench(int mode, const uint8_t *d_in, uint64_t n, const uint32_t *d_lut, uint32_t *d_out,
                                    int items, int threads, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    static MbBlk *d_rec = nullptr;
    static u64 *d_off = nullptr;
    if (!d_rec) {
        (void)hipMalloc((void **)&d_rec, sizeof(MbBlk));
        (void)hipMalloc((void **)&d_off, (n / 4096 + 16) * 8);
        (void)hipMemset(d_off, 0, (n / 4096 + 16) * 8);
        MbBlk h = {d_in, d_lut, n, d_off};
        (void)hipMemcpy(d_rec, &h, sizeof(h), hipMemcpyHostToDevice);
    }
#define MB(M, I, T)                                                                                   \
    if (mode == M && items == I && threads == T) {                                                    \
        const u64 per = (u64)T * 16 * I;                                                              \
        hipLaunchKernelGGL((mb_kernel<M, I, T>), dim3((u32)((n + per - 1) / per)), dim3(T), 0, st, d_in, n, d_lut, d_out, d_rec); \
        return hipGetLastError() == hipSuccess ? 0 : 9;                                               \
    }
    MB(0, 4, 256) MB(1, 4, 256) MB(2, 4, 256) MB(3, 4, 256)
    MB(0, 1, 256) MB(1, 1, 256) MB(2, 1, 256) MB(3, 1, 256)
    MB(0, 2, 256) MB(1, 2, 256) MB(2, 2, 256) MB(3, 2, 256)
    MB(4, 2, 256) MB(5, 2, 256) MB(6, 2, 256) MB(4, 4, 256) MB(6, 4, 256) MB(7, 2, 256) MB(8, 2, 256) MB(7, 4, 256) MB(8, 4, 256)
    return 1;
}
