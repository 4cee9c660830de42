// sf_tables.hip — Module T's core on the device: Shannon-Fano code construction for the 256-bin histograms of a launch
// (reference t.c:74-210; host counterpart host/sfcodes.c, whose rule set this follows):
//   - symbols ranked by count, descending, ties by ascending symbol (t.c:87); only symbols with a non-zero count take part
//     (t.c:202-210); fewer than two of them => no codes;
//   - a range [a, b] of ranks is cut behind rank d, the first d at which |2 left - total| stops strictly decreasing
//     (t.c:138-149): '0' is appended to a..d, '1' to d + 1..b, both halves are cut the same way (t.c:187-193).
// One workgroup per block, one thread per rank.  x(d) = 2 left(d) - total increases strictly with d (the counts are
// positive), so with j the first rank at which x >= 0 (binary search over the prefix sums) the cut is j - 1 when j > a and
// -x(j - 1) <= x(j), else j.  A thread follows ITS OWN range down the tree — every thread of a range computes the same cut —
// so the descent needs no barrier and takes as many steps as the thread's code has bits (<= 255).
// Counts are the device's own histograms (their sum is a block's size): 64-bit sums.  A hand-made .freq with counts whose sum
// passes 2^64 is the host's case (host/sfcodes.c: 128-bit sums); this kernel flags it (SHAFA_OUTSIDE_MODULE) and leaves no codes.
// The launchers of the encoder and the decoder choose their kernels from the tables' longest code on the HOST, so the
// product path still builds its tables there; this entry point is the device-resident Module T for callers
// that keep histograms and tables on the GPU.
#include "common.hpp"
#include "internal.hpp"

namespace {

constexpr int SFT_THREADS = 256;

__global__ __launch_bounds__(SFT_THREADS) void sft_build_kernel(const u64 *__restrict__ freq, shafa_code_table *__restrict__ out,
                                                               int *__restrict__ err)
{
    __shared__ u64 cnt[SFT_THREADS];                    // counts by symbol, then by rank
    __shared__ u64 cum[SFT_THREADS + 1];                // cum[i] = sum of the first i ranked counts
    __shared__ u64 wtot[4];
    __shared__ u32 sym_of[SFT_THREADS];
    __shared__ u32 used_s, over_s;
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const u64 f = freq[(size_t)blockIdx.x * 256 + tid];
    cnt[tid] = f;
    if (tid == 0) { used_s = 0; over_s = 0; }
    __syncthreads();
    // rank of symbol tid: symbols with a larger count, or the same count and a smaller value, come first
    u32 rank = 0;
    for (u32 t = 0; t < 256; ++t) {
        const u64 g = cnt[t];
        rank += (g > f || (g == f && t < tid)) ? 1u : 0u;
    }
    if (f) atomicAdd(&used_s, 1u);
    __syncthreads();
    sym_of[rank] = tid;
    cnt[rank] = f;                                      // (every rank is taken exactly once)
    __syncthreads();
    const u32 used = used_s;
    // inclusive prefix sums of the ranked counts; a carry out of 64 bits anywhere is the host's case
    const u64 mine = cnt[tid];
    u64 incl = mine;
    bool over = false;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 t = __shfl_up(incl, d, 64);
        if (lane >= (u32)d) { over |= incl + t < incl; incl += t; }
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    u64 base = 0;
    for (u32 w = 0; w < wv; ++w) { over |= base + wtot[w] < base; base += wtot[w]; }
    over |= base + incl < base;
    if (over) over_s = 1u;
    cum[tid + 1] = base + incl;
    if (tid == 0) cum[0] = 0;
    __syncthreads();
    shafa_code_table &tab = out[blockIdx.x];
    u32 bits[8] = {0, 0, 0, 0, 0, 0, 0, 0};             // the code, MSB first: bit i of the code at bit 31 - i % 32 of word i / 32
    u32 len = 0;
    if (over_s) {
        if (tid == 0) set_error(err + blockIdx.x, SHAFA_OUTSIDE_MODULE);
    } else if (used >= 2 && tid < used) {
        u32 a = 0, b = used - 1;
        while (a < b) {
            const u64 ca = cum[a], total = cum[b + 1] - ca;
            // j = first rank in [a, b] with left(j) >= rest(j)  (x(b) = total > 0: it exists)
            u32 lo = a, hi = b;
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                const u64 left = cum[mid + 1] - ca;
                if (left >= total - left) hi = mid; else lo = mid + 1;
            }
            u32 cut = lo;
            if (lo > a) {
                const u64 lj = cum[lo + 1] - ca, lp = cum[lo] - ca;
                const u64 xj = lj - (total - lj), xp = (total - lp) - lp;     // x(j) >= 0, -x(j - 1) > 0
                if (xp <= xj) cut = lo - 1;
            }
            const bool one = tid > cut;
            if (one) {
                const u32 wi = len >> 5, m = 0x80000000u >> (len & 31u);
#pragma unroll
                for (u32 k = 0; k < 8; ++k) bits[k] |= k == wi ? m : 0u;
                a = cut + 1;
            } else {
                b = cut;
            }
            ++len;
        }
    }
    // rank tid's code goes to its symbol: len, and 32 bytes of bits (byte i = bits 8 i .. 8 i + 7 of the code, first bit on top)
    const u32 s = sym_of[tid];
    tab.len[s] = (u8)len;
    u32 *dst = (u32 *)tab.bits[s];
#pragma unroll
    for (u32 k = 0; k < 8; ++k) dst[k] = __builtin_bswap32(bits[k]);
}

}  // namespace

int sftab_launch(Batch *bt, hipStream_t st, int nblocks, const u64 *d_freq, shafa_code_table *d_tables)
{
    if (nblocks <= 0) return SHAFA_SUCCESS;
    if (nblocks > bt->max_blocks) return SHAFA_LACK_OF_MEMORY;
    hipLaunchKernelGGL(sft_build_kernel, dim3((u32)nblocks), dim3(SFT_THREADS), 0, st, d_freq, d_tables, bt->d_err);
    HIP_TRY(hipGetLastError());
    return SHAFA_SUCCESS;
}
