// hist.hip — Module F histogram: make_freq (reference f.c:63-79) on gfx950.
//
// 256-bin byte histogram with 64-bit bins, many blocks per launch.  Each workgroup streams one
// 256 KiB chunk with coalesced 16-byte loads and counts into an LDS histogram that is replicated
// 32x, replica = lane & 31 at dword sym * 32 + replica: every lane of a half-wave owns its bank, so
// the LDS atomics never conflict, whatever the symbol distribution is (a hot symbol of Zipf data is
// in ~25 % of the lanes).  The replicas are summed (rotated reads, conflict-free) and added to the
// block's global bins with at most 256 atomics per workgroup.
//
// hist256_tiles_kernel (round 4): the same pass also leaves the 256 x u16 histogram of every 32 KiB tile (the sidecar that
// lets sf_encode6.hip run as a one-shot grid): running totals per replica, a tile's counts = the totals after it minus
// the totals before it, summed over the replicas with eight rotated ds_read_b128 per lane while the next tile's loads fly.
//
// Algorithmic HBM bytes per block: n read (+ 2 KiB written; with tile histograms + n / 64 written).
#include "common.hpp"
#include "internal.hpp"

namespace {

constexpr int HIST_THREADS = 256;
constexpr u64 HIST_CHUNK = 256 * 1024;
constexpr int HIST_REP = 32;

struct HistBlk { const u8 *in; u64 n; u64 *freq; const u64 *n_dev; u16 *thist; };
constexpr u64 HIST_TILE = 32768;                       // SHAFA_TILE_BYTES: a tile of the sidecar (sf_encode6.hip)

// OF_RLE: the launch that follows rle_encode on its output (f.c:310), sizes on the device — the same code under a name of its
// own, so that a profile tells the two histograms of Module F apart
template <bool OF_RLE>
__global__ __launch_bounds__(HIST_THREADS) void hist256_kernel(const HistBlk *__restrict__ blks)
{
    __shared__ u32 h[256 * HIST_REP];
    const int tid = threadIdx.x;
    const HistBlk blk = blks[blockIdx.y];
    const u64 n = blk.n_dev ? *blk.n_dev : blk.n;      // size produced on the device (RLE output)
    const u64 start = (u64)blockIdx.x * HIST_CHUNK;
    if (start >= n) return;
    const u64 end = (start + HIST_CHUNK < n) ? start + HIST_CHUNK : n;

    for (int i = tid; i < 256 * HIST_REP; i += HIST_THREADS) h[i] = 0;
    __syncthreads();

    const u32 rep = tid & (HIST_REP - 1);
    auto count16 = [&](const uint4 v) {
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u32 sym = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            atomicAdd(&h[sym * HIST_REP + rep], 1u);
        }
    };
    constexpr u64 STEP = (u64)HIST_THREADS * 16;
    u64 p = start + (u64)tid * 16;
    for (; p + 3 * STEP + 16 <= end; p += 4 * STEP) {   // four loads in flight per lane
        const uint4 v0 = gload_nt<uint4>(blk.in + p), v1 = gload_nt<uint4>(blk.in + p + STEP);
        const uint4 v2 = gload_nt<uint4>(blk.in + p + 2 * STEP), v3 = gload_nt<uint4>(blk.in + p + 3 * STEP);
        count16(v0);
        count16(v1);
        count16(v2);
        count16(v3);
    }
    for (; p < end; p += STEP) {
        if (p + 16 <= end) count16(*(const uint4 *)(blk.in + p));
        else for (u64 q = p; q < end; ++q) atomicAdd(&h[(u32)blk.in[q] * HIST_REP + rep], 1u);
    }
    __syncthreads();
    u32 c = 0;
#pragma unroll
    for (int r = 0; r < HIST_REP; ++r) c += h[tid * HIST_REP + ((r + tid) & (HIST_REP - 1))];
    if (c) atomicAdd((unsigned long long *)(blk.freq + tid), (unsigned long long)c);
}

// The same with the SIDECAR of the one-shot Shannon-Fano encoder (sf_encode6.hip): every 32 KiB tile's own histogram,
// 256 x u16 (a tile holds at most 32768 of one byte), at thist + 256 * tile.  The LDS replicas keep RUNNING totals over the
// workgroup's eight tiles — nothing is zeroed between tiles; behind each tile every lane sums the 32 replicas of its symbol
// and stores the difference to the previous sum.  Two barriers per tile: counts complete before they are summed, sums
// read before the next tile's counts arrive; the next tile's loads are in flight meanwhile.
template <bool OF_RLE>
__global__ __launch_bounds__(HIST_THREADS) void hist256_tiles_kernel(const HistBlk *__restrict__ blks)
{
    __shared__ __attribute__((aligned(16))) u32 h[256 * HIST_REP];
    const int tid = threadIdx.x;
    const HistBlk blk = blks[blockIdx.y];
    const u64 n = blk.n_dev ? *blk.n_dev : blk.n;
    const u64 start = (u64)blockIdx.x * HIST_CHUNK;
    if (start >= n) return;
    const u64 end = (start + HIST_CHUNK < n) ? start + HIST_CHUNK : n;

    for (int i = tid; i < 256 * HIST_REP; i += HIST_THREADS) h[i] = 0;
    __syncthreads();

    const u32 rep = tid & (HIST_REP - 1);
    auto count16 = [&](const uint4 v) {
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u32 sym = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            atomicAdd(&h[sym * HIST_REP + rep], 1u);
        }
    };
    constexpr u64 STEP = (u64)HIST_THREADS * 16;
    constexpr int TL = (int)(HIST_TILE / STEP);        // 16-byte loads per lane and tile: 8
    u16 *th = blk.thist + (start / HIST_TILE) * 256;
    u32 prev = 0;
    auto sum_tile = [&]() {                            // behind the counts of a tile: its histogram = sums - previous sums
        __syncthreads();
        u32 c = 0;                                     // the lane's 32 replicas as eight 16-byte reads, rotated by tid / 2: the
        const uint4 *row = (const uint4 *)(h + tid * HIST_REP);          //   sixteen lanes of an LDS pass hit sixteen bank quads
#pragma unroll
        for (int j = 0; j < HIST_REP / 4; ++j) {
            const uint4 v = row[(j + (tid >> 1)) & (HIST_REP / 4 - 1)];
            c += v.x + v.y + v.z + v.w;
        }
        th[tid] = (u16)(c - prev);
        prev = c;
        th += 256;
        __syncthreads();
    };
    // full tiles: all eight loads of a tile in flight, and the NEXT tile's requested before this tile's barriers, so that the
    // two barriers and the sums of a tile boundary do not leave the memory pipe idle
    const u64 nfull = (end - start) / HIST_TILE;
    uint4 cur[TL], nxt[TL];
    if (nfull) {
#pragma unroll
        for (int k = 0; k < TL; ++k) cur[k] = gload_nt<uint4>(blk.in + start + (u64)k * STEP + (u64)tid * 16);
    }
    for (u64 t = 0; t < nfull; ++t) {
        if (t + 1 < nfull) {
            const u8 *nb = blk.in + start + (t + 1) * HIST_TILE + (u64)tid * 16;
#pragma unroll
            for (int k = 0; k < TL; ++k) nxt[k] = gload_nt<uint4>(nb + (u64)k * STEP);
        }
#pragma unroll
        for (int k = 0; k < TL; ++k) count16(cur[k]);
        sum_tile();
#pragma unroll
        for (int k = 0; k < TL; ++k) cur[k] = nxt[k];
    }
    const u64 ts = start + nfull * HIST_TILE;          // the block's ragged last tile
    if (ts < end) {
        u64 p = ts + (u64)tid * 16;
        for (; p < end; p += STEP) {
            if (p + 16 <= end) count16(*(const uint4 *)(blk.in + p));
            else for (u64 q = p; q < end; ++q) atomicAdd(&h[(u32)blk.in[q] * HIST_REP + rep], 1u);
        }
        sum_tile();
    }
    if (prev) atomicAdd((unsigned long long *)(blk.freq + tid), (unsigned long long)prev);
}

}  // namespace

// h_in_n[b] is the block size, or (with d_n != NULL) an upper bound of the size the device wrote to d_n[b]
int hist_launch_dev(Batch *bt, hipStream_t st, int nblocks, const u8 *d_in, const u64 *h_in_off,
                    const u64 *h_in_n, const u64 *d_n, u64 *d_freq, u8 *d_thist, const u64 *h_thist_off)
{
    if (nblocks <= 0) return SHAFA_SUCCESS;
    if (nblocks > bt->max_blocks) return SHAFA_LACK_OF_MEMORY;
    if ((d_thist == nullptr) != (h_thist_off == nullptr)) return SHAFA_OUTSIDE_MODULE;
    const size_t pbytes = (size_t)nblocks * sizeof(HistBlk);
    HistBlk *hp = (HistBlk *)batch_stage(bt, st, pbytes);
    if (!hp) return SHAFA_LACK_OF_MEMORY;
    u64 max_n = 0;
    for (int b = 0; b < nblocks; ++b) {
        if ((h_in_off[b] & 15) || (d_thist && (h_thist_off[b] & 15))) return SHAFA_OUTSIDE_MODULE;
        hp[b].thist = d_thist ? (u16 *)(d_thist + h_thist_off[b]) : nullptr;
        hp[b].in = d_in + h_in_off[b];
        hp[b].n = h_in_n[b];
        hp[b].freq = d_freq + (size_t)b * 256;
        hp[b].n_dev = d_n ? d_n + b : nullptr;
        if (hp[b].n > max_n) max_n = hp[b].n;
    }
    HIP_TRY(hipMemsetAsync(d_freq, 0, (size_t)nblocks * 256 * sizeof(u64), st));
    if (max_n == 0) return SHAFA_SUCCESS;
    { const int urc = batch_upload(bt, st, bt->d_par_hist, hp, pbytes); if (urc) return urc; }
    const dim3 grid((u32)ceil_div_u64(max_n, HIST_CHUNK), (u32)nblocks);
    const HistBlk *dp = (const HistBlk *)bt->d_par_hist;
    if (d_thist && d_n) hipLaunchKernelGGL(hist256_tiles_kernel<true>, grid, dim3(HIST_THREADS), 0, st, dp);
    else if (d_thist) hipLaunchKernelGGL(hist256_tiles_kernel<false>, grid, dim3(HIST_THREADS), 0, st, dp);
    else if (d_n) hipLaunchKernelGGL(hist256_kernel<true>, grid, dim3(HIST_THREADS), 0, st, dp);
    else hipLaunchKernelGGL(hist256_kernel<false>, grid, dim3(HIST_THREADS), 0, st, dp);
    HIP_TRY(hipGetLastError());
    return SHAFA_SUCCESS;
}

int hist_launch(Batch *bt, hipStream_t st, int nblocks, const u8 *d_in, const u64 *h_in_off,
                const u64 *h_in_n, u64 *d_freq)
{
    return hist_launch_dev(bt, st, nblocks, d_in, h_in_off, h_in_n, nullptr, d_freq);
}
