// sf_encode3.hip — Shannon-Fano bit-pack encoder for codes <= 32 bits as three dependency-free kernels (the path for
// launches with few blocks and for 17..32-bit codes; launches with many blocks of <= 16-bit codes: sf_encode4.hip).
//
// With few blocks per launch a chained scan has too few independent chains to hide its look-backs, so the prefix
// problem is split off into its own kernels:
//
//   sfe3_count : per tile, sum of code lengths                    (reads n; byte LUT, no grouping)
//   sfe3_scan  : per block, exclusive scan of the tile totals     (4096 tiles per 64 MiB block)
//   sfe3_pack  : per tile, look up + group + merge into an LDS bit-stream + aligned stores
//
// No tickets, no descriptors, no inter-workgroup waiting; every workgroup is independent, which is
// what lets the hardware run 6-8 of them per CU.  The price is a second read of the input (n more
// bytes of traffic; it is served from the 256 MiB Infinity Cache when the launch covers <= ~2 blocks).
#include "common.hpp"
#include "internal.hpp"

#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int E3_THREADS = 256;
constexpr int E3_ITEMS = 2;                            // 16-byte items per lane and tile
constexpr int E3_TILE = E3_THREADS * 16 * E3_ITEMS;    // 8 KiB of symbols
constexpr int E3_SW64 = 1280;                          // LDS bit-stream window: 10 KiB; tiles that expand more take two rounds

// ------------------------------------------------------------------------------------------------
// count: tile_bits[tile] = sum of code lengths; grid = (tiles, blocks)
// ------------------------------------------------------------------------------------------------
constexpr int E3_CTPW = 16;                            // tiles per workgroup of sfe3_count (one LUT fill)

// LUT64: the block's table holds 64-bit entries code | len << 32 (codes of up to 32 bits)
template <bool LUT64>
__global__ __launch_bounds__(E3_THREADS) void sfe3_count(const EncBlk *__restrict__ blks, u32 *__restrict__ tile_bits)
{
    // Entry = len | absent << 16: a lane sums 128 of them (len sum <= 4096), so one add per symbol carries
    // both the bit total and the number of symbols without a code.  The 256 entries are replicated 32 times,
    // copy c at dword sym * 32 + c, and lane l reads copy l & 31: every look-up hits its own bank, so a wave's
    // 64 random look-ups take the minimum two LDS passes instead of ~4.5 with a shared 1 KiB table.
    __shared__ __attribute__((aligned(16))) u32 lut[256 * 32];
    const EncBlk blk = blks[blockIdx.y];
    const u32 tile0 = blockIdx.x * E3_CTPW;
    if (tile0 >= blk.n_tiles) return;
    const int tid = threadIdx.x;
    {
        u32 e;
        if (LUT64) {
            const u32 len = gload<u32>((const u32 *)blk.lut + 2 * tid + 1);
            e = len ? len : (1u << 16);                // empty code = symbol absent from the table
        } else {
            const u32 x = gload<u32>((const u32 *)blk.lut + tid);
            e = ((x >> 16) & 31u) | ((x >> 31) << 16);
        }
        uint4 *dst = (uint4 *)(lut + tid * 32);
#pragma unroll
        for (int c = 0; c < 8; ++c) dst[c] = make_uint4(e, e, e, e);
    }
    __syncthreads();
    const u32 copy4 = (u32)(tid & 31) << 2;            // byte offset of this lane's copy inside a 128-byte row
    const u8 *lutb = (const u8 *)lut;
    const u32 lane = tid & 63, wv = tid >> 6;
    const u32 tile_end = tile0 + E3_CTPW < blk.n_tiles ? tile0 + E3_CTPW : blk.n_tiles;
    u32 absent = 0;
    // after the fill the waves are independent: wave w counts tiles tile0 + w, + 4, ... (8 KiB = 8 x 16 B per lane),
    // the next tile's loads are issued before the current tile's look-ups
    constexpr int CI = E3_TILE / (64 * 16);
    uint4 cur[CI], nxt[CI];
    auto load = [&](u32 tile, uint4 *v) {
        const u64 base = (u64)tile * E3_TILE;
        if (tile < tile_end && base + E3_TILE <= blk.n) {
#pragma unroll
            for (int it = 0; it < CI; ++it) v[it] = gload<uint4>(blk.in + base + (u64)it * 1024 + (u64)lane * 16);
        }
    };
    load(tile0 + wv, cur);
    for (u32 tile = tile0 + wv; tile < tile_end; tile += 4) {
        const u64 base = (u64)tile * E3_TILE;
        load(tile + 4, nxt);
        u32 tot = 0;
        if (base + E3_TILE <= blk.n) {
#pragma unroll
            for (int it = 0; it < CI; ++it) {
                const u32 w[4] = {cur[it].x, cur[it].y, cur[it].z, cur[it].w};
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    tot += *(const u32 *)(lutb + ((((w[j >> 2] >> (8 * (j & 3))) & 0xFFu) << 7) | copy4));
            }
        } else {                                       // ragged last tile of the block
            for (int it = 0; it < CI; ++it) {
                const u64 idx = base + (u64)it * 1024 + (u64)lane * 16;
                for (int j = 0; j < 16; ++j)
                    if (idx + j < blk.n) tot += lut[(u32)gload<u8>(blk.in + idx + j) * 32];
            }
        }
        absent |= tot >> 16;
        const u32 s = dpp_scan_add(tot & 0xFFFFu);
        if (lane == 63) tile_bits[blk.desc_base + tile] = s;
#pragma unroll
        for (int it = 0; it < CI; ++it) cur[it] = nxt[it];
    }
    if (absent) set_error_over(blk.err, SHAFA_FILE_UNRECOGNIZABLE, SHAFA_LACK_OF_MEMORY);
}

// ------------------------------------------------------------------------------------------------
// scan: per block, tile_off[tile] = bits before the tile; out_n[b] = ceil(total / 8)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(E3_THREADS) void sfe3_scan(const EncBlk *__restrict__ blks, const u32 *__restrict__ tile_bits,
                                                        u64 *__restrict__ tile_off)
{
    __shared__ u64 wtot[4];
    __shared__ u64 carry;
    const EncBlk blk = blks[blockIdx.x];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (u32 t0 = 0; t0 < blk.n_tiles; t0 += E3_THREADS) {
        const u32 t = t0 + tid;
        const u64 c = t < blk.n_tiles ? (u64)tile_bits[blk.desc_base + t] : 0ull;
        const u64 incl = wave_incl_scan_add<u64>(c);
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        u64 base = carry;
        for (u32 w = 0; w < wv; ++w) base += wtot[w];
        if (t < blk.n_tiles) tile_off[blk.desc_base + t] = base + incl - c;
        __syncthreads();
        if (tid == 0) carry += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
    if (tid == 0) gstore<u64>(blk.out_n, (carry + 7) >> 3);
}

// ------------------------------------------------------------------------------------------------
// pack: the tile's groups are OR-ed into a zeroed LDS window of 64-bit words that is ALREADY aligned to
// the global output (bit offset = tile-local offset + (B mod 64)), so storing is a straight copy.  A tile
// owns every output u64 that begins inside it; the (B mod 64) leading bits of its first word belong
// to the previous tile and are re-encoded from the 64 input bytes before the tile.
// ------------------------------------------------------------------------------------------------
struct Pack3Shared {
    u64 stage[E3_SW64 + 2];
    u64 lut[256];                // LUT64: code | len << 32; else only the low halves are used: code | len << 16
    u32 wtot[4 * E3_ITEMS];
    u32 prev[16];
};

__device__ __forceinline__ u64 bswap64(u64 x)
{
    return ((u64)bswap32((u32)x) << 32) | bswap32((u32)(x >> 32));
}

template <bool LUT64>
__global__ __launch_bounds__(E3_THREADS) void sfe3_pack(const EncBlk *__restrict__ blks, const u64 *__restrict__ tile_off)
{
    __shared__ __attribute__((aligned(16))) Pack3Shared sh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);            // wave-uniform: keeps the wave selects scalar
    const EncBlk blk = blks[blockIdx.y];
    const u32 tile = blockIdx.x;
    uint4 cur[E3_ITEMS];
    if (tile >= blk.n_tiles) return;
    const u64 base = (u64)tile * E3_TILE;
    const bool full = base + E3_TILE <= blk.n;

    // everything this tile needs from memory is requested up front, nothing depends on anything else
    const u64 B = tile_off[blk.desc_base + tile];
#pragma unroll
    for (int it = 0; it < E3_ITEMS; ++it) {
        const u64 idx = base + (u64)it * (E3_THREADS * 16) + (u64)tid * 16;
        if (full) cur[it] = gload<uint4>(blk.in + idx);
        else {
            u32 w[4] = {0, 0, 0, 0};
            if (idx < blk.n) {
                const int nv = blk.n - idx >= 16 ? 16 : (int)(blk.n - idx);
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (q < nv) w[q >> 2] |= (u32)gload<u8>(blk.in + idx + q) << (8 * (q & 3));
            }
            cur[it] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    u32 dropmask[E3_ITEMS] = {};
    u32 pv = 0;
    if (tid < 16 && tile > 0) pv = gload<u32>(blk.in + base - 64 + 4 * tid);
    {
        // no code: "length" 4096, caught in the item totals below
        if (LUT64) {
            const u64 x = gload<u64>((const u64 *)blk.lut + tid);
            sh.lut[tid] = (x >> 32) ? x : (1ull << 44);
        } else {
            const u32 x = gload<u32>((const u32 *)blk.lut + tid);
            ((u32 *)sh.lut)[tid] = (x >> 31) ? (1u << 28) : x;
        }
    }
    for (int i = tid; i < E3_SW64 + 2; i += E3_THREADS) sh.stage[i] = 0;
    if (tid < 16) sh.prev[tid] = pv;
    __syncthreads();                                                                           // 1

    // ---- look up + group (4 symbols -> one group of <= 64 bits) ---------------------------------------
    u64 grp[E3_ITEMS][4];
    u32 glen[E3_ITEMS];
    u32 itot[E3_ITEMS];
    if (!full) {                                       // ragged last tile: symbols past the block encode as nothing
#pragma unroll
        for (int it = 0; it < E3_ITEMS; ++it) {
            const u64 idx = base + (u64)it * (E3_THREADS * 16) + (u64)tid * 16;
            const u32 keep = idx >= blk.n ? 0u : (blk.n - idx >= 16 ? 16u : (u32)(blk.n - idx));
            dropmask[it] = keep >= 16 ? 0u : (0xFFFFu << keep) & 0xFFFFu;      // bytes past the block (loaded as 0)
        }
    }
#pragma unroll
    for (int it = 0; it < E3_ITEMS; ++it) {
        const u32 wds[4] = {cur[it].x, cur[it].y, cur[it].z, cur[it].w};
        const u32 dm = dropmask[it];
        u32 tot = 0, packed = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32 L;
            if (LUT64) {
                u64 e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = sh.lut[(wds[g] >> (8 * j)) & 0xFFu];
                if (__builtin_expect(dm != 0, 0)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if ((dm >> (4 * g + j)) & 1u) e[j] = 0;
                }
                const u32 l0 = (u32)(e[0] >> 32), l1 = (u32)(e[1] >> 32), l2 = (u32)(e[2] >> 32), l3 = (u32)(e[3] >> 32);
                const u64 a = ((u64)(u32)e[0] << l1) | (u32)e[1];        // <= 64 bits
                const u64 c = ((u64)(u32)e[2] << l3) | (u32)e[3];
                L = l0 + l1 + l2 + l3;                                   // > 64 (rare): re-encoded as two halves at merge time
                grp[it][g] = (a << ((l2 + l3) & 63u)) | c;
            } else {
                const u32 *lut32 = (const u32 *)sh.lut;
                u32 e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = lut32[(wds[g] >> (8 * j)) & 0xFFu];
                if (__builtin_expect(dm != 0, 0)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if ((dm >> (4 * g + j)) & 1u) e[j] = 0;
                }
                const u32 l0 = e[0] >> 16, l1 = e[1] >> 16, l2 = e[2] >> 16, l3 = e[3] >> 16;
                const u32 a = ((e[0] & 0xFFFFu) << l1) | (e[1] & 0xFFFFu);
                const u32 c = ((e[2] & 0xFFFFu) << l3) | (e[3] & 0xFFFFu);
                L = l0 + l1 + l2 + l3;
                grp[it][g] = ((u64)a << (l2 + l3)) | c;
            }
            packed |= L << (8 * g);
            tot += L;
        }
        glen[it] = packed;
        itot[it] = tot;
    }
    {
        u32 absent = 0;
#pragma unroll
        for (int it = 0; it < E3_ITEMS; ++it) { absent |= itot[it] >> 12; itot[it] &= 0xFFFu; }
        if (absent) set_error_over(blk.err, SHAFA_FILE_UNRECOGNIZABLE, SHAFA_LACK_OF_MEMORY);     // data symbol without a code (output undefined, in bounds)
    }
    u32 incl[E3_ITEMS];
#pragma unroll
    for (int it = 0; it < E3_ITEMS; ++it) incl[it] = dpp_scan_add(itot[it]);
    if (lane == 63) {
#pragma unroll
        for (int it = 0; it < E3_ITEMS; ++it) sh.wtot[it * 4 + wv] = incl[it];
    }
    __syncthreads();                                                                           // 2
    u32 ioff[E3_ITEMS];
    u32 run = 0;
#pragma unroll
    for (int it = 0; it < E3_ITEMS; ++it) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w == wv) ioff[it] = run + incl[it] - itot[it];
            run += sh.wtot[it * 4 + w];
        }
    }
    const u32 T = run;                                 // tile bit total
    const bool last = (tile == blk.n_tiles - 1);
    const u32 s = (u32)B & 63;                         // bit phase of the tile inside its first output u64
#pragma unroll
    for (int it = 0; it < E3_ITEMS; ++it) ioff[it] += s;
    const u64 g64 = B >> 6;                            // first output u64 owned by the tile
    const u64 E = B + T;
    const u32 nfull = (u32)((E >> 6) - g64);           // whole output words owned
    const u64 total_bytes = (E + 7) >> 3;              // block size when this is the last tile
    const u32 ntail = last ? (u32)(total_bytes - 8 * (E >> 6)) : 0u;     // 0..7 bytes of a final partial word
    const u32 nwords = nfull + (ntail ? 1u : 0u);

    if (tid == 255 && s) {                             // lead bits: the last s bits before the tile
        u64 acc = 0;
        u32 got = 0;
        const u8 *pb = (const u8 *)sh.prev;
        for (int p = 63; p >= 0 && got < s; --p) {     // >= 1 bit per symbol: 63 symbols suffice
            u64 code;
            u32 len;
            if (LUT64) { const u64 x = sh.lut[pb[p]]; code = (u32)x; len = (u32)(x >> 32) & 63u; }
            else { const u32 x = ((const u32 *)sh.lut)[pb[p]]; code = x & 0xFFFFu; len = (x >> 16) & 31u; }
            acc |= (got < 64 ? code << got : 0ull);
            got += len;
        }
        acc &= (~0ull) >> (64 - s);
        atomicOr((unsigned long long *)&sh.stage[0], (unsigned long long)(acc << (64 - s)));
    }

    // ---- merge into the window(s), store --------------------------------------------------------------------
    u64 *out64 = (u64 *)blk.out + g64;
    const u64 cap64 = blk.out_cap >> 3;
    for (u32 r0 = 0; r0 < (nwords ? nwords : 1u); r0 += E3_SW64) {
        if (r0) {                                      // tile expands past one window (codes near 16 bits)
            __syncthreads();
            for (int i = tid; i < E3_SW64 + 2; i += E3_THREADS) sh.stage[i] = 0;
            __syncthreads();
        }
#pragma unroll
        for (int it = 0; it < E3_ITEMS; ++it) {
            u32 q = ioff[it];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32 L = (glen[it] >> (8 * g)) & 0xFFu;
                // `bits` (right-aligned, nb <= 64 of them) ORed into the window at bit offset qq: at most two 64-bit words
                auto put = [&](u64 bits, u32 nb, u32 qq) {
                    const u64 Gl = bits << ((64 - nb) & 63);            // left-aligned (nb == 0: the group is 0)
                    const u32 sft = qq & 63, i = (qq >> 6) - r0;        // i == -1: only the low part is in this window
                    const u64 hi = Gl >> sft;
                    const u64 lo = (Gl << 1) << (63 - sft);             // == Gl << (64 - sft), 0 when sft == 0
                    if (i < (u32)E3_SW64) atomicOr((unsigned long long *)&sh.stage[i], (unsigned long long)hi);
                    if (lo && i + 1 < (u32)E3_SW64) atomicOr((unsigned long long *)&sh.stage[i + 1], (unsigned long long)lo);
                };
                if (LUT64 && __builtin_expect(L > 64, 0)) {
                    // four codes of more than 64 bits together (rare): look the symbols up again, two halves of <= 64 bits
                    const u32 wd = g == 0 ? cur[it].x : g == 1 ? cur[it].y : g == 2 ? cur[it].z : cur[it].w;
                    u64 e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        e[j] = sh.lut[(wd >> (8 * j)) & 0xFFu];
                        if ((dropmask[it] >> (4 * g + j)) & 1u) e[j] = 0;
                    }
                    const u32 l0 = (u32)(e[0] >> 32), l1 = (u32)(e[1] >> 32), l2 = (u32)(e[2] >> 32), l3 = (u32)(e[3] >> 32);
                    put(((u64)(u32)e[0] << (l1 & 63u)) | (u32)e[1], l0 + l1, q);
                    put(((u64)(u32)e[2] << (l3 & 63u)) | (u32)e[3], l2 + l3, q + l0 + l1);
                } else {
                    put(grp[it][g], L, q);
                }
                q += L;
            }
        }
        __syncthreads();                                                                       // 3
        // straight copy of the owned words of this window: two u64 (16 bytes) per lane
        const u32 wend = (nfull < r0 + E3_SW64) ? nfull : r0 + E3_SW64;
        if (g64 + wend > cap64) {
            if (tid == 0) set_error(blk.err, SHAFA_LACK_OF_MEMORY);
        } else {
            for (u32 i = r0 + 2 * (u32)tid; i < wend; i += 2 * E3_THREADS) {
                const u64 a = bswap64(sh.stage[i - r0]);
                if (i + 1 < wend) {
                    const u64 c = bswap64(sh.stage[i - r0 + 1]);
                    gstore<uint4>(out64 + i, make_uint4((u32)a, (u32)(a >> 32), (u32)c, (u32)(c >> 32)));
                } else gstore<u64>(out64 + i, a);
            }
        }
        if (ntail && tid == 0 && nfull >= r0 && nfull < r0 + E3_SW64) {     // the block's final partial word
            const u64 w = sh.stage[nfull - r0];
            if (total_bytes <= blk.out_cap) {
                for (u32 q = 0; q < ntail; ++q) gstore<u8>(blk.out + 8 * (g64 + nfull) + q, (u8)(w >> (56 - 8 * q)));
            } else set_error(blk.err, SHAFA_LACK_OF_MEMORY);
        }
    }
}

}  // namespace

// launched from sfenc_launch (sf_encode.hip); d_tile_bits: u32 per tile, d_tile_off: u64 per tile;
// lut64: the blocks' tables are 64-bit (codes of 17..32 bits somewhere in the launch)
void sfenc3_launch(hipStream_t st, const EncBlk *dblk, int count, u32 max_tiles, u32 *d_tile_bits, u64 *d_tile_off, bool lut64)
{
    const dim3 grid(max_tiles, (u32)count), grid_c((max_tiles + E3_CTPW - 1) / E3_CTPW, (u32)count);
    if (lut64) hipLaunchKernelGGL(sfe3_count<true>, grid_c, dim3(E3_THREADS), 0, st, dblk, d_tile_bits);
    else hipLaunchKernelGGL(sfe3_count<false>, grid_c, dim3(E3_THREADS), 0, st, dblk, d_tile_bits);
    hipLaunchKernelGGL(sfe3_scan, dim3((u32)count), dim3(E3_THREADS), 0, st, dblk, (const u32 *)d_tile_bits, d_tile_off);
    if (lut64) hipLaunchKernelGGL(sfe3_pack<true>, grid, dim3(E3_THREADS), 0, st, dblk, (const u64 *)d_tile_off);
    else hipLaunchKernelGGL(sfe3_pack<false>, grid, dim3(E3_THREADS), 0, st, dblk, (const u64 *)d_tile_off);
}
