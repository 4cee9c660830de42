// sfd_generic.hpp — the byte-map kernels: exact decode of ANY prefix code (incomplete tables, codes of up to 255 bits)
// Part of sf_decode.hip's translation unit (included there; not compiled on its own).
#pragma once

// sfd_sync -> sfd_tiles -> sfd_count -> (sfd_offsets) -> sfd_write.  What a hand-made .cod gets; Module T's complete codes
// of <= 32 bits take the packed kernels of sfd_dp.hpp and the speculative entries / staged symbol pass of sf_decode.hip.

namespace {

// ------------------------------------------------------------------------------------------------
// sfd_sync: chunk maps (global, [tile][d][chunk]) and tile maps ([tile][d])
// dynamic LDS: data[DATA_WORDS*9/8+8] u32 | ring[R*256] u8 | lut[2^K] u16 | wfn[4*R] u8
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DEC_THREADS) void sfd_sync(const DecBlk *__restrict__ blks, u32 R, u32 l2cap,
                                                        u8 *__restrict__ chunkfn, u8 *__restrict__ tilefn)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const DecBlk blk = blks[blockIdx.y];
    const u32 tile = blockIdx.x;
    if (tile >= blk.n_tiles) return;
    u32 *data = (u32 *)smem;
    u8 *ring = smem + (size_t)(DATA_WORDS + DATA_WORDS / 8 + 8) * 4;
    u16 *lut = (u16 *)(ring + (size_t)R * DEC_THREADS);
    u8 *wfn = (u8 *)(lut + (1u << LUT_MAXK) + l2cap);
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 lmax = blk.lmax, K = blk.K, Rm = R - 1;

    load_tile(data, blk, tile);
    load_lut(lut, blk);
    __syncthreads();

    // backward DP over the chunk's bit positions: ring[(p mod R)][tid] = exit(p)
    const u32 cbase = tid * CH_BITS;
    for (int p = CH_BITS - 1; p >= 0; --p) {
        const Code c = code_at(data, lut, blk.trie, K, cbase + (u32)p);
        const u32 nx = (u32)p + c.len;
        u32 x;
        if (nx >= (u32)CH_BITS) x = nx - CH_BITS;
        else x = ring[((nx & Rm) << 8) + tid];
        ring[(((u32)p & Rm) << 8) + tid] = (u8)x;
    }
    __syncthreads();

    // chunk maps to global: rows d < lmax, 256 bytes each (coalesced)
    u8 *cf = chunkfn + ((size_t)(blk.tile_base + tile) * R << 8);
    for (u32 i = tid; i < lmax * (DEC_THREADS / 4); i += DEC_THREADS)
        ((u32 *)cf)[i] = ((const u32 *)ring)[i];

    // wave maps: lane d chases entry d through the wave's 64 chunks (d < lmax, in groups of 64)
    for (u32 d0 = 0; d0 < lmax; d0 += 64) {
        const u32 d = d0 + lane;
        u32 v = d < lmax ? d : 0;
        for (u32 c = 0; c < 64; ++c) v = ring[(v << 8) + wv * 64 + c];
        if (d < lmax) wfn[wv * R + d] = (u8)v;
    }
    __syncthreads();
    // tile map = wave 0 then 1, 2, 3
    u8 *tf = tilefn + (size_t)(blk.tile_base + tile) * R;
    for (u32 d = tid; d < lmax; d += DEC_THREADS) {
        u32 v = d;
#pragma unroll
        for (int w = 0; w < 4; ++w) v = wfn[w * R + v];
        tf[d] = (u8)v;
    }
}

// ------------------------------------------------------------------------------------------------
// sfd_tiles: per block, chase the tile maps from entry 0 -> entry of every tile
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DEC_THREADS) void sfd_tiles(const DecBlk *__restrict__ blks, u32 R,
                                                         const u8 *__restrict__ tilefn, u8 *__restrict__ tile_entry)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];   // 256 * R bytes of maps + 256 entries
    __shared__ u8 segmap[8 * 32], segent[8];
    if (dp_skipped_early(blks + blockIdx.x)) return;    // the block's speculative entries verified
    const DecBlk blk = blks[blockIdx.x];
    u8 *maps = smem;
    u8 *ent = smem + (size_t)R * DEC_THREADS;
    const u32 tid = threadIdx.x;
    u32 v = 0;                                   // carried by thread 0
    for (u32 t0 = 0; t0 < blk.n_tiles; t0 += DEC_THREADS) {
        const u32 nt = (blk.n_tiles - t0 < (u32)DEC_THREADS) ? blk.n_tiles - t0 : (u32)DEC_THREADS;
        const u8 *src = tilefn + (size_t)(blk.tile_base + t0) * R;
        for (u32 i = tid; i < nt * R / 4; i += DEC_THREADS) ((u32 *)maps)[i] = ((const u32 *)src)[i];
        __syncthreads();
        if (R == 32) {
            // thread (sg, d) follows entry d through segment sg (32 tiles), thread 0 links the 8 segments, then one
            // thread per segment walks it from its real entry: 32 + 8 + 32 dependent reads instead of 256
            const u32 sg = tid >> 5, d = tid & 31u, lo = sg * 32 < nt ? sg * 32 : nt, hi = lo + 32 < nt ? lo + 32 : nt;
            u32 x = d;
            for (u32 t = lo; t < hi; ++t) x = maps[t * R + (x & 31u)];
            segmap[sg * 32 + d] = (u8)x;
            __syncthreads();
            if (tid == 0) {
                for (u32 q = 0; q < 8; ++q) { segent[q] = (u8)v; v = segmap[q * 32 + (v & 31u)]; }
            }
            __syncthreads();
            if (d == 0) {
                u32 y = segent[sg];
                for (u32 t = lo; t < hi; ++t) { ent[t] = (u8)y; y = maps[t * R + (y & 31u)]; }
            }
        } else if (tid == 0) {
            for (u32 t = 0; t < nt; ++t) { ent[t] = (u8)v; v = maps[t * R + v]; }
        }
        __syncthreads();
        if (tid < nt) tile_entry[blk.tile_base + t0 + tid] = ent[tid];
        __syncthreads();
    }
}

// decode the chunk's own symbols starting at bit `entry`; Sink(sym, ok) per symbol.  Returns the count.
// Only codes that end inside the stream (tile-local bit `limit`) are symbols: zero padding past the
// last byte must not be counted, or a truncated stream would go unnoticed.
template <typename Sink>
__device__ __forceinline__ u32 decode_chunk(const u32 *data, const u16 *lut, const DecBlk &blk, u32 cbase,
                                            u32 entry, u32 limit, u32 max_syms, Sink sink)
{
    // 64-bit bit buffer, next stream bit at the MSB, >= 32 valid bits before every look-up
    u32 p = entry, cnt = 0;
    u32 pos = cbase + p;
    u32 wnext = (pos >> 5) + 2;
    u64 buf = ((((u64)data[widx(pos >> 5)]) << 32) | data[widx((pos >> 5) + 1)]) << (pos & 31);
    int avail = 64 - (int)(pos & 31);
    const u32 K = blk.K;
    while (p < (u32)CH_BITS && cnt < max_syms) {
        const u32 win = (u32)(buf >> 32);
        u32 e = lut[win >> (32 - K)];
        if (e & 0x8000u) {                              // codes of K+1..K+8 bits: one more table
            const u32 nb = ((e >> 12) & 7u) + 1;
            e = lut[(1u << LUT_MAXK) + (e & 0xFFFu) + ((win << K) >> (32 - nb))];
        }
        u32 len = e >> 8, sym = e & 0xFF;
        bool ok = true;
        if (__builtin_expect(e == 0, 0)) {              // longer code or missing branch: trie walk
            const Code c = code_at(data, lut, blk.trie, K, cbase + p);
            len = c.len; sym = c.sym; ok = c.ok;
        }
        if (cbase + p + len > limit) break;
        sink(sym, ok);
        p += len;
        ++cnt;
        if (__builtin_expect(len > 31, 0)) {            // rebuild the buffer after a very long code
            pos = cbase + p;
            wnext = (pos >> 5) + 2;
            buf = ((((u64)data[widx(pos >> 5)]) << 32) | data[widx((pos >> 5) + 1)]) << (pos & 31);
            avail = 64 - (int)(pos & 31);
        } else {
            buf <<= len;
            avail -= (int)len;
            if (avail < 32) {
                buf |= (u64)data[widx(wnext)] << (32 - avail);
                avail += 32;
                ++wnext;
            }
        }
    }
    return cnt;
}

// tile-local bit index of the end of the stream (clamped to the staged window)
__device__ __forceinline__ u32 tile_bit_limit(const DecBlk &blk, u32 tile)
{
    const u64 start = (u64)tile * DTILE;
    const u64 left = blk.in_n > start ? blk.in_n - start : 0;
    const u64 cap = (u64)DTILE + HALO_WORDS * 4;
    return (u32)((left < cap ? left : cap) * 8);
}

// ------------------------------------------------------------------------------------------------
// sfd_count: chunk entries (from the chunk maps) + symbols per chunk and per tile
// dynamic LDS: data | maps[R*256] | lut | went[4] wfn[4*R] | ent[256]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DEC_THREADS) void sfd_count(const DecBlk *__restrict__ blks, u32 R, u32 l2cap,
                                                         const u8 *__restrict__ chunkfn,
                                                         const u8 *__restrict__ tile_entry,
                                                         u8 *__restrict__ chunk_entry, u16 *__restrict__ chunk_cnt,
                                                         u32 *__restrict__ tile_cnt)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const DecBlk blk = blks[blockIdx.y];
    const u32 tile = blockIdx.x;
    if (tile >= blk.n_tiles) return;
    u32 *data = (u32 *)smem;
    u8 *maps = smem + (size_t)(DATA_WORDS + DATA_WORDS / 8 + 8) * 4;      // R * 256 u8
    u16 *lut = (u16 *)(maps + (size_t)R * DEC_THREADS);
    u8 *wfn = (u8 *)(lut + (1u << LUT_MAXK) + l2cap);
    u8 *ent = wfn + 4 * R;
    u32 *wsum = (u32 *)(ent + DEC_THREADS);
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 lmax = blk.lmax;
    const size_t gt = (size_t)blk.tile_base + tile;

    load_tile(data, blk, tile);
    load_lut(lut, blk);
    {
        const u8 *cf = chunkfn + (gt * R << 8);
        for (u32 i = tid; i < lmax * (DEC_THREADS / 4); i += DEC_THREADS) ((u32 *)maps)[i] = ((const u32 *)cf)[i];
    }
    __syncthreads();

    // wave maps (as in sfd_sync), then wave entries from the tile entry, then every chunk's entry
    {
        for (u32 d0 = 0; d0 < lmax; d0 += 64) {
            const u32 d = d0 + lane;
            u32 v = d < lmax ? d : 0;
            for (u32 c = 0; c < 64; ++c) v = maps[(v << 8) + wv * 64 + c];
            if (d < lmax) wfn[wv * R + d] = (u8)v;
        }
        __syncthreads();
        if (lane == 0) {
            u32 v = tile_entry[gt];
            for (u32 w = 0; w < wv; ++w) v = wfn[w * R + v];
            for (u32 c = 0; c < 64; ++c) {
                ent[wv * 64 + c] = (u8)v;
                v = maps[(v << 8) + wv * 64 + c];
            }
        }
    }
    __syncthreads();

    const u32 entry = ent[tid];
    const u32 cnt = decode_chunk(data, lut, blk, tid * CH_BITS, entry, tile_bit_limit(blk, tile), 0xFFFFFFFFu, [](u32, bool) {});
    chunk_entry[gt * DEC_THREADS + tid] = (u8)entry;
    chunk_cnt[gt * DEC_THREADS + tid] = (u16)cnt;
    const u32 tot = wave_reduce_add<u32>(cnt);
    if (lane == 0) wsum[wv] = tot;
    __syncthreads();
    if (tid == 0) tile_cnt[gt] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ------------------------------------------------------------------------------------------------
// sfd_write: decode every chunk from its entry and store the symbols (index < n_sym only)
// dynamic LDS: data | lut | wsum
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DEC_THREADS) void sfd_write(const DecBlk *__restrict__ blks, u32 l2cap,
                                                         const u8 *__restrict__ chunk_entry,
                                                         const u16 *__restrict__ chunk_cnt,
                                                         const u64 *__restrict__ tile_off)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const DecBlk blk = blks[blockIdx.y];
    const u32 tile = blockIdx.x;
    if (tile >= blk.n_tiles) return;
    u32 *data = (u32 *)smem;
    u16 *lut = (u16 *)(smem + (size_t)(DATA_WORDS + DATA_WORDS / 8 + 8) * 4);
    u32 *wsum = (u32 *)(lut + (1u << LUT_MAXK) + l2cap);
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t gt = (size_t)blk.tile_base + tile;
    const u64 toff = tile_off[gt];
    if (toff >= blk.n_sym) return;                      // the whole tile is padding / past the end

    load_tile(data, blk, tile);
    load_lut(lut, blk);
    const u32 entry = chunk_entry[gt * DEC_THREADS + tid];
    const u32 cnt = chunk_cnt[gt * DEC_THREADS + tid];
    const u32 incl = wave_incl_scan_add<u32>(cnt);
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    u32 base = 0;
    for (u32 w = 0; w < wv; ++w) base += wsum[w];
    const u64 first = toff + base + incl - cnt;         // global index of this chunk's first symbol

    // symbols of this chunk that fall inside the block (the stream's padding decodes to extra ones)
    const u64 nsym = blk.n_sym;
    const u32 want = first >= nsym ? 0u : (nsym - first < (u64)cnt ? (u32)(nsym - first) : cnt);
    u8 *op = blk.out + first;
    u32 acc = 0, na = 0;
    bool bad = false;
    decode_chunk(data, lut, blk, tid * CH_BITS, entry, tile_bit_limit(blk, tile), want, [&](u32 sym, bool ok) {
        bad |= !ok;
        acc |= sym << (8 * na);
        if (++na == 4) {                                // four symbols per store (any byte alignment)
            gstore<u32>(op, acc);
            op += 4;
            acc = 0;
            na = 0;
        }
    });
    for (u32 q = 0; q < na; ++q) gstore<u8>(op + q, (u8)(acc >> (8 * q)));
    if (bad) set_error(blk.err, SHAFA_FILE_UNRECOGNIZABLE);
}

}  // namespace
