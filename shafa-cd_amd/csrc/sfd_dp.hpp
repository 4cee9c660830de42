// sfd_dp.hpp — the exact entries of complete codes of <= 32 bits: packed backward DP, map chase, counting automaton
// Part of sf_decode.hip's translation unit (included there; not compiled on its own).
#pragma once

// sfd_tables -> sfd_sync16 / sfd_sync32 -> sfd_tiles16 / sfd_tiles -> sfd_countfsm / sfd_countfsm32: what a block takes whose
// code does not re-synchronise (uniform-like data) or whose speculative entries did not verify (sf_decode.hip: sfd_scan).

namespace {

// ================================================================================================
// Packed path (every block of the launch has Lmax <= 16): a chunk map is 16 nibbles in one u64,
// the DP ring lives in registers, code lengths come from a byte LUT of up to 13 index bits.
// ================================================================================================

// code length at tile-local bit p by trie walk (codes longer than the LUT index, or missing branch: 1)
__device__ __noinline__ u32 slow_len(const u32 *data, const u32 *trie, u32 p)
{
    u32 node = 0, q = p, depth = 0;
    for (;;) {
        const u32 bit = (data[widx(q >> 5)] >> (31 - (q & 31))) & 1u;
        const u32 nx = trie[2 * node + bit];
        ++q; ++depth;
        if (nx == 0xFFFFFFFFu) return 1;
        if (nx & 0x80000000u) return depth;
        node = nx;
        if (depth >= 255) return 1;
    }
}

__device__ __forceinline__ u32 nib(u64 m, u32 v) { return (u32)(m >> (4 * v)) & 15u; }

// ring = (ring << 4) | nibble (sh4 >> 2) of ring; only bits 2..5 of sh4 matter.  v_bfi merges the and + or.
__device__ __forceinline__ u64 ring_push(u64 ring, u32 sh4)
{
    const u32 x = (u32)(ring >> (sh4 & 60u));
    const u64 up = ring << 4;
    u32 lo;
    asm("v_bfi_b32 %0, 15, %1, %2" : "=v"(lo) : "v"(x), "v"((u32)up));
    return (up & 0xFFFFFFFF00000000ull) | lo;
}

// packed 16-entry maps: (a then b)[d] = b[a[d]]
__device__ __forceinline__ u64 map_compose(u64 a, u64 b)
{
    u64 r = 0;
#pragma unroll
    for (int d = 0; d < 16; ++d) r |= (u64)nib(b, nib(a, (u32)d)) << (4 * d);
    return r;
}

// Quarter chase over one wave's 64 chunk maps (LDS, wave-private slice `cmw`): lane (q = lane >> 4, d = lane & 15)
// follows entry d through the 16 chunks of quarter q.  Returns Q_q[d]; when HIST, *hist gets the entry seen at each
// of the 16 chunks (nibble c).  16 dependent LDS reads instead of 64, all 64 lanes busy.
template <bool HIST>
__device__ __forceinline__ u32 quarter_chase(const u64 *cmw, u64 *hist)
{
    const u32 lane = lane_id();
    const u64 *src = cmw + (lane >> 4) * 16;
    u32 v = lane & 15u, hlo = 0, hhi = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (HIST) { if (c < 8) hlo |= v << (4 * c); else hhi |= v << (4 * (c - 8)); }
        v = nib(src[c], v);
    }
    if (HIST) *hist = ((u64)hhi << 32) | hlo;
    return v;
}
// wave map from the four quarter maps held one value per lane (lanes d < 16 return W[d])
__device__ __forceinline__ u32 wave_map_of(u32 qv)
{
    u32 v = lane_id() & 15u;
#pragma unroll
    for (int q = 0; q < 4; ++q) v = __shfl(qv, q * 16 + (int)v, 64);
    return v;
}

// code of 13..16 bits at the head of `win32` (next stream bit at the MSB): binary search of its first 12 bits in
// the sorted prefix list, then 4 more bits index the group.  Returns sym | len << 8, 0 when there is none.
__device__ __forceinline__ u32 long_code(const u16 *lt, u32 win32)
{
    const u32 n = lt[0], key = win32 >> 20;
    const u16 *pfx = lt + 8, *ent = lt + 8 + LONG_PFX;
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (pfx[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < n && pfx[lo] == key) ? ent[lo * 16 + ((win32 >> 16) & 15u)] : 0u;
}

// code of 13..32 bits at the head of `win32`: binary search of its first 12 bits, then a walk of the sub-trie.
// Returns sym | len << 8, 0 when there is none.
__device__ __forceinline__ u32 long_code32(const u16 *lt, u32 win32)
{
    const u32 n = lt[0], key = win32 >> 20;
    const u16 *pfx = lt + 8, *root = lt + 8 + LONG_PFX, *nodes = lt + 8 + 2 * LONG_PFX;
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (pfx[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (lo >= n || pfx[lo] != key) return 0u;
    u32 node = root[lo];
    for (u32 depth = (u32)SYM3_MAXK; depth < 32; ++depth) {
        const u32 c = nodes[2 * node + ((win32 >> (31 - depth)) & 1u)];
        if (c & 0x8000u) return (c & 0xFFu) | ((depth + 1) << 8);
        node = c;
    }
    return 0u;
}

// copy a device table (16-byte aligned, padded to 16 bytes in the workspace) into LDS
__device__ __forceinline__ void fill_lds16(void *dst, const void *src, u32 bytes)
{
    for (u32 i = threadIdx.x; i < (bytes + 15) / 16; i += blockDim.x)
        ((uint4 *)dst)[i] = gload<uint4>((const uint4 *)src + i);
}

// sfd_tables: one workgroup per block expands the host tables (complete codes, Lmax <= 13) into
//   pairlut: the two-positions-per-lookup length table of the DP, and
//   cnt3 / sym3: up to three whole codes per K1-bit window for the symbol passes.
__global__ __launch_bounds__(DEC_THREADS) void sfd_tables(const DecBlk *__restrict__ blks)
{
    const DecBlk blk = blks[blockIdx.x];
    if (!blk.n_tiles) return;
    // gridDim.y workgroups share a block's tables (an entry is a chain of up to fifteen dependent look-ups in global
    // memory: one workgroup per block took 45 us for 12-bit tables, as long as the symbol pass of a small launch)
    const u32 T0 = threadIdx.x + blockIdx.y * DEC_THREADS, TS = DEC_THREADS * gridDim.y;
    const u32 K1 = blk.K1, mask = (1u << K1) - 1;
    for (u32 i = T0; blk.pairlut && i < (2u << K1); i += TS)
        blk.pairlut[i] = (u8)((blk.lenlut[i >> 1] - 1u) | ((blk.lenlut[i & mask] - 1u) << 4));
    const u32 K3 = sym3_window(K1);
    const u32 KW = blk.KW, maskw = (1u << KW) - 1;               // the counting window may be wider than the longest code, or
    for (u32 i = T0; i <= maskw; i += TS) {                      // one bit narrower than the 13-bit table (host: sfdec_launch)
        u32 pos = 0, n = 0, l0 = 0;
        for (; n < 7; ++n) {
            const u32 wv = (i << pos) & maskw;                   // window shifted left, zero filled
            const u32 L = blk.lut13[KW >= K1 ? wv >> (KW - K1) : wv << (K1 - KW)] >> 8;
            if (L == 0 || L > KW - pos) break;                   // longer than the window / would use bits outside it
            if (n == 0) l0 = L;
            pos += L;
        }
        const u32 j = __builtin_bitreverse32(i) >> (32 - KW);    // sfd_scan reads its windows LSB first
        ((u8 *)blk.cnt3)[j] = (u8)(pos | (n << 5));              // bits in the low five: the sum of a fetch's entries is the next look-up's shift
        ((u8 *)blk.cnt3)[(1u << KW) + j] = (u8)l0;
    }
    for (u32 i = T0; i < (1u << K3); i += TS) {   // K3-bit window; n = 0: first code is longer
        u32 pos = 0, n = 0, syms = 0;
        for (; n < 3; ++n) {
            const u32 e = blk.lut13[K3 >= K1 ? (((i << pos) & ((1u << K3) - 1u)) >> (K3 - K1)) : (((i << (K1 - K3)) << pos) & mask)];
            const u32 L = e >> 8;
            if (L == 0 || L > K3 - pos) break;
            syms |= (e & 0xFFu) << (8 * n);
            pos += L;
        }
        blk.sym3[__builtin_bitreverse32(i) >> (32 - K3)] = syms | (pos << 24) | (n << 30);    // sfd_wstage reads its windows LSB first
    }
    // counting automaton: state = internal trie node (0 = root = between two codes); consuming a nibble (or a bit)
    // moves to the next state and completes 0..4 codes.  next state is stored as the byte offset of its row.
    for (u32 i = T0; i < blk.n_states * 16; i += TS) {
        u32 node = i >> 4, done = 0;
        for (int b = 3; b >= 0; --b) {
            const u32 c = blk.trie[2 * node + ((i >> b) & 1u)];
            if (c & 0x80000000u) { ++done; node = 0; } else node = c;
        }
        blk.fsm4[i] = (node * 64u) | (done << 16);
    }
    for (u32 i = T0; i < blk.n_states * 2; i += TS) {
        const u32 c = blk.trie[i];
        blk.fsm1[i] = (c & 0x80000000u) ? (1u << 16) : (c * 64u);
    }
}

// sfd_sync16: dynamic LDS: data | lenlut[2^13] u8 (PAIR: pairlut[2^14]) | cmap[256] u64 | wmb[64] u8
// K1T: the launch's common table width when every block has it (12 or 13: window offsets become constants, a
// pair window that lies inside one stream word is then a single v_bfe), 0 = per block at run time.
template <bool PAIR, bool LONG, int K1T = 0>
__global__ __launch_bounds__(DEC_THREADS) void sfd_sync16(const DecBlk *__restrict__ blks,
                                                          u64 *__restrict__ chunkfn, u64 *__restrict__ tilefn, u32 tpw)
{
    // static LDS: constant addresses fold into the ds_read offset field (no per-lookup address add)
    // (the chunk maps of a wave live in that wave's own rows of the stream frame, which it is done with by then: with a
    // separate 2 KiB for them the pair-table form is 27.8 KB, just above the 26 KiB that let six workgroups share a CU)
    __shared__ __attribute__((aligned(16))) u8 smem[LDS_DATA + (PAIR ? 2 : 1) * (1 << LEN_MAXK) + 64 + (LONG ? LONG_BYTES : 0)];
    if (dp_skipped_early(blks + blockIdx.y)) return;
    const DecBlk blk = blks[blockIdx.y];
    if (blockIdx.x * tpw >= blk.n_tiles) return;
    u32 *data = (u32 *)smem;
    u8 *lenlut = smem + LDS_DATA;
    u8 *wmb = lenlut + (PAIR ? 2u : 1u) * (1u << LEN_MAXK);
    const u16 *lt = (const u16 *)(wmb + 64);
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // wave wv's rows are frame words [576 wv, 576 wv + 576); the wave before reads only the first of them (its last
    // lane's look-ahead word), at the start of its own pass
    u64 *cmapw = (u64 *)(data + 576u * wv + 16u);
    const u32 K1 = K1T ? (u32)K1T : blk.K1;

    if (LONG) {                                        // blocks of the launch without long codes: empty list
        if (blk.longtab) fill_lds16((void *)lt, blk.longtab, LONG_BYTES);
        else if (tid == 0) *(u16 *)lt = 0;
    }
    fill_lds16(lenlut, PAIR ? (const void *)blk.pairlut : (const void *)blk.lenlut, PAIR ? (2u << K1) : (1u << K1));
    const u32 tile_end = (blockIdx.x + 1) * tpw < blk.n_tiles ? (blockIdx.x + 1) * tpw : blk.n_tiles;
    for (u32 tile = blockIdx.x * tpw; tile < tile_end; ++tile) {      // one table load serves tpw tiles
    __syncthreads();                                   // previous tile's LDS reads are done
    load_tile(data, blk, tile);
    __syncthreads();

    // backward DP; nibble j of `ring` = exit(p + 1 + j).  Positions 256..271 (the next chunk's first
    // bits) have exit = their offset, which is the initial ring.
    u64 ring = 0xFEDCBA9876543210ull;
    const u32 cw = tid * (CH_BITS / 32);
    u32 w1 = data[widx(cw + 8)];
    if (PAIR) {
        const u32 sh = 31 - K1;                        // K1+1-bit window: positions r and r+1
        for (int wi = 7; wi >= 0; --wi) {
            const u32 w0 = data[widx(cw + wi)];
            u32 e[16];
#pragma unroll
            for (int q = 15; q >= 0; --q) {
                if (K1T && 2 * q + K1T + 1 <= 32) {              // the K1+1-bit window lies inside w0: one bit-field extract
                    e[q] = lenlut[__builtin_amdgcn_ubfe(w0, 32 - 2 * q - (K1T + 1), K1T + 1)];
                } else {
                    const u32 win = q ? __builtin_amdgcn_alignbit(w0, w1, 32 - 2 * q) : w0;
                    e[q] = lenlut[win >> sh];
                }
            }
#pragma unroll
            for (int q = 15; q >= 0; --q) {
                ring = ring_push(ring, e[q] >> 2);            // position 2q+1 (shift = 4 * high nibble)
                ring = ring_push(ring, e[q] << 2);            // position 2q   (shift = 4 * low nibble)
            }
            w1 = w0;
        }
    } else {
        const u32 sh = 32 - K1;
        for (int wi = 7; wi >= 0; --wi) {
            const u32 w0 = data[widx(cw + wi)];
            u32 len[32];
#pragma unroll
            for (int r = 31; r >= 0; --r) {
                const u32 win = r ? __builtin_amdgcn_alignbit(w0, w1, 32 - r) : w0;
                len[r] = lenlut[win >> sh];
            }
#pragma unroll
            for (int r = 31; r >= 0; --r) {
                u32 l = len[r];
                if (__builtin_expect(l == 0, 0)) {
                    if (LONG) {                        // 14..16 bits: resolved from LDS (complete code: always found)
                        const u32 win = r ? __builtin_amdgcn_alignbit(w0, w1, 32 - r) : w0;
                        l = long_code(lt, win) >> 8;
                        l = l ? l : 1u;
                    } else {
                        l = slow_len(data, blk.trie, (cw + wi) * 32 + r);
                    }
                }
                const u32 x = nib(ring, l - 1);
                ring = (ring << 4) | x;
            }
            w1 = w0;
        }
    }
    // ring nibble d = exit(d) = this chunk's map
    chunkfn[((size_t)blk.tile_base + tile) * DEC_THREADS + tid] = ring;
    cmapw[lane] = ring;                                // wave-private: no barrier needed before the chase
    {
        const u32 w = wave_map_of(quarter_chase<false>(cmapw, nullptr));
        if (lane < 16) wmb[wv * 16 + lane] = (u8)w;
    }
    __syncthreads();
    if (tid < 16) {
        u32 v = tid;
#pragma unroll
        for (int w = 0; w < 4; ++w) v = wmb[w * 16 + v];
        u64 m = (u64)v << (4 * tid);
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) m |= __shfl_xor(m, d, 64);
        if (tid == 0) tilefn[(size_t)blk.tile_base + tile] = m;
    }
    }
}

// ================================================================================================
// Complete codes with 16 < Lmax <= 32 (rare bytes of real files at -b M).  Entry offsets reach 31, so chunk maps
// are 32 bytes (the byte-map plumbing of the generic path: sfd_tiles, [tile][d][chunk] layout), but the heavy loops
// are the fast ones: the DP keeps the low nibble of every exit in the register ring plus one bit per position for
// "exit >= 16"; a code longer than 16 bits at some bit position (rare: its probability) is followed forward to
// the end of the chunk instead of being looked up in the ring.
// ================================================================================================
__device__ __forceinline__ u32 tile_bit_limit(const DecBlk &blk, u32 tile);

__device__ __forceinline__ u32 win32_at(const u32 *data, u32 p)
{
    const u32 w = p >> 5, r = p & 31;
    const u32 w0 = data[widx(w)], w1 = data[widx(w + 1)];
    return r ? __builtin_amdgcn_alignbit(w0, w1, 32 - r) : w0;
}

// length of a code longer than 13 bits: `esc` = 128 + k from lenlut32 names the trie node reached after 13 bits
__device__ __forceinline__ u32 sync32_long_len(const u16 *lt, u32 esc, u32 win)
{
    const u16 *nodes = lt + 8 + 2 * LONG_PFX, *root13 = nodes + 512;
    u32 node = root13[esc - 128u];
    for (u32 depth = 13; depth < 32; ++depth) {
        const u32 c = nodes[2 * node + ((win >> (31 - depth)) & 1u)];
        if (c & 0x8000u) return depth + 1;
        node = c;
    }
    return 1u;
}

// static LDS: data | maps[32][256] u8 | lenlut[2^13] u8 | long32 | wfn[4][32] u8
__global__ __launch_bounds__(DEC_THREADS) void sfd_sync32(const DecBlk *__restrict__ blks, u8 *__restrict__ chunkfn,
                                                          u8 *__restrict__ tilefn, u32 tpw)
{
    constexpr u32 R = 32;
    __shared__ __attribute__((aligned(16))) u8 smem[LDS_DATA + R * DEC_THREADS + (1 << LEN_MAXK) + LONG32_BYTES + 4 * R + 64];
    if (dp_skipped_early(blks + blockIdx.y)) return;    // the block's speculative entries verified
    const DecBlk blk = blks[blockIdx.y];
    if (blockIdx.x * tpw >= blk.n_tiles) return;
    u32 *data = (u32 *)smem;
    u8 *maps = smem + LDS_DATA;
    u8 *lenlut = maps + R * DEC_THREADS;
    const u16 *lt = (const u16 *)(lenlut + (1 << LEN_MAXK));
    u8 *wfn = (u8 *)lt + LONG32_BYTES;
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 K1 = blk.K1, lmax = blk.lmax, sh = 32 - K1;

    if (blk.long32) fill_lds16((void *)lt, blk.long32, LONG32_BYTES);
    else if (tid == 0) *(u16 *)lt = 0;
    fill_lds16(lenlut, blk.lenlut32, 1u << K1);
    const u32 tile_end = (blockIdx.x + 1) * tpw < blk.n_tiles ? (blockIdx.x + 1) * tpw : blk.n_tiles;
    for (u32 tile = blockIdx.x * tpw; tile < tile_end; ++tile) {
    __syncthreads();                                   // previous tile's LDS reads are done
    load_tile(data, blk, tile);
    __syncthreads();

    // ring nibble j / hb bit j = low nibble / "exit >= 16" of position p + 1 + j; positions 256 + j exit at j
    // ring2 nibble j = low nibble of position p + 17 + j (codes of 17..32 bits look there)
    u64 ring = 0xFEDCBA9876543210ull, ring2 = 0xFEDCBA9876543210ull, mapB = 0;
    u32 hb = 0xFFFF0000u, hbB = 0;
    const u32 cw = tid * (CH_BITS / 32);
    u32 w1 = data[widx(cw + 8)];
    for (int wi = 7; wi >= 0; --wi) {
        const u32 w0 = data[widx(cw + wi)];
        u32 len[32];
#pragma unroll
        for (int r = 31; r >= 0; --r) {
            const u32 win = r ? __builtin_amdgcn_alignbit(w0, w1, 32 - r) : w0;
            len[r] = lenlut[win >> sh];
        }
#pragma unroll
        for (int r = 31; r >= 0; --r) {
            u32 l = len[r];
            u32 xlo;
            if (__builtin_expect(l >= 128, 0)) {       // longer than 13 bits (rare): walk the sub-trie; 17..32 bits look in ring2
                l = sync32_long_len(lt, l, r ? __builtin_amdgcn_alignbit(w0, w1, 32 - r) : w0);
                xlo = l > 16 ? nib(ring2, l - 17) : nib(ring, l - 1);
            } else {
                xlo = nib(ring, l - 1);
            }
            const u32 xhi = (hb >> (l - 1)) & 1u;
            ring2 = (ring2 << 4) | (u32)(ring >> 60);
            ring = (ring << 4) | xlo;
            hb = (hb << 1) | xhi;
            if (wi == 0 && r == 16) { mapB = ring; hbB = hb; }     // exits of positions 16..31
        }
        w1 = w0;
    }
    // the 32 exits of this chunk as bytes: maps[d][tid]
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        maps[((u32)d << 8) + tid] = (u8)(nib(ring, (u32)d) | (((hb >> d) & 1u) << 4));
        maps[((u32)(d + 16) << 8) + tid] = (u8)(nib(mapB, (u32)d) | (((hbB >> d) & 1u) << 4));
    }
    __syncthreads();

    // chunk maps to global: rows d < lmax, 256 bytes each (coalesced)
    u8 *cf = chunkfn + ((size_t)(blk.tile_base + tile) * R << 8);
    for (u32 i = tid; i < lmax * (DEC_THREADS / 4); i += DEC_THREADS)
        ((u32 *)cf)[i] = ((const u32 *)maps)[i];
    // wave maps: two halves of 32 chunks, lane (h, d) follows entry d through half h; then joined
    {
        const u32 h = lane >> 5, d = lane & 31u;
        u32 v = d;
        for (u32 c = 0; c < 32; ++c) v = maps[((v & 31u) << 8) + wv * 64 + h * 32 + c];
        const u32 v2 = (u32)__shfl((int)v, 32 + (int)(__shfl((int)v, (int)d, 64) & 31), 64);   // second half applied to the first
        if (lane < 32) wfn[wv * R + d] = (u8)v2;
    }
    __syncthreads();
    u8 *tf = tilefn + (size_t)(blk.tile_base + tile) * R;
    if (tid < lmax) {
        u32 v = tid;
#pragma unroll
        for (int w = 0; w < 4; ++w) v = wfn[w * R + (v & 31u)];
        tf[tid] = (u8)v;
    }
    }
}

// counting automaton with 32-entry byte maps.  static LDS: data | maps[32][256] | fsm4 | fsm1 | wfn[4][32] | went[4][2] | ent[256] | wsum[4]
__global__ __launch_bounds__(DEC_THREADS) void sfd_countfsm32(const DecBlk *__restrict__ blks,
                                                              const u8 *__restrict__ chunkfn,
                                                              const u8 *__restrict__ tile_entry,
                                                              u8 *__restrict__ chunk_entry, u16 *__restrict__ chunk_cnt,
                                                              u32 *__restrict__ tile_cnt, u32 tpw)
{
    constexpr u32 R = 32;
    __shared__ __attribute__((aligned(16))) u8 smem[LDS_DATA + R * DEC_THREADS + 16384 + 2048 + 4 * R + 16 + DEC_THREADS + 64];
    if (dp_skipped_early(blks + blockIdx.y)) return;    // the block's speculative entries verified
    const DecBlk blk = blks[blockIdx.y];
    if (blockIdx.x * tpw >= blk.n_tiles) return;
    u32 *data = (u32 *)smem;
    u8 *maps = smem + LDS_DATA;
    const u8 *f4 = maps + R * DEC_THREADS;
    const u8 *f1 = f4 + 16384;
    u8 *wfn = (u8 *)f1 + 2048;
    u8 *went = wfn + 4 * R;
    u8 *ent = went + 16;
    u32 *wsum = (u32 *)(ent + DEC_THREADS);
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 lmax = blk.lmax;

    fill_lds16((void *)f4, blk.fsm4, blk.n_states * 64);
    fill_lds16((void *)f1, blk.fsm1, blk.n_states * 8);
    const u32 tile_end = (blockIdx.x + 1) * tpw < blk.n_tiles ? (blockIdx.x + 1) * tpw : blk.n_tiles;
    for (u32 tile = blockIdx.x * tpw; tile < tile_end; ++tile) {
    const size_t gt = (size_t)blk.tile_base + tile;
    __syncthreads();
    load_tile(data, blk, tile);
    {
        const u8 *cf = chunkfn + (gt * R << 8);
        for (u32 i = tid; i < lmax * (DEC_THREADS / 4); i += DEC_THREADS) ((u32 *)maps)[i] = ((const u32 *)cf)[i];
    }
    __syncthreads();
    // half maps (lane (h, d) follows entry d through 32 chunks), wave maps, wave / half entries, chunk entries
    const u32 h = lane >> 5, d = lane & 31u;
    u32 hv = d;
    for (u32 c = 0; c < 32; ++c) hv = maps[((hv & 31u) << 8) + wv * 64 + h * 32 + c];
    {
        const u32 v2 = (u32)__shfl((int)hv, 32 + (int)(__shfl((int)hv, (int)d, 64) & 31), 64);
        if (lane < 32) wfn[wv * R + d] = (u8)v2;
    }
    __syncthreads();
    {
        u32 e = tile_entry[gt];
        for (u32 w = 0; w < wv; ++w) e = wfn[w * R + (e & 31u)];         // entry of this wave
        const u32 e1 = (u32)__shfl((int)hv, (int)(e & 31u), 64);                  // entry of its second half
        if (lane == 0 || lane == 32) {
            u32 v = lane ? e1 : e;
            for (u32 c = 0; c < 32; ++c) {
                ent[wv * 64 + h * 32 + c] = (u8)v;
                v = maps[((v & 31u) << 8) + wv * 64 + h * 32 + c];
            }
        }
    }
    __syncthreads();
    const u32 entry = ent[tid];
    const u32 limit = tile_bit_limit(blk, tile);
    const bool last = limit < (u32)(DTILE + HALO_WORDS * 4) * 8;
    const u32 cw = tid * (CH_BITS / 32), cbase = tid * CH_BITS;
    u32 st = 0, cnt = 0, p = entry;
    auto bit_step = [&](u32 q) {
        const u32 bit = (data[widx(q >> 5)] >> (31 - (q & 31))) & 1u;
        const u32 e = *(const u32 *)(f1 + (st >> 3) + bit * 4);
        st = e & 0xFFFFu;
        cnt += e >> 16;
    };
    if (!last) {
        while (p & 3) { bit_step(cbase + p); ++p; }    // p <= 32 afterwards
        const u32 j0 = p >> 2;                          // first whole nibble (0..8): word 0, or the start of word 1
#pragma unroll
        for (int wi = 0; wi < 8; ++wi) {
            const u32 w = data[widx(cw + wi)];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u32 nib4 = j < 7 ? (w >> (26 - 4 * j)) & 0x3Cu : (w << 2) & 0x3Cu;
                const u32 e = *(const u32 *)(f4 + st + nib4);
                if (wi == 0) {                          // nibbles before the entry belong to the previous chunk
                    if ((u32)j >= j0) { st = e & 0xFFFFu; cnt += e >> 16; }
                } else {
                    st = e & 0xFFFFu;
                    cnt += e >> 16;
                }
            }
        }
        cnt += st != 0;
    } else {
        const u32 stop = cbase + CH_BITS < limit ? cbase + CH_BITS : limit;
        u32 q = cbase + p;
        for (; q < stop; ++q) bit_step(q);
        if (st != 0 && q == cbase + CH_BITS) {
            const u32 before = cnt;
            for (; q < limit && cnt == before; ++q) bit_step(q);
        }
    }
    chunk_entry[gt * DEC_THREADS + tid] = (u8)entry;
    chunk_cnt[gt * DEC_THREADS + tid] = (u16)cnt;
    const u32 tot = wave_reduce_add<u32>(cnt);
    if (lane == 0) wsum[wv] = tot;
    __syncthreads();
    if (tid == 0) tile_cnt[gt] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

// sfd_tiles16: per block, the entry offset of every tile.  Batches of 4096 tile maps in LDS; thread (s, d)
// follows entry d through segment s (1/16 of the batch), thread 0 links the 16 segments, then one thread per
// segment walks it again from its real entry: 2 * 256 + 16 dependent LDS reads per batch instead of 4096.
constexpr int TB = 4096;
__global__ __launch_bounds__(DEC_THREADS) void sfd_tiles16(const DecBlk *__restrict__ blks,
                                                           const u64 *__restrict__ tilefn, u8 *__restrict__ tile_entry)
{
    __shared__ u64 maps[TB];
    __shared__ u8 ent[TB];
    __shared__ u8 segmap[16 * 16], segent[16];
    __shared__ u32 carry;
    if (dp_skipped_early(blks + blockIdx.x)) return;
    const DecBlk blk = blks[blockIdx.x];
    const u32 tid = threadIdx.x, sg = tid >> 4, d = tid & 15u;
    if (tid == 0) carry = 0;
    for (u32 t0 = 0; t0 < blk.n_tiles; t0 += TB) {
        const u32 nt = (blk.n_tiles - t0 < (u32)TB) ? blk.n_tiles - t0 : (u32)TB;
        const u32 seg = (nt + 15) / 16;
        for (u32 i = tid; i < nt; i += DEC_THREADS) maps[i] = tilefn[(size_t)blk.tile_base + t0 + i];
        __syncthreads();
        const u32 lo = sg * seg < nt ? sg * seg : nt, hi = lo + seg < nt ? lo + seg : nt;
        {
            u32 v = d;
            for (u32 i = lo; i < hi; ++i) v = nib(maps[i], v);
            segmap[sg * 16 + d] = (u8)v;
        }
        __syncthreads();
        if (tid == 0) {
            u32 e = carry;
            for (u32 q = 0; q < 16; ++q) { segent[q] = (u8)e; e = segmap[q * 16 + e]; }
            carry = e;
        }
        __syncthreads();
        if (d == 0) {
            u32 v = segent[sg];
            for (u32 i = lo; i < hi; ++i) { ent[i] = (u8)v; v = nib(maps[i], v); }
        }
        __syncthreads();
        for (u32 i = tid; i < nt; i += DEC_THREADS) tile_entry[blk.tile_base + t0 + i] = ent[i];
        __syncthreads();
    }
}

// entry offset of every chunk of the tile: quarter chase with history, wave maps through LDS, then each lane
// picks its nibble.  cm = the tile's 256 chunk maps (LDS), hist = 256 u64 (LDS), wmb = 64 bytes (LDS).
__device__ __forceinline__ u32 chunk_entry_of(const u64 *cm, u64 *hist, u8 *wmb, u32 tile_entry_v, u32 tid = threadIdx.x)
{
    const u32 lane = tid & 63, wv = tid >> 6, q = lane >> 4;
    u64 h;
    const u32 qv = quarter_chase<true>(cm + wv * 64, &h);
    hist[tid] = h;                                              // tid == wv*64 + q*16 + d
    const u32 w = wave_map_of(qv);
    if (lane < 16) wmb[wv * 16 + lane] = (u8)w;
    __syncthreads();
    u32 e = tile_entry_v;
    for (u32 w2 = 0; w2 < wv; ++w2) e = wmb[w2 * 16 + e];       // entry of this wave
#pragma unroll
    for (int k = 0; k < 3; ++k) {                               // entry of this lane's quarter
        const u32 nx = __shfl(qv, k * 16 + (int)e, 64);
        if ((u32)k < q) e = nx;
    }
    return nib(hist[wv * 64 + q * 16 + e], lane & 15u);
}

// sfd_countfsm: symbol counts with the nibble automaton (complete codes, Lmax <= 16).  Every lane takes exactly 64
// table steps for its 256 bits whatever the code lengths are: no bit-buffer bookkeeping, no divergence between
// lanes, ~4 VALU instructions per nibble.  static LDS: data | cmap[256] u64 | fsm4[256*16] u32 | fsm1[256*2] u32 |
// hist[256] u64 | wmb[64] | wsum[4]
template <int SUBS>
__global__ __launch_bounds__(DEC_THREADS * SUBS) __attribute__((amdgpu_waves_per_eu(8, 8))) void sfd_countfsm(const DecBlk *__restrict__ blks,
                                                                   const u64 *__restrict__ chunkfn,
                                                                   const u8 *__restrict__ tile_entry,
                                                                   u8 *__restrict__ chunk_entry, u16 *__restrict__ chunk_cnt,
                                                                   u32 *__restrict__ tile_cnt, u32 tpw)
{
    // SUBS groups of 256 lanes, one tile each, share the automaton tables (the chain of 64 dependent look-ups per
    // lane is latency bound: waves per CU is what counts)
    constexpr int PER_SUB = LDS_DATA + DEC_THREADS * 16 + 64 + 16;
    __shared__ __attribute__((aligned(16))) u8 smem[SUBS * PER_SUB + 16384 + 2048 + 64];
    if (dp_skipped_early(blks + blockIdx.y)) return;
    const DecBlk blk = blks[blockIdx.y];
    const u32 first_tile = blockIdx.x * tpw * SUBS;
    if (first_tile >= blk.n_tiles) return;
    const u32 sub = threadIdx.x >> 8, tid = threadIdx.x & 255u, lane = tid & 63, wv = tid >> 6;
    u8 *mine = smem + sub * PER_SUB;
    u32 *data = (u32 *)mine;
    u64 *cm = (u64 *)(mine + LDS_DATA);
    u64 *hist = cm + DEC_THREADS;
    u8 *wmb = (u8 *)(hist + DEC_THREADS);
    u32 *wsum = (u32 *)(wmb + 64);
    const u8 *f4 = smem + SUBS * PER_SUB;
    const u8 *f1 = f4 + 16384;

    fill_lds16((void *)f4, blk.fsm4, blk.n_states * 64);
    fill_lds16((void *)f1, blk.fsm1, blk.n_states * 8);
    for (u32 it = 0; it < tpw && first_tile + it * SUBS < blk.n_tiles; ++it) {
    const u32 tile = first_tile + it * SUBS + sub;
    const bool active = tile < blk.n_tiles;            // uniform per 256-lane group
    const size_t gt = (size_t)blk.tile_base + (active ? tile : first_tile);
    __syncthreads();                                   // previous tile's LDS reads are done
    load_tile(data, blk, active ? tile : first_tile, tid);
    cm[tid] = chunkfn[gt * DEC_THREADS + tid];
    __syncthreads();
    const u32 entry = chunk_entry_of(cm, hist, wmb, tile_entry[gt], tid);
    const u32 limit = tile_bit_limit(blk, tile);
    const bool last = limit < (u32)(DTILE + HALO_WORDS * 4) * 8;          // the stream ends inside this window
    const u32 cw = tid * (CH_BITS / 32), cbase = tid * CH_BITS;
    u32 st = 0, cnt = 0, p = entry;
    auto bit_step = [&](u32 q) {                        // consume tile-local bit q
        const u32 bit = (data[widx(q >> 5)] >> (31 - (q & 31))) & 1u;
        const u32 e = *(const u32 *)(f1 + (st >> 3) + bit * 4);
        st = e & 0xFFFFu;
        cnt += e >> 16;
    };
    if (!last) {
        while (p & 3) { bit_step(cbase + p); ++p; }    // up to the next nibble boundary (p <= 16 afterwards)
        const u32 j0 = p >> 2;                          // first whole nibble of word 0 (0..4)
#pragma unroll
        for (int wi = 0; wi < 8; ++wi) {
            const u32 w = data[widx(cw + wi)];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u32 nib4 = j < 7 ? (w >> (26 - 4 * j)) & 0x3Cu : (w << 2) & 0x3Cu;
                const u32 e = *(const u32 *)(f4 + st + nib4);
                if (wi == 0 && j < 4) {                 // nibbles before the entry belong to the previous chunk
                    if ((u32)j >= j0) { st = e & 0xFFFFu; cnt += e >> 16; }
                } else {
                    st = e & 0xFFFFu;
                    cnt += e >> 16;
                }
            }
        }
        cnt += st != 0;                                 // the code in progress at bit 256 started in this chunk
    } else {
        // last tile of the block: bit by bit; only codes that end inside the stream are symbols
        const u32 stop = cbase + CH_BITS < limit ? cbase + CH_BITS : limit;
        u32 q = cbase + p;
        for (; q < stop; ++q) bit_step(q);
        if (st != 0 && q == cbase + CH_BITS) {          // finish the code in progress (it started in this chunk)
            const u32 before = cnt;
            for (; q < limit && cnt == before; ++q) bit_step(q);
        }
    }
    if (active) {
        chunk_entry[gt * DEC_THREADS + tid] = (u8)entry;
        chunk_cnt[gt * DEC_THREADS + tid] = (u16)cnt;
    }
    const u32 tot = wave_reduce_add<u32>(cnt);
    if (lane == 0) wsum[wv] = tot;
    __syncthreads();
    if (active && tid == 0) tile_cnt[gt] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

}  // namespace
