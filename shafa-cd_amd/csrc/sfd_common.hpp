// sfd_common.hpp — constants, the per-block record and the helpers every sf_decode kernel family uses
// Part of sf_decode.hip's translation unit (included there; not compiled on its own).
#pragma once

namespace {


constexpr int DEC_THREADS = 256;
constexpr int CH_BYTES = 32;                       // chunk = 256 bits per lane
constexpr int CH_BITS = CH_BYTES * 8;
constexpr int DTILE = DEC_THREADS * CH_BYTES;      // 8 KiB of stream per tile
constexpr int HALO_WORDS = 16;                     // 64 bytes past the tile (windows + trie walks)
constexpr int DATA_WORDS = DTILE / 4 + HALO_WORDS;
constexpr int LUT_MAXK = 11;
constexpr int LUT2_MAX = 4096;                     // level-2 entries kept in LDS (8 KiB)
constexpr int SPEC_MINW = 1;                       // narrowest window of the counting table of sfd_scan (1 = the longest code's
                                                   // length; measured on Lmax = 10 data: 11 bits no gain, 12 bits 3 % slower:
                                                   // fewer steps, but an 8 KiB table costs two workgroups per CU)
constexpr int SYM3_MINW = 1;                       // narrowest window of the three-symbols table of sfd_wstage (1 = the longest
                                                   // code; 12 on Lmax = 10 data: 14 % fewer look-ups, but a 16 KiB table leaves
                                                   // four workgroups per CU instead of six: decode 10.1 -> 11.1 ms)
constexpr int SYM3_MAXK = 12;                      // widest window of the three-codes table of sfd_wstage: 16 KiB
constexpr int LDS_DATA = (DATA_WORDS + DATA_WORDS / 8 + 8) * 4;
constexpr int LONG_PFX = 128;                      // codes longer than SYM3_MAXK bits, grouped by their first SYM3_MAXK bits
constexpr int LONG_BYTES = 16 + LONG_PFX * 2 + LONG_PFX * 16 * 2;   // [n u16 x8 pad][pfx u16 x128][ent u16 x128x16]
// codes of 13..32 bits: [n u16 x8 pad][pfx u16 x128][root u16 x128][nodes {child0, child1} u16 x2 x256]; child = node | 0x8000|sym
constexpr int LONG32_BYTES = 16 + LONG_PFX * 2 + LONG_PFX * 2 + 256 * 4 + LONG_PFX * 2;   // + root13 u16 x128 (13-bit prefixes)
constexpr int LEN_MAXK = 13;                       // length-only LUT of the packed DP: 8 KiB

struct DecBlk {
    const u8 *in;
    u8 *out;
    u64 in_n;
    u64 n_sym;
    int *err;
    const u16 *lut;        // level 1, 2^K entries: sym | len << 8 ; 0x8000 | (nb-1) << 12 | base = level-2
                           //   group (next nb bits index lut2[base..]) ; 0 = go to the trie
    const u16 *lut2;       // level 2 (codes of K+1 .. K+8 bits): sym | len << 8 ; 0 = go to the trie
    const u8 *lenlut;      // 2^K1 entries: len only (DP of the packed path); 0 = longer than K1 bits
    u16 *cnt3;             // sfd_scan's tables, 2^KW bytes each: [total bits | codes << 5 of the whole codes (at most 7) in a
                           // window] then [length of the window's first code]
    u32 *sym3;             // 2^K3 entries: sym0 | sym1 << 8 | sym2 << 16 | total bits << 24 (6 bits) | n << 30
    u8 *pairlut;           // 2^(K1+1) entries: (len(p)-1) | (len(p+1)-1) << 4 from a K1+1-bit window (complete codes)
    const u16 *lut13;      // 2^K1 entries: sym | len << 8, single level (only when Lmax <= 13), else NULL
    const u32 *trie;       // pairs {child0, child1}: 0x80000000|sym = leaf, 0xFFFFFFFF = missing
    u32 K;
    u32 K1;
    u32 lmax;
    u32 tile_base;         // first tile of this block in the per-tile arrays
    u32 n_tiles;
    u32 n_l2;              // level-2 entries
    u32 n_states;          // internal trie nodes = states of the counting automaton (<= 255 for a complete code)
    u32 KW;                // window of sfd_scan's counting tables (spec_window(K1), or 12 for 13-bit tables whose 13-bit codes are few)
    u32 *fsm4;             // [state][nibble]: next state * 64 | codes completed << 16   (complete codes; sfd_tables)
    u32 *fsm1;             // [state][bit]   : same, for one bit
    const u8 *lenlut32;    // 2^13 entries: len <= 13, or 128 + k = internal node root13[k] of long32 (16 < Lmax <= 32 launches)
    const u16 *long32;     // LONG32_BYTES: sorted 12-bit prefixes of the codes of 13..32 bits + their sub-tries
    const u16 *longtab;    // LONG_BYTES: sorted 12-bit prefixes of the codes of 13..16 bits + 16 entries sym | len << 8 each
    u32 *run_dp;           // speculative launches: *run_dp != 0 <=> the block needs the exact (DP) kernels: it was not
                           //   tried speculatively or did not verify; NULL: no speculation, the DP kernels always run
};

// window of the code-counting table of sfd_scan: wider than the longest code when that is short (more bits per look-up)
__host__ __device__ __forceinline__ u32 spec_window(u32 K1) { return K1 < (u32)SPEC_MINW ? (u32)SPEC_MINW : K1; }
// window of the three-symbols table of sfd_wstage
__host__ __device__ __forceinline__ u32 sym3_window(u32 K1)
{
    const u32 k = K1 < (u32)SYM3_MINW ? (u32)SYM3_MINW : K1;
    return k < (u32)SYM3_MAXK ? k : (u32)SYM3_MAXK;
}

// the exact kernels of a launch that also runs the speculative ones: skip the blocks that verified
__device__ __forceinline__ bool dp_skipped(const DecBlk &blk)
{
    return blk.run_dp && __hip_atomic_load(blk.run_dp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}
// the same question before the block record is copied (the usual answer is "skipped": read one pointer, not the record)
__device__ __forceinline__ bool dp_skipped_early(const DecBlk *p)
{
    u32 *const rd = p->run_dp;
    return rd && __hip_atomic_load(rd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// stream words are kept big-endian in LDS, one pad word per 8 (chunk stride 9 words: no bank conflicts)
__device__ __forceinline__ u32 widx(u32 w) { return w + (w >> 3); }

struct Code { u32 len; u32 sym; bool ok; };

// length (and symbol) of the code that starts at tile-local bit position p
__device__ __forceinline__ Code code_at(const u32 *data, const u16 *lut, const u32 *trie, u32 K, u32 p,
                                        bool trie_only = false)
{
    const u32 w = p >> 5, r = p & 31;
    const u64 two = ((u64)data[widx(w)] << 32) | data[widx(w + 1)];
    const u32 win = (u32)((two << r) >> 32);
    u32 e = trie_only ? 0u : lut[win >> (32 - K)];
    Code c;
    if (e & 0x8000u) {                              // codes of K+1..K+8 bits: one more table
        const u32 nb = ((e >> 12) & 7u) + 1;
        e = lut[(1u << LUT_MAXK) + (e & 0xFFFu) + ((win << K) >> (32 - nb))];
    }
    if (e) { c.len = e >> 8; c.sym = e & 0xFF; c.ok = true; return c; }
    // slow path: walk the trie bit by bit (codes longer than K bits, or an incomplete tree)
    u32 node = 0, q = p, depth = 0;
    for (;;) {
        const u32 bit = (data[widx(q >> 5)] >> (31 - (q & 31))) & 1u;
        const u32 nx = trie[2 * node + bit];
        ++q; ++depth;
        if (nx == 0xFFFFFFFFu) { c.len = 1; c.sym = 0; c.ok = false; return c; }
        if (nx & 0x80000000u) { c.len = depth; c.sym = nx & 0xFF; c.ok = true; return c; }
        node = nx;
        if (depth >= 255) { c.len = 1; c.sym = 0; c.ok = false; return c; }
    }
}

// stage one tile (+halo) of the stream into LDS as big-endian words; bytes past in_n read as zero
__device__ __forceinline__ void load_tile(u32 *data, const DecBlk &blk, u32 tile, u32 t = threadIdx.x)
{
    // all of a lane's pieces are requested before the first is used (three loads in flight, not three round trips)
    const u64 base = (u64)tile * DTILE;
    constexpr u32 UNITS = DATA_WORDS / 4, NIT = (UNITS + DEC_THREADS - 1) / DEC_THREADS;
    uint4 v[NIT];
#pragma unroll
    for (u32 it = 0; it < NIT; ++it) {
        const u32 i = t + it * DEC_THREADS;
        const u64 off = base + (u64)i * 16;
        v[it] = make_uint4(0, 0, 0, 0);
        if (i < UNITS && off + 16 <= blk.in_n) v[it] = gload<uint4>(blk.in + off);
    }
#pragma unroll
    for (u32 it = 0; it < NIT; ++it) {
        const u32 i = t + it * DEC_THREADS;
        if (i >= UNITS) break;
        const u64 off = base + (u64)i * 16;
        u32 w[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
        if (off < blk.in_n && off + 16 > blk.in_n) {    // the piece the stream ends in
            const int nv = (int)(blk.in_n - off);
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < nv) w[q >> 2] |= (u32)gload<u8>(blk.in + off + q) << (8 * (q & 3));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) data[widx(4 * i + q)] = bswap32(w[q]);
    }
}

__device__ __forceinline__ void load_lut(u16 *lut, const DecBlk &blk)
{
    const u32 n32 = (1u << blk.K) / 2;            // K >= 1
    for (u32 i = threadIdx.x; i < (n32 ? n32 : 1); i += DEC_THREADS)
        ((u32 *)lut)[i] = gload<u32>((const u32 *)blk.lut + i);
    u32 *l2 = (u32 *)(lut + (1u << LUT_MAXK));
    for (u32 i = threadIdx.x; i < (blk.n_l2 + 1) / 2; i += DEC_THREADS) l2[i] = gload<u32>((const u32 *)blk.lut2 + i);
}

}  // namespace
