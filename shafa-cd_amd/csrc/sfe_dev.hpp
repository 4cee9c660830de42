// sfe_dev.hpp — device code shared by the one-pass Shannon-Fano encoders (sf_encode4.hip: chained tiles; sf_encode6.hip:
// tiles whose output offsets are known before the launch): {code, len} look-ups, octs / quads (the bit strings of 8 / 4
// symbols in registers), their placement in an LDS window (atomic OR: `place`; plain stores with carries: `emit_oct`), the
// window's way out (funnel shift, 16-byte stores) and the one-workgroup kernels for a block's ragged last tile.
// Reference: compress_to_buffer + binary_coding, c.c:52-237.
#pragma once
#include "common.hpp"
#include "internal.hpp"

namespace {

// A workgroup has NT = 256 or 1024 lanes and a tile is four rows of NT octs: 8 KiB or 32 KiB of symbols.  The wide form
// (launches whose three 32 KiB-tile windows fit a CU's LDS: Lmax <= 12; one workgroup per CU) pays wave 0's chain work
// once per 32 KiB instead of once per 8 KiB and spreads the window stores over fifteen waves instead of three.
constexpr int E4_GUARD = 8;                            // dwords in front of the window: an oct writes up to 4 dwords before its last one
constexpr u32 E4_NONE = 0xFFFFFFFFu;
#define E4_WPS 4                                       // waves per SIMD the register allocation aims at

struct E4Static {
    u64 lut[256];                // {code (low dword), len (high dword)}; a symbol without a code: len = 1 << 16
    u32 wtot[64];                // sfe4: [row][wave] bit totals, i.e. in stream order (4 rows x up to 8 waves); sfe5: [wave]
    u32 dump[64];                // sfe5: where the stores of lanes that have nothing to store go (one word per lane)
    // sfe5 has ONE barrier per iteration, so what the waves hand each other across it exists twice (iteration parity):
    u32 wtot5[2][16];            //   bit total of every wave's string
    u32 tail5[2][16];            //   the last 32 bits of every wave's string
    u64 prefix5[2];              //   bits before the tile that is stored in this iteration
    u32 tick5[2];                //   next ticket of the block
    u64 prefix;                  // bits before the pending tile
    u32 tick;                    // next ticket of the block (broadcast)
    u32 pad;
};

struct Oct {                     // 8 symbols: right-aligned 128-bit string r3:r2:r1:r0, ll bits (bit 16+: a symbol had no code)
    u32 r0, r1, r2, r3, ll;
};

// four {code, len} entries -> right-aligned group g of L bits (L <= 64)
__device__ __forceinline__ void quad(u64 e0, u64 e1, u64 e2, u64 e3, u64 &g, u32 &L)
{
    const u32 c0 = (u32)e0, c1 = (u32)e1, c2 = (u32)e2, c3 = (u32)e3;
    const u32 l0 = (u32)(e0 >> 32), l1 = (u32)(e1 >> 32), l2 = (u32)(e2 >> 32), l3 = (u32)(e3 >> 32);
    const u32 a = (c0 << (l1 & 31u)) | c1;             // <= 32 bits
    const u32 b = (c2 << (l3 & 31u)) | c3;
    const u32 lb = l2 + l3;
    g = ((u64)a << (lb & 63u)) | b;
    L = l0 + l1 + lb;
}

// The 8 symbols of two input dwords -> oct.  All eight look-ups are issued before the first is used.
// SAFE: lengths may be 0 (symbols past the block encode as nothing); L16: a group may be exactly 64 bits (four 16-bit codes)
template <bool SAFE, bool L16, bool HAVE_R3>
__device__ __forceinline__ Oct make_oct(const u64 *lut, u32 w0, u32 w1, u32 drop8)
{
    u64 e[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        e[j] = lut[(w0 >> (8 * j)) & 0xFFu];
        e[4 + j] = lut[(w1 >> (8 * j)) & 0xFFu];
    }
    if (SAFE) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if ((drop8 >> ((j & 3) + 4 * (j >> 2) )) & 1u) e[j] = 0;
    }
    u64 g0, g1;
    u32 L0, L1;
    quad(e[0], e[1], e[2], e[3], g0, L0);
    quad(e[4], e[5], e[6], e[7], g1, L1);
    const u32 s = L1 & 0xFFFFu;
    u64 lo = g0 << (s & 63u);                          // s in [4, 60] on the plain path
    u64 hi = g0 >> ((64u - s) & 63u);
    if (L16 && s >= 64u) lo = 0;                       // everything moved into hi (shift by 64 wraps to 0 in hardware)
    if (SAFE && s == 0u) hi = 0;
    lo |= g1;
    Oct o;
    o.r0 = (u32)lo;
    o.r1 = (u32)(lo >> 32);
    o.r2 = (u32)hi;
    o.r3 = HAVE_R3 ? (u32)(hi >> 32) : 0u;
    o.ll = L0 + L1;
    return o;
}

// OR an oct that ends at window bit e (exclusive) into the window; NW = dwords an oct of this launch can touch
template <int NW>
__device__ __forceinline__ void place(u32 *win, const Oct &o, u32 e)
{
    u32 *p = win + (e >> 5);
    const u32 x0 = __builtin_amdgcn_alignbit(o.r0, 0u, e);          // e mod 32 == 0: nothing in dword e >> 5
    const u32 x1 = __builtin_amdgcn_alignbit(o.r1, o.r0, e);
    __hip_atomic_fetch_or(p, x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_or(p - 1, x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (NW >= 3) {
        const u32 x2 = __builtin_amdgcn_alignbit(o.r2, o.r1, e);
        __hip_atomic_fetch_or(p - 2, x2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (NW >= 4) {
        const u32 x3 = __builtin_amdgcn_alignbit(o.r3, o.r2, e);
        __hip_atomic_fetch_or(p - 3, x3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (NW >= 5) {
        const u32 x4 = o.r3 >> (e & 31u);
        __hip_atomic_fetch_or(p - 4, x4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

struct TileIn {
    uint2 v[4];                  // row k of the tile = bytes [2048 k, 2048 k + 2048): lane t holds bytes 8 t .. 8 t + 7 of it (one oct)
};

// A FULL tile.  8 bytes per lane and row: in the instruction that places "the oct of row k" neighbouring lanes hold
// neighbouring octs, so a 32-lane LDS group spans ~1.6 dwords per lane instead of ~3.3 with 16 bytes per lane.
// Uniform base + 32-bit lane offset: the loads take the SGPR-base form, no 64-bit address registers.
template <int NT>
__device__ __forceinline__ void load_tile(const u8 *in, u32 tile, int tid, TileIn &t)
{
    const u8 *tb = in + (u64)tile * (32u * NT);
#pragma unroll
    for (int k = 0; k < 4; ++k) t.v[k] = gload_nt_off<uint2>(tb, (u32)k * (8u * NT) + (u32)tid * 8u);
}

// one tile's look-ups, groups and lane scans; returns the packed inclusive lane prefixes of rows (0,1) and (2,3)
template <bool SAFE, int NW, bool L16, int NT>
__device__ __forceinline__ void tile_octs(const u64 *lut, const TileIn &in, u32 keep_base, int tid, Oct (&oct)[4], u32 (&incl)[2],
                                          u32 &absent)
{
    constexpr bool R3 = NW >= 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        u32 drop = 0;
        if (SAFE) {                                    // keep_base = symbols of the tile that exist
            const u32 idx = (u32)k * (8u * NT) + (u32)tid * 8u;
            const u32 keep = idx >= keep_base ? 0u : (keep_base - idx >= 8u ? 8u : keep_base - idx);
            drop = (0xFFu << keep) & 0xFFu;
        }
        oct[k] = make_oct<SAFE, L16, R3>(lut, in.v[k].x, in.v[k].y, drop);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        absent |= oct[k].ll >> 16;
        oct[k].ll &= 0xFFFFu;                          // <= 128 bits; a wave's row total <= 8192 < 2^16
    }
    incl[0] = dpp_scan_add(oct[0].ll | (oct[1].ll << 16));
    incl[1] = dpp_scan_add(oct[2].ll | (oct[3].ll << 16));
}

// the (row, wave) totals in stream order -> this wave's four row offsets and the tile total: one DPP scan
// (NWV = 4: 16 totals, a scan inside the 16-lane DPP rows; NWV = 8: 32 totals, the wave-wide scan)
template <int NWV>
__device__ __forceinline__ u32 tile_offsets(const u32 *wtot, int lane, int wv, u32 (&roff)[4])
{
    u32 tot, sc;
    if (NWV == 4) {
        tot = wtot[lane & 15];
        sc = tot;
        sc += (u32)__builtin_amdgcn_update_dpp(0, (int)sc, 0x111, 0xf, 0xf, false);   // row_shr:1
        sc += (u32)__builtin_amdgcn_update_dpp(0, (int)sc, 0x112, 0xf, 0xf, false);   // row_shr:2
        sc += (u32)__builtin_amdgcn_update_dpp(0, (int)sc, 0x114, 0xf, 0xf, false);   // row_shr:4
        sc += (u32)__builtin_amdgcn_update_dpp(0, (int)sc, 0x118, 0xf, 0xf, false);   // row_shr:8
    } else {
        tot = lane < 4 * NWV ? wtot[lane] : 0u;
        sc = dpp_scan_add(tot);
    }
    const u32 ex = sc - tot;
#pragma unroll
    for (int k = 0; k < 4; ++k) roff[k] = (u32)__builtin_amdgcn_readlane((int)ex, k * NWV + wv);
    return (u32)__builtin_amdgcn_readlane((int)sc, 4 * NWV - 1);
}

// lead bits of a tile that starts at bit B of its block: the last r = B mod 32 bits before it, right-aligned, ORed into
// the lead word pwin[-1] (wave 0, all lanes; pv = symbol (tile start - 1 - lane) in lanes 0..31)
__device__ __forceinline__ void lead_bits(const u64 *lut, u32 *pwin, u32 pv, u32 r, int lane)
{
    const u64 ent = lane < 32 ? lut[pv & 0xFFu] : 0ull;
    const u32 code = (u32)ent, len = (u32)(ent >> 32) & 0xFFFFu;
    const u32 D = dpp_scan_add(len);                   // bits from this symbol's first bit to the tile start
    if (len && D - len < r)
        __hip_atomic_fetch_or(pwin - 1, code << ((D - len) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// A tile leaves LDS: output dword j of the tile = bits [32 j - r, 32 j - r + 32) of its window (r = B mod 32; the bits
// in front of the window come from the lead word pwin[-1]).  The tile owns the output dwords [B >> 5, E >> 5), the
// block's last tile also the final partial dword's bytes.
__device__ __forceinline__ void store_window(const u32 *pwin, u8 *out, u64 out_cap, int *err, u64 B, u32 T, bool last, int tid,
                                             u32 nthreads)
{
    const u32 r = (u32)B & 31u;
    const u64 E = B + T;
    const u64 total_bytes = (E + 7) >> 3;              // the block's size when this is its last tile
    const u64 gd0 = B >> 5;
    const u32 count = (u32)((last ? (total_bytes >> 2) : (E >> 5)) - gd0);      // owned dwords
    const u64 end_bytes = last ? total_bytes : 4 * (E >> 5);
    if (end_bytes > out_cap) {
        if (tid == 0) set_error(err, SHAFA_LACK_OF_MEMORY);
        return;
    }
    u8 *o = out + 4 * gd0;                             // owned dword 0
    u32 h = (u32)(0 - gd0) & 3u;                       // dwords up to the first 16-byte boundary
    if (h > count) h = count;
    const u32 nq = (count - h) >> 2;                   // aligned 16-byte pieces
    // A 16-byte piece needs the five window dwords j-1 .. j+3, j = h + 4 q: every lane reads the two 16-byte aligned
    // LDS pieces that hold them (conflict-free ds_read_b128; five ds_read_b32 at a 16-byte lane stride would be 4-way
    // bank conflicts) and picks its five dwords by the uniform phase (h - 1) mod 4.
    const u32 ph = (h + 3u) & 3u;                      // (j - 1) mod 4, the same for every piece of the tile
    const uint4 *p4 = (const uint4 *)(pwin - 4) + ((h + 3u) >> 2);     // 16-byte piece that holds dword h - 1
    for (u32 q = (u32)tid; q < nq; q += nthreads) {
        const uint4 a = p4[q], c = p4[q + 1];
        u32 w0, w1, w2, w3, w4;
        if (ph == 0) { w0 = a.x; w1 = a.y; w2 = a.z; w3 = a.w; w4 = c.x; }
        else if (ph == 1) { w0 = a.y; w1 = a.z; w2 = a.w; w3 = c.x; w4 = c.y; }
        else if (ph == 2) { w0 = a.z; w1 = a.w; w2 = c.x; w3 = c.y; w4 = c.z; }
        else { w0 = a.w; w1 = c.x; w2 = c.y; w3 = c.z; w4 = c.w; }
        gstore_nt_off<uint4>(o, 4 * (h + 4 * q), make_uint4(bswap32(__builtin_amdgcn_alignbit(w0, w1, r)),
                                                            bswap32(__builtin_amdgcn_alignbit(w1, w2, r)),
                                                            bswap32(__builtin_amdgcn_alignbit(w2, w3, r)),
                                                            bswap32(__builtin_amdgcn_alignbit(w3, w4, r))));
    }
    const u32 t0 = h + 4 * nq;                         // tail dwords [t0, count)
    if ((u32)tid < 8) {
        const u32 j = (u32)tid < 4 ? (u32)tid : t0 + (u32)tid - 4;
        const bool ok = (u32)tid < 4 ? j < h : j < count;
        if (ok) gstore_off<u32>(o, 4 * j, bswap32(__builtin_amdgcn_alignbit(pwin[(int)j - 1], pwin[j], r)));
    } else if (last && (u32)tid < 11) {                // the block's final 1..3 bytes
        const u32 q = (u32)tid - 8;
        const u32 w = __builtin_amdgcn_alignbit(pwin[(int)count - 1], pwin[count], r);
        if (q < (u32)(total_bytes & 3)) gstore_off<u8>(o, 4 * count + q, (u8)(w >> (24 - 8 * q)));
    }
}

// The same for sfe5_kernel, written so that the compiler cannot take it apart: two 16-byte aligned ds_read_b128 per
// 16-byte piece issued for TWO pieces per lane before the first is used (the C++ form above compiles to ds_read_b64 /
// ds_read2_b32 / ds_read_b32 mixes behind scalar branches, one LDS round trip per piece; in sfe5 the output stage
// was 0.63 of 3.39 ms).  pw4 = the tile's window as 16-byte pieces, pw4[-1].w = the lead word.
typedef unsigned int e5_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ e5_u32x4 e5_lds_read128(u32 byte_addr)
{
    e5_u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr) : "memory");
    return v;
}
template <int PH>
__device__ __forceinline__ uint4 e5_piece(const e5_u32x4 a, const e5_u32x4 c, u32 r)
{
    const u32 w0 = PH == 0 ? a.x : PH == 1 ? a.y : PH == 2 ? a.z : a.w;
    const u32 w1 = PH == 0 ? a.y : PH == 1 ? a.z : PH == 2 ? a.w : c.x;
    const u32 w2 = PH == 0 ? a.z : PH == 1 ? a.w : PH == 2 ? c.x : c.y;
    const u32 w3 = PH == 0 ? a.w : PH == 1 ? c.x : PH == 2 ? c.y : c.z;
    const u32 w4 = PH == 0 ? c.x : PH == 1 ? c.y : PH == 2 ? c.z : c.w;
    return make_uint4(bswap32(__builtin_amdgcn_alignbit(w0, w1, r)), bswap32(__builtin_amdgcn_alignbit(w1, w2, r)),
                      bswap32(__builtin_amdgcn_alignbit(w2, w3, r)), bswap32(__builtin_amdgcn_alignbit(w3, w4, r)));
}
template <int PH>
__device__ __forceinline__ void e5_store_pieces(u32 lds_base, u8 *o, u32 obase, u32 nq, u32 r, int tid, u32 nthreads)
{
    for (u32 q = (u32)tid; q < nq; q += 2u * nthreads) {
        const u32 q2 = q + nthreads;
        const bool two = q2 < nq;                      // nearly wave-uniform: only the tile's last wave splits
        e5_u32x4 a0 = e5_lds_read128(lds_base + 16u * q), c0 = e5_lds_read128(lds_base + 16u * q + 16u);
        e5_u32x4 a1 = a0, c1 = c0;
        if (two) {
            a1 = e5_lds_read128(lds_base + 16u * q2);
            c1 = e5_lds_read128(lds_base + 16u * q2 + 16u);
        }
        // the registers are operands of the wait: nothing that uses them may be scheduled in front of it
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(c0), "+v"(a1), "+v"(c1) : : "memory");
        gstore_nt_off<uint4>(o, obase + 16u * q, e5_piece<PH>(a0, c0, r));
        if (two) gstore_nt_off<uint4>(o, obase + 16u * q2, e5_piece<PH>(a1, c1, r));
    }
}
__device__ __forceinline__ void store_window5(const u32 *pwin, u8 *out, u64 out_cap, int *err, u64 B, u32 T, bool last, int tid,
                                              u32 nthreads)
{
    const u32 r = (u32)B & 31u;
    const u64 E = B + T;
    const u64 total_bytes = (E + 7) >> 3;              // the block's size when this is its last tile
    const u64 gd0 = B >> 5;
    const u32 count = (u32)((last ? (total_bytes >> 2) : (E >> 5)) - gd0);      // owned dwords
    const u64 end_bytes = last ? total_bytes : 4 * (E >> 5);
    if (end_bytes > out_cap) {
        if (tid == 0) set_error(err, SHAFA_LACK_OF_MEMORY);
        return;
    }
    u8 *o = out + 4 * gd0;                             // owned dword 0
    u32 h = (u32)(0 - gd0) & 3u;                       // dwords up to the first 16-byte boundary
    if (h > count) h = count;
    const u32 nq = (count - h) >> 2;                   // aligned 16-byte pieces
    const u32 ph = (h + 3u) & 3u;                      // (j - 1) mod 4 of a piece's first window dword j - 1: uniform
    const u32 lds_base = lds_addr(pwin) - 16u + 16u * ((h + 3u) >> 2);          // the 16-byte piece that holds dword h - 1
    if (ph == 0) e5_store_pieces<0>(lds_base, o, 4u * h, nq, r, tid, nthreads);
    else if (ph == 1) e5_store_pieces<1>(lds_base, o, 4u * h, nq, r, tid, nthreads);
    else if (ph == 2) e5_store_pieces<2>(lds_base, o, 4u * h, nq, r, tid, nthreads);
    else e5_store_pieces<3>(lds_base, o, 4u * h, nq, r, tid, nthreads);
    const u32 t0 = h + 4 * nq;                         // tail dwords [t0, count)
    if ((u32)tid < 8) {
        const u32 j = (u32)tid < 4 ? (u32)tid : t0 + (u32)tid - 4;
        const bool ok = (u32)tid < 4 ? j < h : j < count;
        if (ok) gstore_off<u32>(o, 4 * j, bswap32(__builtin_amdgcn_alignbit(pwin[(int)j - 1], pwin[j], r)));
    } else if (last && (u32)tid < 11) {                // the block's final 1..3 bytes
        const u32 q = (u32)tid - 8;
        const u32 w = __builtin_amdgcn_alignbit(pwin[(int)count - 1], pwin[count], r);
        if (q < (u32)(total_bytes & 3)) gstore_off<u8>(o, 4 * count + q, (u8)(w >> (24 - 8 * q)));
    }
}

// =====================================================================================================================
// Windows filled by PLAIN LDS stores (no atomics, no zeroing): sfe5_kernel (sf_encode4.hip) and sfe6_kernel (sf_encode6.hip).
//
// A lane owns 32 CONSECUTIVE symbols of the tile (four octs), so its bit string is ~6 dwords long and all but its two
// end dwords belong to it alone.  An oct that covers window bits [s, e) stores every dword it has bits in EXCEPT the one
// it ends in: that partial dword (x0) travels on as a carry and is ORed into the first dword of whatever comes next — the
// lane's next oct, the next lane's first oct (DPP wave_shr:1 of the lane's final partial dword, which is known before any
// oct is placed: it is the last e mod 32 bits of the lane's string), or the next wave's first oct (the waves publish
// their last 32 bits next to their totals).  Every window dword in [0, T >> 5) is stored exactly once, the tile's last
// lane adds the final partial dword and a zero behind it (the padding of a block's last byte).  Model with the algebra
// checked against a direct concatenation: tools/model/sfe5_model.py.
//
// Per lane and tile: 32 table look-ups and <= 12 ds_write_b32 (a store that is not due goes to the lane's dump word)
// where round 2's atomic-OR windows took 32 look-ups, 16 ds_or_b32 and a zeroing pass (`place`, kept for the ragged-tail
// kernels, whose strings may be shorter than a dword).
// =====================================================================================================================
#define E5_STORE_W0 1                                  // the waves E5_STORE_W0 .. NWV-1 store the resolved window
struct TileIn5 {
    uint4 v[2];                  // lane t holds bytes [32 t, 32 t + 32) of the tile
};

template <int NT>
__device__ __forceinline__ void load_tile5(const u8 *in, u32 tile, int tid, TileIn5 &t)
{
    const u8 *tb = in + (u64)tile * (32u * NT);
    // 32 bytes per lane = two 16-byte loads at a 32-byte lane stride: each instruction uses half of every cache line it
    // touches, the other half is the second instruction's.  Default (L1-allocating) policy: with `nt` the second
    // instruction fetches the lines again (measured: 4.05 ms per 8 GiB with nt, 3.72 plain, 3.66 with a fully coalesced
    // but wrong symbol order).
    t.v[0] = gload_off<uint4>(tb, (u32)tid * 32u);
    t.v[1] = gload_off<uint4>(tb, (u32)tid * 32u + 16u);
}

// Codes of up to 32 bits (class 2: a real file's rare bytes at -b M): the unit is a QUAD, the 4 symbols of one input
// dword as a right-aligned string of up to 128 bits in the same Oct registers; eight quads per lane instead of four octs.
// SAFE: lengths may be 0 (symbols past the block's end encode as nothing).
template <bool SAFE>
__device__ __forceinline__ Oct make_quad4(const u64 *lut, u32 w, u32 drop4)
{
    u64 e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = lut[(w >> (8 * j)) & 0xFFu];
    if (SAFE) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((drop4 >> j) & 1u) e[j] = 0;
    }
    const u32 c0 = (u32)e[0], c1 = (u32)e[1], c2 = (u32)e[2], c3 = (u32)e[3];
    const u32 l0 = (u32)(e[0] >> 32), l1 = (u32)(e[1] >> 32), l2 = (u32)(e[2] >> 32), l3 = (u32)(e[3] >> 32);
    const u64 p0 = ((u64)c0 << (l1 & 63u)) | c1;       // <= 64 bits
    const u64 p1 = ((u64)c2 << (l3 & 63u)) | c3;
    const u32 s = (l2 + l3) & 0xFFFFu;                 // 2 .. 64 on the plain path
    u64 lo = p0 << (s & 63u);
    u64 hi = p0 >> ((64u - s) & 63u);                  // s == 64: everything moves into hi (a shift by 64 wraps to 0)
    if (s >= 64u) lo = 0;
    if (SAFE && s == 0u) hi = 0;
    lo |= p1;
    Oct o;
    o.r0 = (u32)lo;
    o.r1 = (u32)(lo >> 32);
    o.r2 = (u32)hi;
    o.r3 = (u32)(hi >> 32);
    o.ll = l0 + l1 + l2 + l3;
    return o;
}

// the lane's four octs, its inclusive bit prefix inside the wave and the last 32 bits of its string
template <int NW, bool L16>
__device__ __forceinline__ void tile_octs5(const u64 *lut, const TileIn5 &in, Oct (&oct)[4], u32 &tot, u32 &incl, u32 &tail,
                                           u32 &absent)
{
    constexpr bool R3 = NW >= 5;
    oct[0] = make_oct<false, L16, R3>(lut, in.v[0].x, in.v[0].y, 0u);
    oct[1] = make_oct<false, L16, R3>(lut, in.v[0].z, in.v[0].w, 0u);
    oct[2] = make_oct<false, L16, R3>(lut, in.v[1].x, in.v[1].y, 0u);
    oct[3] = make_oct<false, L16, R3>(lut, in.v[1].z, in.v[1].w, 0u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        absent |= oct[k].ll >> 16;
        oct[k].ll &= 0xFFFFu;
    }
    tot = oct[0].ll + oct[1].ll + oct[2].ll + oct[3].ll;           // <= 512
    incl = dpp_scan_add(tot);                                      // a wave's total <= 32768
    tail = oct[3].r0;                                              // the last oct has >= 32 bits: its low dword
    if (__any(oct[3].ll < 32u)) {                                  // rare: short codes only (wave-uniform branch)
        u32 v = oct[0].r0;
#pragma unroll
        for (int k = 1; k < 4; ++k) v = oct[k].ll >= 32u ? oct[k].r0 : ((v << (oct[k].ll & 31u)) | oct[k].r0);
        tail = v;
    }
}

// the same for the quad form: the lane's eight quads (a quad of short codes has fewer than 32 bits: the last 32 bits of the
// lane's string always come from several of them)
__device__ __forceinline__ void tile_quads5(const u64 *lut, const TileIn5 &in, Oct (&qd)[8], u32 &tot, u32 &incl, u32 &tail,
                                            u32 &absent)
{
    const u32 w[8] = {in.v[0].x, in.v[0].y, in.v[0].z, in.v[0].w, in.v[1].x, in.v[1].y, in.v[1].z, in.v[1].w};
    tot = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        qd[k] = make_quad4<false>(lut, w[k], 0u);
        absent |= qd[k].ll >> 16;
        qd[k].ll &= 0xFFFFu;
        tot += qd[k].ll;                                           // <= 1024
    }
    incl = dpp_scan_add(tot);                                      // a wave's total <= 65536
    u32 v = qd[0].r0;
#pragma unroll
    for (int k = 1; k < 8; ++k) v = qd[k].ll >= 32u ? qd[k].r0 : ((v << (qd[k].ll & 31u)) | qd[k].r0);
    tail = v;
}

// exclusive offset of this wave and the tile total from the NWV (4 or 16) wave totals: a scan inside one 16-lane DPP row
template <int NWV>
__device__ __forceinline__ u32 tile_offsets5(const u32 *wtot, int lane, int wv, u32 &woff)
{
    const u32 tot = wtot[lane & (NWV - 1)];
    u32 sc = tot;
    sc += (u32)__builtin_amdgcn_update_dpp(0, (int)sc, 0x111, 0xf, 0xf, false);   // row_shr:1
    sc += (u32)__builtin_amdgcn_update_dpp(0, (int)sc, 0x112, 0xf, 0xf, false);   // row_shr:2
    if (NWV > 4) {
        sc += (u32)__builtin_amdgcn_update_dpp(0, (int)sc, 0x114, 0xf, 0xf, false);   // row_shr:4
        sc += (u32)__builtin_amdgcn_update_dpp(0, (int)sc, 0x118, 0xf, 0xf, false);   // row_shr:8
    }
    woff = (u32)__builtin_amdgcn_readlane((int)(sc - tot), wv);
    return (u32)__builtin_amdgcn_readlane((int)sc, NWV - 1);
}

// One oct that ends at window bit e (exclusive), with the partial dword c in front of it (the bits of the dword that
// holds the oct's first bit which belong to earlier octs; 0 when the oct starts on a dword boundary).  Stores the dwords
// [s >> 5, e >> 5) and returns the new partial dword.  NW = dwords an oct of this launch can touch (x0 .. x[NW-1]).
template <int NW>
__device__ __forceinline__ u32 emit_oct(u32 *win, u32 *dump, const Oct &o, u32 e, u32 c)
{
    const u32 p = e >> 5, ps = (e - o.ll) >> 5;
    const u32 j = p - ps;                              // dword boundaries inside (s, e]: 0 .. NW - 1
    const u32 x0 = __builtin_amdgcn_alignbit(o.r0, 0u, e);             // e mod 32 == 0: nothing in dword p
    const u32 x1 = __builtin_amdgcn_alignbit(o.r1, o.r0, e);
    const u32 x2 = __builtin_amdgcn_alignbit(NW >= 4 ? o.r2 : 0u, o.r1, e);
    const u32 x3 = NW >= 4 ? __builtin_amdgcn_alignbit(NW >= 5 ? o.r3 : 0u, o.r2, e) : 0u;
    const u32 x4 = NW >= 5 ? (o.r3 >> (e & 31u)) : 0u;
    u32 v = j >= 1u ? x1 : x0;                         // the dword the oct starts in
    v = j >= 2u ? x2 : v;
    if (NW >= 4) v = j >= 3u ? x3 : v;
    if (NW >= 5) v = j >= 4u ? x4 : v;
    v |= c;
    // Branch-free: a store that is not due goes to the lane's dump word.  (As `if (j >= k) win[..] = ..` the compiler
    // moved every store out of line behind s_cbranch_execnz: two taken branches per store, 24 per lane and tile.)
    u32 *w = win + p;
    *(j >= 1u ? win + ps : dump) = v;
    *(j >= 2u ? w - 1 : dump) = x1;
    if (NW >= 4) *(j >= 3u ? w - 2 : dump) = x2;
    if (NW >= 5) *(j >= 4u ? w - 3 : dump) = x3;
    return j >= 1u ? x0 : v;
}

// The ragged remainder (less than a tile) of every block whose size is not a multiple of the tile: one workgroup per block,
// launched after the chained kernel, so the prefix of the last full tile is final (toff == nullptr), or next to sfe6_kernel,
// whose tile offsets exist before the launch (toff: one u64 per tile, blk.desc_base = the block's first).
template <int NW, bool L16, int NT>
__global__ __launch_bounds__(NT) void sfe4_tail_kernel(const EncBlk *__restrict__ blks, const u64 *__restrict__ desc,
                                                               u32 win_stride, const u32 *__restrict__ redo_only,
                                                               const u64 *__restrict__ toff)
{
    if (redo_only && gload<u32>(redo_only + blockIdx.x) == 0u) return;      // follow-up launch: flagged blocks only
    __shared__ E4Static sh;
    extern __shared__ __attribute__((aligned(16))) u32 dynwin[];
    u32 *win = dynwin + E4_GUARD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NWV = NT / 64;
    constexpr u32 TILE = 32u * NT, TSHIFT = NT == 256 ? 13 : NT == 512 ? 14 : 15;
    const EncBlk blk = blks[blockIdx.x];
    const u32 rem = (u32)(blk.n & (TILE - 1));
    if (!rem) return;
    const u32 nfull = (u32)(blk.n >> TSHIFT);
    const u8 *tb = blk.in + (u64)nfull * TILE;
    if (tid < 256) sh.lut[tid] = gload<u64>((const u64 *)blk.lut + tid);
    for (u32 i = (u32)tid; i < win_stride / 4; i += NT) ((uint4 *)dynwin)[i] = make_uint4(0, 0, 0, 0);
    TileIn in;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32 idx = (u32)k * (8u * NT) + (u32)tid * 8u;
        u32 w[2] = {0, 0};
        if (idx + 8 <= rem) {
            const uint2 x = gload<uint2>(tb + idx);
            w[0] = x.x; w[1] = x.y;
        } else if (idx < rem) {
            const int nv = (int)(rem - idx);
            for (int q = 0; q < nv; ++q) w[q >> 2] |= (u32)gload<u8>(tb + idx + q) << (8 * (q & 3));
        }
        in.v[k] = make_uint2(w[0], w[1]);
    }
    const u32 pv = (tid < 32 && nfull > 0) ? (u32)gload<u8>(tb - 1 - tid) : 0u;
    // bits in front of the ragged tile: the inclusive prefix of the last full tile of the chained kernels, or (sfe6) the
    // tile's offset, known before the launch
    const u64 B = toff ? gload<u64>(toff + blk.desc_base + nfull)
                       : (nfull ? (desc_load(desc + blk.desc_base + nfull - 1) & DESC_VALUE_MASK) : 0ull);
    __syncthreads();
    Oct oct[4];
    u32 incl[2], absent = 0, roff[4];
    tile_octs<true, NW, L16, NT>(sh.lut, in, rem, tid, oct, incl, absent);
    if (absent) set_error_over(blk.err, SHAFA_FILE_UNRECOGNIZABLE, SHAFA_LACK_OF_MEMORY);
    if (lane == 63) {
        sh.wtot[wv] = incl[0] & 0xFFFFu;
        sh.wtot[NWV + wv] = incl[0] >> 16;
        sh.wtot[2 * NWV + wv] = incl[1] & 0xFFFFu;
        sh.wtot[3 * NWV + wv] = incl[1] >> 16;
    }
    if (wv == 0 && ((u32)B & 31u) && nfull > 0) lead_bits(sh.lut, win, pv, (u32)B & 31u, lane);
    __syncthreads();
    const u32 T = tile_offsets<NWV>(sh.wtot, lane, wv, roff);
#pragma unroll
    for (int k = 0; k < 4; ++k) place<NW>(win, oct[k], roff[k] + ((incl[k >> 1] >> (16 * (k & 1))) & 0xFFFFu));
    __syncthreads();
    store_window(win, blk.out, blk.out_cap, blk.err, B, T, true, tid, NT);
    if (tid == 0) gstore<u64>(blk.out_n, (B + T + 7) >> 3);
}

// The ragged remainder of blocks with codes of up to 32 bits (quads): one workgroup per block, launched after the main
// kernel; lanes take 32 consecutive symbols as there, symbols past the block's end encode as nothing, quads are ORed into
// a zeroed window (atomics: a quad may be shorter than a dword here).
template <int NT>
__global__ __launch_bounds__(NT) void sfe5q_tail_kernel(const EncBlk *__restrict__ blks, const u64 *__restrict__ desc,
                                                                u32 win_stride, const u32 *__restrict__ redo_only)
{
    if (redo_only && gload<u32>(redo_only + blockIdx.x) == 0u) return;      // follow-up launch: flagged blocks only
    __shared__ E4Static sh;
    extern __shared__ __attribute__((aligned(16))) u32 dynwin[];
    u32 *win = dynwin + E4_GUARD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NWV = NT / 64;
    constexpr u32 TILE = 32u * NT, TSHIFT = NT == 256 ? 13 : NT == 512 ? 14 : 15;
    const EncBlk blk = blks[blockIdx.x];
    const u32 rem = (u32)(blk.n & (TILE - 1));
    if (!rem) return;
    const u32 nfull = (u32)(blk.n >> TSHIFT);
    const u8 *tb = blk.in + (u64)nfull * TILE;
    if (tid < 256) sh.lut[tid] = gload<u64>((const u64 *)blk.lut + tid);
    for (u32 i = (u32)tid; i < win_stride / 4; i += NT) ((uint4 *)dynwin)[i] = make_uint4(0, 0, 0, 0);
    u32 w[8];
    const u32 base = (u32)tid * 32u;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const u32 idx = base + 4u * (u32)q;
        w[q] = 0;
        if (idx + 4 <= rem) w[q] = gload<u32>(tb + idx);
        else if (idx < rem)
            for (u32 j = 0; j < rem - idx; ++j) w[q] |= (u32)gload<u8>(tb + idx + j) << (8 * j);
    }
    const u32 pv = (tid < 32 && nfull > 0) ? (u32)gload<u8>(tb - 1 - tid) : 0u;
    const u64 B = nfull ? (desc_load(desc + blk.desc_base + nfull - 1) & DESC_VALUE_MASK) : 0ull;   // inclusive prefix of the last full tile
    __syncthreads();
    Oct qd[8];
    u32 tot = 0, absent = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const u32 idx = base + 4u * (u32)q;
        const u32 keep = idx >= rem ? 0u : (rem - idx >= 4u ? 4u : rem - idx);
        qd[q] = make_quad4<true>(sh.lut, w[q], (0xFu << keep) & 0xFu);
        absent |= qd[q].ll >> 16;
        qd[q].ll &= 0xFFFFu;
        tot += qd[q].ll;
    }
    if (absent) set_error_over(blk.err, SHAFA_FILE_UNRECOGNIZABLE, SHAFA_LACK_OF_MEMORY);
    const u32 incl = dpp_scan_add(tot);
    if (lane == 63) sh.wtot5[0][wv] = incl;
    if (wv == 0 && ((u32)B & 31u) && nfull > 0) lead_bits(sh.lut, win, pv, (u32)B & 31u, lane);
    __syncthreads();
    u32 woff;
    const u32 T = tile_offsets5<NWV>(sh.wtot5[0], lane, wv, woff);
    u32 e = woff + incl - tot;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        e += qd[q].ll;
        if (qd[q].ll) place<5>(win, qd[q], e);
    }
    __syncthreads();
    store_window(win, blk.out, blk.out_cap, blk.err, B, T, true, tid, NT);
    if (tid == 0) gstore<u64>(blk.out_n, (B + T + 7) >> 3);
}

}  // namespace
