// sf_encode4.hip — Shannon-Fano bit-pack encoder for codes <= 16 bits in ONE pass over the input
// (compress_to_buffer + binary_coding, reference c.c:52-237): n bytes read + ceil(bits / 8) bytes written,
// which is the kernel's algorithmic HBM traffic.
//
// Why one pass: the three-kernel form (sf_encode3.hip: count, scan, pack) reads the input twice, 2.67 n of traffic on
// Zipf data, and sits at the HBM ceiling of that traffic (a stream that reads n and writes 2n/3 tops out at
// 5.3-5.9 TB/s on this part, tools/ubench).  The prefix problem (where does tile t start in the output?) is solved
// here with a chained scan whose waiting never reaches the critical path:
//
//   * workgroups are persistent and stick to one block at a time (block = blockIdx % nconc, then + nconc): the
//     block's look-up table is loaded once, and with many blocks per launch every block is a separate chain with
//     only a handful of tiles in flight, so a look-back is one 64-entry window;
//   * tiles of a block are handed out by a per-block ticket (atomicAdd), requested three iterations ahead: a tile's
//     predecessors were always taken by workgroups that are running, so the chain cannot deadlock whatever part
//     of the grid is resident (other kernels on other streams included);
//   * iteration i looks up, groups and scans tile i, publishes its bit total (aggregate) and ORs its bit strings
//     into an LDS window at tile-local offsets; the descriptor window of the tile is requested in iteration i+1 (its
//     predecessors hold earlier tickets and have had a whole tile time to publish) and consumed in iteration i+2
//     (a whole tile time to arrive), when the tile's LDS window - one of three - is stored.
//
// Per tile (8 KiB of symbols = four rows of 256 lanes x 8 bytes): every lane turns the 8 symbols of a row into an "oct"
// (<= 128 bits, right-aligned in four dwords) with a tree of shift-or steps on {code, len} pairs read
// from a 2 KiB LDS table with ds_read_b64 (no unpacking); oct bit totals are scanned lane -> wave -> tile (DPP);
// an oct that ENDS at window bit e is ORed into the LDS window with one v_alignbit_b32 + one ds_or_b32 per dword
// (alignbit by e mod 32 needs no special case for 0).  On the way out every output dword is one more alignbit of
// two neighbouring window dwords by (B mod 32), B = the tile's bit offset in the block: byte-swapped, 16-byte
// aligned coalesced stores.  A tile owns the output dwords [B >> 5, E >> 5); the B mod 32 leading bits of its
// first dword are re-encoded from the up to 31 symbols before the tile (each code has >= 1 bit) into the "lead
// word" in front of the window, so every output dword is written exactly once: no global atomics, no pre-zeroed
// output.
#include "common.hpp"
#include "internal.hpp"

#include <mutex>

int g_sfe4_wide = 1;                                   // 0: always the 256-lane form (A/B and tests)
int g_sfe_window_bits = 0;                             // test knob: bits per symbol of the wide form's windows (0: min(Lmax, 12))
int g_sfe_lanes = 0;                                   // 0: widest form that fits; 256 / 512: that workgroup width (A/B and tests)
int g_sfe_variant = 5;                                 // 5: plain-store windows (sfe5_kernel); 4: atomic-OR windows (sfe4_kernel)

namespace {

// A workgroup has NT = 256 or 1024 lanes and a tile is four rows of NT octs: 8 KiB or 32 KiB of symbols.  The wide form
// (launches whose three 32 KiB-tile windows fit a CU's LDS: Lmax <= 12; one workgroup per CU) pays wave 0's chain work
// once per 32 KiB instead of once per 8 KiB and spreads the window stores over fifteen waves instead of three.
constexpr int E4_GUARD = 8;                            // dwords in front of the window: an oct writes up to 4 dwords before its last one
constexpr u32 E4_NONE = 0xFFFFFFFFu;
#ifndef E4_WPS
#define E4_WPS 4                                       // waves per SIMD the register allocation aims at
#endif

#ifdef E4_STAMPS
// diagnostic build only (tools/dbg): per-wave cycle totals of the phases of an iteration
__device__ unsigned long long e4_stamp_buf[2048 * 16 * 8];
#define E4_T0() unsigned long long _t_prev = __builtin_amdgcn_s_memtime(), _t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define E4_T(ph) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); _t_acc[ph] += _t - _t_prev; _t_prev = _t; } while (0)
#define E4_TEND() do { if (lane == 0) for (int _q = 0; _q < 8; ++_q) e4_stamp_buf[((blockIdx.x & 2047) * 16 + wv) * 8 + _q] = _t_acc[_q]; } while (0)
#else
#define E4_T0()
#define E4_T(ph)
#define E4_TEND()
#endif

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

// NW: dwords an oct can touch (3: Lmax <= 8, 4: <= 12, 5: <= 16); L16: Lmax == 16
//
// The FULL tiles of every block.  A three-stage software pipeline per workgroup, one LDS window buffer per stage:
//   iteration i, all waves: tile i is looked up, grouped, scanned and ORed into its window at TILE-LOCAL bit offsets (the
//     output position is not needed for that); tile i-2, whose prefix B is known by now, is funnel-shifted by (B mod 32)
//     on the way out of LDS and stored.
//   wave 0 also runs the chain, and never waits for a round trip doing so: at the top of iteration i it hands over the
//     ticket that was requested in iteration i-1 and requests the next; it resolves the prefix of tile i-2 from the
//     descriptor window it requested in iteration i-1 — a full iteration after that tile's aggregate went out, so its
//     predecessors (earlier tickets) have published theirs unless they lag a whole tile time behind; after barrier A it
//     publishes tile i's aggregate and requests the window of tile i-1.
// Two barriers per tile.  A block's ragged remainder (< 8192 symbols) is left to sfe4_tail_kernel.
template <int NW, bool L16, int NT>
__global__ __launch_bounds__(NT, E4_WPS) void sfe4_kernel(const EncBlk *__restrict__ blks, int nblk, int nconc,
                                                             u64 *__restrict__ desc, u32 *__restrict__ tickets, u32 win_stride)
{
    __shared__ E4Static sh;
    extern __shared__ __attribute__((aligned(16))) u32 dynwin[];     // three buffers of [E4_GUARD][window dwords]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NWV = NT / 64;                       // waves
    constexpr u32 TILE = 32u * NT, TSHIFT = NT == 256 ? 13 : NT == 512 ? 14 : 15;        // symbols per tile
    E4_T0();

    for (int b = (int)(blockIdx.x % (u32)nconc); b < nblk; b += nconc) {
        const EncBlk *bp = blks + b;
        const u8 *in = bp->in;
        const u32 nfull = (u32)(bp->n >> TSHIFT);      // full tiles of this kernel's size
        const bool ragged = (bp->n & (TILE - 1)) != 0;
        u64 *bdesc = desc + bp->desc_base;
        u32 *tick = tickets + bp->ticket;
        __syncthreads();                               // the previous block's table and windows are no longer in use
        if (tid < 256) sh.lut[tid] = gload<u64>((const u64 *)bp->lut + tid);
        if (tid == 0) sh.tick = atomicAdd(tick, 3u);
        __syncthreads();
        // tickets are four deep so that no atomic's round trip is ever waited for: `cur` is processed, `nxt` is being
        // loaded, `nn` is known, and thread 0 holds the request issued one iteration ago (req)
        u32 cur = sh.tick, nxt = cur + 1, nn = cur + 2;
        if (cur >= nfull) continue;
        u32 req = E4_NONE;
        if (tid == 0) req = atomicAdd(tick, 1u);
        TileIn cin, nin, nin2;                         // inputs are requested two tiles ahead
        load_tile<NT>(in, cur, tid, cin);
        if (nxt < nfull) load_tile<NT>(in, nxt, tid, nin);
        // wave 0: the descriptor window and the 32 symbols in front of the tile computed one iteration ago (q) are requested
        // after barrier A and used at the top of the next iteration, when that tile is p: one variable each, never copied
        // (a copy of a loaded register is a wait for the load)
        u64 first_w = 0;
        u32 pv_w = 0;

        u32 q_tile = E4_NONE, q_T = 0;                 // computed in the previous iteration: aggregate out, window requested now
        u32 p_tile = E4_NONE, p_T = 0;                 // computed two iterations ago: resolved and stored now
        u32 dirty[3] = {win_stride - E4_GUARD, win_stride - E4_GUARD, win_stride - E4_GUARD};   // dwords that may be non-zero
        u32 buf = 0;                                   // window buffer of `cur`; q: buf - 1, p: buf - 2 (mod 3)
        bool rotate_in = false;

        for (;;) {
            const bool cur_ok = cur < nfull;
            const bool have_q = q_tile != E4_NONE, have_p = p_tile != E4_NONE;
            if (!cur_ok && !have_q && !have_p) break;
            const u32 pbuf = buf >= 2 ? buf - 2 : buf + 1;
            u32 *win = dynwin + buf * win_stride + E4_GUARD;               // this tile's window
            u32 *pwin = dynwin + pbuf * win_stride + E4_GUARD;             // the window that is stored in this iteration
            Oct c_oct[4];
            u32 incl[2] = {0, 0};                      // rows (0,1) and (2,3): two 16-bit running sums per dword, one DPP scan each

            E4_T(7);
            // ---- wave 0, before it issues anything new: everything it consumes here was requested at least half an
            //      iteration ago (its youngest outstanding memory operation is the descriptor request after barrier A; it
            //      takes no part in the window stores below), so these waits are short --------------------------------
            if (wv == 0) {
                if (lane == 0) sh.tick = req;          // last iteration's ticket request
                E4_T(2);
                if (have_p) {                          // prefix and lead bits of the tile that is stored in this iteration
                    u64 B = 0;
                    if (p_tile > 0) {
                        B = lookback_sum_dpp(bdesc, (int)p_tile, bp->err, first_w);
                        E4_T(0);
                        if (lane == 0) {
                            desc_store(bdesc + p_tile, DESC_PREFIX, B + p_T);
                            if (!ragged && p_tile == nfull - 1) gstore<u64>(bp->out_n, (B + p_T + 7) >> 3);
                        }
                        const u32 r = (u32)B & 31u;
                        if (r) lead_bits(sh.lut, pwin, pv_w, r, lane);
                    }
                    if (lane == 0) sh.prefix = B;
                }
            }
            E4_T(7);

            // ---- this tile: zero its window, look up, group, scan ------------------------------------------------
            if (cur_ok) {
                for (u32 i = (u32)tid; i < ((dirty[buf] + 3u) >> 2) + 1u; i += NT)
                    ((uint4 *)win)[(int)i - 1] = make_uint4(0, 0, 0, 0);              // from dword -4: the lead word is win[-1]
            }
            // the input registers move up one place here, not at the end of the iteration: a wave that loads and stores
            // has to drain its whole memory queue before it may touch a loaded register (loads and stores complete out of
            // order with respect to each other), and here the window stores of the previous iteration are oldest
            if (rotate_in) { cin = nin; nin = nin2; }
            rotate_in = true;
            if (cur_ok && nn < nfull) load_tile<NT>(in, nn, tid, nin2);
            // the next ticket: unconditional (a few tickets past the block's end are harmless), and issued only now, after
            // the wave's one drain of the iteration: the returned value is not touched before the next iteration's top
            if (tid == 0) req = atomicAdd(tick, 1u);
            if (cur_ok) {
                u32 absent = 0;
                tile_octs<false, NW, L16, NT>(sh.lut, cin, 0u, tid, c_oct, incl, absent);
                if (absent) set_error(bp->err, SHAFA_FILE_UNRECOGNIZABLE);   // data symbol without a code (output undefined, in bounds)
                if (lane == 63) {
                    sh.wtot[wv] = incl[0] & 0xFFFFu;
                    sh.wtot[NWV + wv] = incl[0] >> 16;
                    sh.wtot[2 * NWV + wv] = incl[1] & 0xFFFFu;
                    sh.wtot[3 * NWV + wv] = incl[1] >> 16;
                }
            }
            E4_T(1);
            lds_barrier();                                                             // A
            E4_T(3);
            const u32 n3 = sh.tick;
            // waves 1..3 store the resolved tile's window FIRST: their next use of a loaded register (the input rotation in
            // the next iteration) has to wait for these stores, so they go out as early as the iteration allows
            if (wv != 0 && have_p)
                store_window(pwin, bp->out, bp->out_cap, bp->err, sh.prefix, p_T, !ragged && p_tile == nfull - 1, tid - 64,
                             NT - 64);
            u32 c_T = 0;
            if (cur_ok) {
                u32 roff[4];
                c_T = tile_offsets<NWV>(sh.wtot, lane, wv, roff);
                if (tid == 0) {
                    if (cur == 0) {
                        desc_store(bdesc, DESC_PREFIX, c_T);
                        if (!ragged && nfull == 1) gstore<u64>(bp->out_n, ((u64)c_T + 7) >> 3);
                    } else desc_store(bdesc + cur, DESC_AGG, c_T);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)            // the oct ends at: row offset of the wave + inclusive lane prefix
                    place<NW>(win, c_oct[k], roff[k] + ((incl[k >> 1] >> (16 * (k & 1))) & 0xFFFFu));
                dirty[buf] = (c_T >> 5) + 2u;
            }
            E4_T(4);
            if (wv == 0) {
                if (have_q && q_tile > 0) {            // descriptor window and leading symbols of the tile computed one iteration ago
                    const int idx = (int)q_tile - 1 - lane;
                    first_w = desc_load(bdesc + (idx > 0 ? idx : 0));
                    if (lane < 32) pv_w = gload_off<u8>(in + (u64)q_tile * TILE - 32, 31u - (u32)lane);
                }
            }
            E4_T(5);
            lds_barrier();                                                             // B
            E4_T(6);

            // ---- rotate ------------------------------------------------------------------------------------------
            p_tile = q_tile; p_T = q_T;
            q_tile = cur_ok ? cur : E4_NONE; q_T = c_T;
            buf = buf == 2 ? 0 : buf + 1;
            cur = nxt;
            nxt = nn;
            nn = n3;
        }
    }
    E4_TEND();
}

// =====================================================================================================================
// sfe5: the same three-stage pipeline with a window that is filled by PLAIN LDS stores (no atomics, no zeroing).
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
// Per lane and tile: 32 table look-ups and <= 12 exec-masked ds_write_b32 instead of 32 look-ups, 16 ds_or_b32 and the
// window zeroing (sfe4: LDS pipe busy 60 % of the time, half of it bank conflicts of the atomics and look-ups).
// =====================================================================================================================
#ifndef E5_STORE_W0
#define E5_STORE_W0 1                                  // the waves E5_STORE_W0 .. NWV-1 store the resolved window
#endif
struct TileIn5 {
    uint4 v[2];                  // lane t holds bytes [32 t, 32 t + 32) of the tile
};

template <int NT>
__device__ __forceinline__ void load_tile5(const u8 *in, u32 tile, int tid, TileIn5 &t)
{
    const u8 *tb = in + (u64)tile * (32u * NT);
#ifdef E5_LOADTEST                                     // timing experiment only (wrong symbol order): fully coalesced loads
    t.v[0] = gload_nt_off<uint4>(tb, (u32)tid * 16u);
    t.v[1] = gload_nt_off<uint4>(tb, (u32)tid * 16u + 16u * NT);
#elif defined(E5_NTLOAD)
    t.v[0] = gload_nt_off<uint4>(tb, (u32)tid * 32u);
    t.v[1] = gload_nt_off<uint4>(tb, (u32)tid * 32u + 16u);
#else
    // 32 bytes per lane = two 16-byte loads at a 32-byte lane stride: each instruction uses half of every cache line it
    // touches, the other half is the second instruction's.  Default (L1-allocating) policy: with `nt` the second
    // instruction fetches the lines again (measured: 4.05 ms per 8 GiB with nt, 3.72 plain, 3.66 with a fully coalesced
    // but wrong symbol order).
    t.v[0] = gload_off<uint4>(tb, (u32)tid * 32u);
    t.v[1] = gload_off<uint4>(tb, (u32)tid * 32u + 16u);
#endif
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

// cap_bits: what a window holds.  Launches whose longest code lets a tile outgrow the CU's LDS (13..16-bit codes in the
// 1024-lane form) run with windows sized for 12 bits per symbol: a tile that does not fit (the block's rare symbols — every
// one of them at most 2^-13 of the block — would have to fill a whole 32 KiB tile) is not placed, its block is flagged in
// redo[] and encoded again by the 256-lane form (worst-case windows) in a follow-up launch that looks at flagged blocks
// only (redo_only).  redo == nullptr: the windows hold the worst case.
template <int NW, bool L16, int NT, int UNITS = 4>
__global__ __launch_bounds__(NT, E4_WPS) void sfe5_kernel(const EncBlk *__restrict__ blks, int nblk, int nconc,
                                                             u64 *__restrict__ desc, u32 *__restrict__ tickets, u32 win_stride,
                                                             u32 cap_bits, u32 *__restrict__ redo, int redo_only)
{
    __shared__ E4Static sh;
    extern __shared__ __attribute__((aligned(16))) u32 dynwin[];     // three buffers of [E4_GUARD][window dwords]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NWV = NT / 64;                       // waves
    constexpr u32 TILE = 32u * NT, TSHIFT = NT == 256 ? 13 : NT == 512 ? 14 : 15;        // symbols per tile

    for (int b = (int)(blockIdx.x % (u32)nconc); b < nblk; b += nconc) {
        if (redo_only && gload<u32>(redo + b) == 0u) continue;        // follow-up launch: flagged blocks only (uniform)
        const EncBlk *bp = blks + b;
        const u8 *in = bp->in;
        const u32 nfull = (u32)(bp->n >> TSHIFT);      // full tiles of this kernel's size
        const bool ragged = (bp->n & (TILE - 1)) != 0;
        u64 *bdesc = desc + bp->desc_base;
        u32 *tick = tickets + bp->ticket;
        __syncthreads();                               // the previous block's table and windows are no longer in use
        if (tid < 256) sh.lut[tid] = gload<u64>((const u64 *)bp->lut + tid);
        if (tid == 0) sh.tick = atomicAdd(tick, 3u);
        __syncthreads();
        u32 cur = sh.tick, nxt = cur + 1, nn = cur + 2;        // tickets four deep, as in sfe4_kernel
        if (cur >= nfull) continue;
        u32 req = E4_NONE;
        if (tid == 0) req = atomicAdd(tick, 1u);
        // inputs are requested two tiles ahead (octs) or one (quads: their iterations are half as long again and the forty
        // registers of eight quads leave no room for a third input set)
        constexpr bool PF2 = UNITS == 4;
        TileIn5 cin, nin, nin2;
        load_tile5<NT>(in, cur, tid, cin);
        if (PF2 && nxt < nfull) load_tile5<NT>(in, nxt, tid, nin);
        u64 first_w = 0;
        u32 pv_w = 0;

        u32 q_tile = E4_NONE, q_T = 0;                 // computed in the previous iteration: aggregate out, window requested now
        u32 p_tile = E4_NONE, p_T = 0;                 // computed two iterations ago: resolved and stored now
        u32 buf = 0;                                   // window buffer of `cur`; q: buf - 1, p: buf - 2 (mod 3)
        u32 par = 0;                                   // iteration parity: which copy of the hand-over words is this iteration's
        bool rotate_in = false;

        // (Measured and dropped, ms per 8 GiB against 3.34: this loop unrolled three times so that the three input register
        // sets take turns without the sixteen v_mov of the rotation 3.40 - the code no longer fits the instruction cache
        // as well; s_setprio 3 around wave 0's chain work 3.34; odd waves storing the resolved window AFTER they place
        // the new tile 3.51; only waves 8..15 / 12..15 storing 3.37 / 3.47; 512-lane workgroups, two per CU, 3.48; wave 0
        // requesting q's descriptor window at the top and resolving it behind the barrier while the others store 3.44 - a
        // request that early finds predecessors that have not published yet and the look-back polls.)
        for (;;) {
            const bool cur_ok = cur < nfull;
            const bool have_q = q_tile != E4_NONE, have_p = p_tile != E4_NONE;
            if (!cur_ok && !have_q && !have_p) break;
            const u32 pbuf = buf >= 2 ? buf - 2 : buf + 1;
            u32 *win = dynwin + buf * win_stride + E4_GUARD;               // this tile's window
            u32 *pwin = dynwin + pbuf * win_stride + E4_GUARD;             // the window that is stored in this iteration
            Oct c_oct[UNITS];
            u32 tot = 0, incl = 0, tail = 0;

            // ---- wave 0, before it issues anything new (see sfe4_kernel) -----------------------------------------------
            if (wv == 0) {
                if (lane == 0) sh.tick5[par] = req;    // last iteration's ticket request
                if (have_p) {                          // prefix and lead bits of the tile that is stored in this iteration
                    u64 B = 0;
                    if (p_tile > 0) {
                        B = lookback_sum_dpp(bdesc, (int)p_tile, bp->err, first_w);
                        if (lane == 0) {
                            desc_store(bdesc + p_tile, DESC_PREFIX, B + p_T);
                            if (!ragged && p_tile == nfull - 1) gstore<u64>(bp->out_n, (B + p_T + 7) >> 3);
                        }
                        const u32 r = (u32)B & 31u;
                        if (r) lead_bits(sh.lut, pwin, pv_w, r, lane);
                    }
                    if (lane == 0) sh.prefix5[par] = B;
                }
            }

            // ---- this tile: look up, group, scan -----------------------------------------------------------------------
            if (PF2) {
                if (rotate_in) { cin = nin; nin = nin2; }
                if (cur_ok && nn < nfull) load_tile5<NT>(in, nn, tid, nin2);
            } else {
                if (rotate_in) cin = nin;
                if (cur_ok && nxt < nfull) load_tile5<NT>(in, nxt, tid, nin);
            }
            rotate_in = true;
            if (tid == 0) req = atomicAdd(tick, 1u);
            if (cur_ok) {
                u32 absent = 0;
                if constexpr (UNITS == 4) tile_octs5<NW, L16>(sh.lut, cin, c_oct, tot, incl, tail, absent);
                else tile_quads5(sh.lut, cin, c_oct, tot, incl, tail, absent);
                if (absent) set_error(bp->err, SHAFA_FILE_UNRECOGNIZABLE);   // data symbol without a code (output undefined, in bounds)
                if (lane == 63) {
                    sh.wtot5[par][wv] = incl;
                    sh.tail5[par][wv] = tail;
                }
            }
            // The only barrier of the iteration.  Behind it: this tile's window (buffer buf) was last READ by the stores
            // of the previous iteration, the window that is stored now (pbuf) was filled two iterations ago, and the
            // hand-over words of this parity were last read two iterations ago.
            lds_barrier();
            const u32 n3 = sh.tick5[par];
#ifndef E5_ABL_NOSTORE                                 // timing ablations (tools/dbg): wrong output
            if (wv >= E5_STORE_W0 && have_p)
#else
            if (wv >= E5_STORE_W0 && have_p && bp->n == 12345)
#endif
                store_window5(pwin, bp->out, bp->out_cap, bp->err, sh.prefix5[par], p_T, !ragged && p_tile == nfull - 1,
                              tid - 64 * E5_STORE_W0, NT - 64 * E5_STORE_W0);
            u32 c_T = 0;
            if (cur_ok) {
                u32 woff;
                c_T = tile_offsets5<NWV>(sh.wtot5[par], lane, wv, woff);
                if (tid == 0) {
                    if (cur == 0) {
                        desc_store(bdesc, DESC_PREFIX, c_T);
                        if (!ragged && nfull == 1) gstore<u64>(bp->out_n, ((u64)c_T + 7) >> 3);
                    } else desc_store(bdesc + cur, DESC_AGG, c_T);
                }
                const bool fits = c_T <= cap_bits;     // (uniform) else: the tile is not placed and its block is encoded again
                if (!fits && tid == 0 && redo) gstore<u32>(redo + b, 1u);
                const u32 E = woff + incl;             // the lane's string is window bits [E - tot, E)
                u32 e = E - tot;
                // partial dword in front of the lane: the previous lane's last E mod 32 bits (wave_shr:1; lane 0: nothing
                // arrives, the old value 0 stays), lane 0: the previous wave's (a tile starts at bit 0: alignbit by 0 of
                // {x, 0} is 0, whatever tail[] holds)
                const u32 fin = __builtin_amdgcn_alignbit(tail, 0u, E);
                u32 c = (u32)__builtin_amdgcn_update_dpp(0, (int)fin, 0x138, 0xf, 0xf, false);
                if (lane == 0) c = __builtin_amdgcn_alignbit(sh.tail5[par][wv ? wv - 1 : 0], 0u, e);
#ifdef E5_ABL_NOEMIT
                if (bp->n == 12345)
#endif
                if (fits) {
#pragma unroll
                    for (int k = 0; k < UNITS; ++k) {
                        e += c_oct[k].ll;
                        c = emit_oct<NW>(win, sh.dump + lane, c_oct[k], e, c);
                    }
                }
                if (fits && tid == NT - 1) {           // the tile's final partial dword and the zero behind it; the lead word
                    win[e >> 5] = c;                   //   (lead_bits ORs into it two iterations from now)
                    win[(e >> 5) + 1] = 0u;
                    win[-1] = 0u;
                }
            }
            if (wv == 0) {
                if (have_q && q_tile > 0) {            // descriptor window and leading symbols of the tile computed one iteration ago
                    const int idx = (int)q_tile - 1 - lane;
                    first_w = desc_load(bdesc + (idx > 0 ? idx : 0));
                    if (lane < 32) pv_w = gload_off<u8>(in + (u64)q_tile * TILE - 32, 31u - (u32)lane);
                }
            }
            par ^= 1u;
            p_tile = q_tile; p_T = q_T;
            q_tile = cur_ok ? cur : E4_NONE; q_T = c_T;
            buf = buf == 2 ? 0 : buf + 1;
            cur = nxt;
            nxt = nn;
            nn = n3;
        }
    }
}

// The ragged remainder (less than a tile) of every block whose size is not a multiple of the tile: one workgroup per block,
// launched after sfe4_kernel, so the prefix of the last full tile is final.
template <int NW, bool L16, int NT>
__global__ __launch_bounds__(NT) void sfe4_tail_kernel(const EncBlk *__restrict__ blks, const u64 *__restrict__ desc,
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
    const u64 B = nfull ? (desc_load(desc + blk.desc_base + nfull - 1) & DESC_VALUE_MASK) : 0ull;   // inclusive prefix of the last full tile
    __syncthreads();
    Oct oct[4];
    u32 incl[2], absent = 0, roff[4];
    tile_octs<true, NW, L16, NT>(sh.lut, in, rem, tid, oct, incl, absent);
    if (absent) set_error(blk.err, SHAFA_FILE_UNRECOGNIZABLE);
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
    if (absent) set_error(blk.err, SHAFA_FILE_UNRECOGNIZABLE);
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

// the quad form's launcher (codes of 17..32 bits): as e4_launch_nt, windows of lmax_win bits per symbol
template <int NT>
int e5q_launch_nt(hipStream_t st, const EncBlk *dblk, int count, u64 *d_desc, u32 *d_tickets, u32 lmax_win, u32 lmax, bool any_ragged,
                  u32 *d_redo, int redo_only)
{
    constexpr int MAXDEV = 64;
    static int wgs_by_dev_lmax[MAXDEV][33], cus_by_dev[MAXDEV], tail_attr_by_dev[MAXDEV];
    static std::mutex mu;
    const u32 win_stride = ((u32)E4_GUARD + (u32)(((size_t)(32 * NT) * lmax_win) >> 5) + 8u + 3u) & ~3u;   // dwords per buffer
    const u32 tail_stride = ((u32)E4_GUARD + (u32)(((size_t)(32 * NT) * lmax) >> 5) + 8u + 3u) & ~3u;
    const size_t dyn = (size_t)win_stride * 3 * 4;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAXDEV || lmax_win > 32 || lmax > 32) return SHAFA_OUTSIDE_MODULE;
    int wgs_per_cu = 0, cus = 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        int *wgs_by_lmax = wgs_by_dev_lmax[dev];
        if (!wgs_by_lmax[lmax_win]) {
            int occ = 0;
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, dev));
            cus_by_dev[dev] = prop.multiProcessorCount;
            if (dyn > 65536)
                HIP_TRY(hipFuncSetAttribute((const void *)sfe5_kernel<5, false, NT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)sfe5_kernel<5, false, NT, 8>, NT, dyn));
            wgs_by_lmax[lmax_win] = occ < 1 ? 1 : (occ > 6 ? 6 : occ);
        }
        if (any_ragged && (size_t)tail_stride * 4 > 65536 && tail_attr_by_dev[dev] < (int)tail_stride) {
            HIP_TRY(hipFuncSetAttribute((const void *)sfe5q_tail_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)((size_t)tail_stride * 4)));
            tail_attr_by_dev[dev] = (int)tail_stride;
        }
        wgs_per_cu = wgs_by_lmax[lmax_win];
        cus = cus_by_dev[dev];
    }
    int target = cus * wgs_per_cu;
    int nconc = count < target ? count : target;
    int per = target / nconc;
    if (per < 1) per = 1;
    const u32 cap_bits = (u32)(32u * NT) * lmax_win;
    hipLaunchKernelGGL((sfe5_kernel<5, false, NT, 8>), dim3((u32)(nconc * per)), dim3(NT), dyn, st, dblk, count, nconc, d_desc, d_tickets,
                       win_stride, cap_bits, d_redo, redo_only);
    if (any_ragged)
        hipLaunchKernelGGL((sfe5q_tail_kernel<NT>), dim3((u32)count), dim3(NT), (size_t)tail_stride * 4, st, dblk,
                           (const u64 *)d_desc, tail_stride, (const u32 *)(redo_only ? d_redo : nullptr));
    return SHAFA_SUCCESS;
}

// lmax_win: bits per symbol the three windows are sized for (the launch's longest code, or less: see sfe5_kernel's
// cap_bits); lmax: the launch's longest code (the tail kernel's one window always holds the worst case)
template <int NW, bool L16, int NT>
int e4_launch_nt(hipStream_t st, const EncBlk *dblk, int count, u64 *d_desc, u32 *d_tickets, u32 lmax_win, u32 lmax, bool any_ragged,
                 u32 *d_redo, int redo_only)
{
    // per device: the LDS attribute below belongs to the device's copy of the kernel, and devices may differ in CUs.
    // (Statics of a template: one set per instantiation.  Guarded: pipes on several host threads launch concurrently.)
    constexpr int MAXDEV = 64;
    static int wgs_by_dev_lmax[MAXDEV][17], cus_by_dev[MAXDEV], tail_attr_by_dev[MAXDEV];
    static std::mutex mu;
    const u32 win_stride = ((u32)E4_GUARD + (u32)(((size_t)(32 * NT) * lmax_win) >> 5) + 8u + 3u) & ~3u;   // dwords per buffer
    const u32 tail_stride = ((u32)E4_GUARD + (u32)(((size_t)(32 * NT) * lmax) >> 5) + 8u + 3u) & ~3u;
    const size_t dyn = (size_t)win_stride * 3 * 4;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAXDEV || lmax_win > 16 || lmax > 16) return SHAFA_OUTSIDE_MODULE;
    int wgs_per_cu_v = 0, cus = 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        int *wgs_by_lmax = wgs_by_dev_lmax[dev];
        if (!wgs_by_lmax[lmax_win]) {
            int occ = 0;
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, dev));
            cus_by_dev[dev] = prop.multiProcessorCount;
            if (dyn > 65536) {                         // more than the default 64 KiB of dynamic LDS per workgroup
                HIP_TRY(hipFuncSetAttribute((const void *)sfe4_kernel<NW, L16, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
                HIP_TRY(hipFuncSetAttribute((const void *)sfe5_kernel<NW, L16, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            }
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)sfe4_kernel<NW, L16, NT>, NT, dyn));
            wgs_by_lmax[lmax_win] = occ < 1 ? 1 : (occ > 6 ? 6 : occ);   // residency is a matter of speed only (tickets), 6 = what the registers allow
        }
        if (any_ragged && (size_t)tail_stride * 4 > 65536 && tail_attr_by_dev[dev] < (int)tail_stride) {
            HIP_TRY(hipFuncSetAttribute((const void *)sfe4_tail_kernel<NW, L16, NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)((size_t)tail_stride * 4)));
            tail_attr_by_dev[dev] = (int)tail_stride;
        }
        wgs_per_cu_v = wgs_by_lmax[lmax_win];
        cus = cus_by_dev[dev];
    }
    const int wgs_per_cu = wgs_per_cu_v;
    // the grid is a multiple of the number of concurrently served blocks, so a workgroup stays with one block
    int target = cus * wgs_per_cu;
    int nconc = count < target ? count : target;
    int per = target / nconc;
    if (per < 1) per = 1;
    const u32 cap_bits = (u32)(32u * NT) * lmax_win;
    if (g_sfe_variant == 5 || d_redo)
        hipLaunchKernelGGL((sfe5_kernel<NW, L16, NT>), dim3((u32)(nconc * per)), dim3(NT), dyn, st, dblk, count, nconc, d_desc, d_tickets,
                           win_stride, cap_bits, d_redo, redo_only);
    else
        hipLaunchKernelGGL((sfe4_kernel<NW, L16, NT>), dim3((u32)(nconc * per)), dim3(NT), dyn, st, dblk, count, nconc, d_desc, d_tickets,
                           win_stride);
    if (any_ragged)
        hipLaunchKernelGGL((sfe4_tail_kernel<NW, L16, NT>), dim3((u32)count), dim3(NT), (size_t)tail_stride * 4, st, dblk,
                           (const u64 *)d_desc, tail_stride, (const u32 *)(redo_only ? d_redo : nullptr));
    return SHAFA_SUCCESS;
}

constexpr u32 E5_CAP_LMAX = 12;                        // the 1024-lane form's windows hold at most this many bits per symbol

__host__ u32 e5_window_bits(u32 lmax)                  // bits per symbol the wide form's windows are sized for
{
    u32 w = lmax < E5_CAP_LMAX ? lmax : E5_CAP_LMAX;
    if (g_sfe_window_bits > 0 && (u32)g_sfe_window_bits < w) w = (u32)g_sfe_window_bits;    // test knob: provoke the re-encode path
    return w;
}

// the wide form — ONE workgroup of 1024 lanes per CU, 32 KiB tiles — with three windows that fit the CU's 160 KiB next to
// the 2.5 KiB of static LDS: sized for the launch's longest code up to 12 bits, for 12 bits per symbol beyond (13..16-bit
// codes: with the flag-and-encode-again fall-back of sfe5_kernel, which needs the second chain `x`)
template <int NW, bool L16>
int e4_launch_t(hipStream_t st, const EncBlk *dblk, int count, u64 *d_desc, u32 *d_tickets, u32 lmax, u32 ragged, const SfeRedo &x)
{
    const bool ragged8 = ragged & 1u, ragged16 = ragged & 2u, ragged32 = ragged & 4u;
    if (g_sfe_lanes == 512 && g_sfe_variant == 5)
        return e4_launch_nt<NW, L16, 512>(st, dblk, count, d_desc, d_tickets, lmax, lmax, ragged16, nullptr, 0);
    if (g_sfe_lanes == 256 || !g_sfe4_wide)
        return e4_launch_nt<NW, L16, 256>(st, dblk, count, d_desc, d_tickets, lmax, lmax, ragged8, nullptr, 0);
    const u32 wbits = e5_window_bits(lmax);
    if (wbits == lmax) return e4_launch_nt<NW, L16, 1024>(st, dblk, count, d_desc, d_tickets, lmax, lmax, ragged32, nullptr, 0);
    if (g_sfe_variant != 5 || !x.redo)                 // the atomic-OR form has no fall-back: its 256-lane form holds the worst case
        return e4_launch_nt<NW, L16, 256>(st, dblk, count, d_desc, d_tickets, lmax, lmax, ragged8, nullptr, 0);
    int rc = e4_launch_nt<NW, L16, 1024>(st, dblk, count, d_desc, d_tickets, wbits, lmax, ragged32, x.redo, 0);
    if (rc) return rc;
    return e4_launch_nt<NW, L16, 256>(st, dblk, count, x.desc2, x.tickets2, lmax, lmax, ragged8, x.redo, 1);
}

}  // namespace

#ifdef E4_STAMPS
extern "C" int shafa_e4_read_stamps(unsigned long long *dst, int n)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(e4_stamp_buf), (size_t)n * 8) == hipSuccess ? 0 : 9;
}
#endif

// launched from sfenc_launch (sf_encode.hip) for blocks whose codes are <= 16 bits when the launch holds enough blocks
// to keep every chain short; desc (one u64 per tile) and tickets (one u32 per block) are zeroed by the caller;
// tables are 256 x u64 {code, len}
// does a launch with this longest code run the wide form with windows smaller than its worst case (then it needs `x`)?
bool sfenc4_needs_redo(u32 lmax)
{
    return g_sfe_variant == 5 && g_sfe4_wide && g_sfe_lanes == 0 && e5_window_bits(lmax) < lmax;
}

// can launches of codes of 17..32 bits take the one-pass encoder (quad form)?  It always runs with windows smaller than
// its worst case, so it needs the second chain.
bool sfenc4_long_ok() { return g_sfe_variant == 5 && g_sfe4_wide && g_sfe_lanes == 0; }

// blocks whose codes are 17..32 bits (tables 256 x u64 {code, len}, len = 1 << 16 for a symbol without a code)
int sfenc4_launch_long(hipStream_t st, const EncBlk *dblk, int count, u64 *d_desc, u32 *d_tickets, u32 lmax, u32 ragged, const SfeRedo &x)
{
    if (!x.redo) return SHAFA_OUTSIDE_MODULE;
    const u32 wbits = e5_window_bits(lmax);
    int rc = e5q_launch_nt<1024>(st, dblk, count, d_desc, d_tickets, wbits, lmax, (ragged & 4u) != 0, x.redo, 0);
    if (rc) return rc;
    return e5q_launch_nt<256>(st, dblk, count, x.desc2, x.tickets2, lmax, lmax, (ragged & 1u) != 0, x.redo, 1);
}

int sfenc4_launch(hipStream_t st, const EncBlk *dblk, int count, u64 *d_desc, u32 *d_tickets, u32 lmax, u32 ragged, const SfeRedo &x)
{
    if (lmax <= 8) return e4_launch_t<3, false>(st, dblk, count, d_desc, d_tickets, lmax, ragged, x);
    if (lmax <= 12) return e4_launch_t<4, false>(st, dblk, count, d_desc, d_tickets, lmax, ragged, x);
    if (lmax <= 15) return e4_launch_t<5, false>(st, dblk, count, d_desc, d_tickets, lmax, ragged, x);
    return e4_launch_t<5, true>(st, dblk, count, d_desc, d_tickets, lmax, ragged, x);
}
