// sf_decode.hip — Module D hot path: Shannon-Fano (prefix-code) decode on gfx950.
//
// Replaces create_tree + shafa_block_decompressor (reference d.c:466-551): a bit-serial tree walk that stops after
// block_size symbols.  The .shaf format has no sync markers, so decoding a 64 MiB block in parallel needs the codeword
// boundaries first:   next(p) = p + len(code that starts at bit p)   is a function graph over the stream's bit positions,
// the true boundaries are the path from bit 0.  The stream is cut into 256-bit chunks (256 chunks = one 8 KiB tile); a
// chunk's ENTRY is the offset of the first code that starts in it (< Lmax), its COUNT the number of codes that start in it.
// Every launch goes   entries + counts  ->  sfd_offsets (scan of the tile counts)  ->  symbol pass:
//
//   code class (per launch)                 entries + counts                                   symbols
//   complete, Lmax <= 16 (Module T's)       sfd_scan + sfd_ends (speculative, verified exactly; here)   sfd_wstage (here)
//                                           else exact: sfd_sync16 .. sfd_countfsm (sfd_dp.hpp)
//   complete, 16 < Lmax <= 32               sfd_scan<2> + sfd_ends<2>, else sfd_sync32 / sfd_countfsm32 sfd_wstage<2>
//   anything else (hand-made .cod)          sfd_sync / sfd_tiles / sfd_count (sfd_generic.hpp)          sfd_write
//
// A speculating launch is eight kernels: sfd_tables, sfd_scan (every 16 KiB unit but the ones a stream ends in), sfd_ends (those
// units, the links between all units, the verdict per block), the three exact kernels (fat grids that return at once for blocks
// that verified), sfd_offsets, sfd_wstage.
// This file: the kernels the time goes to on ordinary data (sfd_scan, sfd_wstage), sfd_ends, sfd_offsets, and the launcher.
// Algorithmic HBM bytes per block: sf_n read + n_symbols written.
#include <type_traits>
#include "common.hpp"
#include "internal.hpp"

#include <atomic>
#include <mutex>
#include <thread>

#include <algorithm>

#include <stdlib.h>

#include "sfd_common.hpp"
#include "sfd_dp.hpp"
#include "sfd_generic.hpp"

namespace {

// ================================================================================================
// Speculative chunk entries for self-synchronising codes (complete codes, Lmax <= 32), verified exactly.
//
// The packed DP above prices every bit position (6.7 VALU per bit) to be exact for ANY code.  Shannon-Fano codes of
// skewed data re-synchronise: a decoder started at an arbitrary bit agrees with the true parse after a few dozen bits
// (measured: started 256 bits early it is off at the chunk boundary in 0.2 % of the chunks of Zipf(1.2) mod 256 data,
// 0.02 % for Zipf truncated to 256 ranks, 85 % for uniform bytes, whose 8/9-bit codes never merge).  So a lane
//   1. walks the 256 bits in front of its strip from their first bit (three codes per look-up), which leaves it at its
//      guess of the strip's entry, then walks the strip, noting entry and code count of each chunk, to its exit;
//   2. compares its guess with the exit of the lane before it; lanes that differ take that exit as their entry and walk
//      again until they are back on their earlier path.
// The first lane of a wave is compared with the wave before it by sfd_ends, which walks the few chunks that differ again.  When
// every comparison of a block holds — entry(c) == exit(c-1) for all chunks, entry 0 at the block's first bit — the
// entries ARE the true parse, by induction: nothing is approximate.  A block that does not get there in the fixed number
// of rounds sets *run_dp and goes through the exact kernels, which skip the blocks that verified; which blocks try at all
// is decided per table on the host (spec_worthwhile()).  Outputs are those of sfd_countfsm.
// ================================================================================================
typedef __attribute__((address_space(3))) u32 lds_u32;
typedef __attribute__((address_space(3))) u16 lds_u16;
typedef __attribute__((address_space(3))) u8 lds_u8;
// entries / exits of a chunk are < 16 (< 32 for LONG == 2: codes of up to 32 bits); the value itself marks "the stream
// ended in front of this chunk"
template <int LONG> constexpr u32 spec_emask() { return LONG == 2 ? 31u : 15u; }

// ================================================================================================
// sfd_scan: the speculative entries, shaped for the way gfx950's LDS and vector ALUs price the walk.
// Outputs: entry and code count of every 256-bit chunk, tile counts, a guess / exit pair per unit for sfd_ends.
//   * a lane owns a STRIP of SC_SB = 256 bytes (8 chunks) and walks it front to back with its exact position carried
//     from chunk to chunk, so only the strip's first entry is a guess: the 256-bit run-up is paid once per 2048 bits;
//   * the strip passes through LDS one 64-byte PHASE at a time (17 words per lane, as before), each lane's words in
//     its own COLUMN (word i of lane c at dword i * 256 + c): a lane only ever reads bank c % 32, so the stream
//     reads are conflict free whatever word each lane is at (rows 17 words apart met on a bank whenever two lanes 2,
//     15 or 17 apart were 2 or 1 words apart: ds_read_b32 banks are (a / 4) % 32 within each half wave);
//   * a wave loads, walks and verifies its own 64 strips (16 KiB of stream): nothing in the walk waits for another wave,
//     there is no workgroup barrier after the table fill;
//   * the phase's words are fetched by lane PAIRS (32 contiguous bytes per pair and instruction, two lanes per column
//     and store: a 2-way store conflict is free), one phase ahead of the walk.
// A phase's row holds stream words 16 k - 1 .. 16 k + 15 of the strip (word 0 = the previous phase's last word, carried
// over), so a window never needs a word that is not there: a fetch at row bit q reads words q >> 5 and (q >> 5) + 1 <= 16.
// Row bit 32 is the strip's bit 512 k (a chunk boundary), 288 the next one, 544 = the next phase's 32.  The walk of a
// phase: finish the chunk in progress (single codes up to row bit 32), walk chunk 2 k (three codes per look-up, then
// single codes up to 288), start chunk 2 k + 1 (three codes per look-up while the fetch stays inside the row: q < 512).
// The run-up is the same last step on a row that holds the 32 bytes in front of the strip.
// Lanes whose guess differs from the exit of the lane before them walk again from that exit until they are back on
// their first path (scan_strip_redo: the wave reloads the phases, only those lanes walk); the first lane of a wave
// is compared with the wave before it by sfd_ends (sf_links), which walks the unit's first chunks again from the true entry.
// ================================================================================================
constexpr int SC_PHW = 16;                          // stream words per phase and strip (64 bytes)
constexpr int SC_M = 4;                             // phases per strip
constexpr int SC_SB = SC_PHW * 4 * SC_M;            // bytes per strip
constexpr int SC_CH = SC_SB / CH_BYTES;             // chunks per strip
constexpr int SC_STRIPS_TILE = DTILE / SC_SB;       // strips per tile (32: half a wave)
constexpr int SC_WTILES = 64 / SC_STRIPS_TILE;      // tiles per wave (2): the "unit" of the links sfd_ends checks
constexpr int SC_TILES = DEC_THREADS / SC_STRIPS_TILE;   // tiles per workgroup (8)
constexpr int SC_ROWW = SC_PHW + 1;                 // row words: the carried word, then the phase's 16
constexpr int SC_COLB = DEC_THREADS * 4;            // bytes between two words of a column
constexpr int SC_LDS_ROWS = SC_ROWW * SC_COLB;
constexpr int SC_MISC = 64;                         // flags of wg_any
static_assert(SC_COLB == 1024 && SC_WTILES == 2 && SC_STRIPS_TILE == 32, "sfd_scan's lane maps");

struct ScanWin {
    u32 lo, hi, wa, cb;                                 // words at LDS address wa, wa + SC_COLB; cb: the lane's column
    __device__ __forceinline__ void init(u32 col) { cb = col; wa = 0xFFFFFFFFu; hi = lo = 0; }
    __device__ __forceinline__ void flush() { wa = 0xFFFFFFFFu; }      // the row was rewritten
    __device__ __forceinline__ u32 at(u32 q)            // the 32 stream bits from row bit q on
    {
        const u32 a = cb + ((q >> 5) << 10);
        if (a != wa) {
            lo = *(const lds_u32 *)(size_t)a;
            hi = *(const lds_u32 *)(size_t)(a + SC_COLB);
            wa = a;
        }
        return __builtin_amdgcn_alignbit(hi, lo, q);
    }
};

template <int LONG>
__device__ __forceinline__ u32 scan_long_len(const u16 *lt, ScanWin &sw, const u32 qq)
{
    const u32 win = __builtin_bitreverse32(sw.at(qq));
    const u32 l = (LONG == 1 ? long_code(lt, win) : long_code32(lt, win)) >> 8;
    return l ? l : 1u;                                  // at least 1: the walk must move
}

// whole fetches (two or three look-ups of up to seven codes each) from row bit q while every code taken ends by `qe`
// and the fetch (and a long code's window behind it) stays inside the row; counts the codes started on the way.
// An entry is bits | codes << 5 and a fetch uses at most 30 bits, so the running sum of a fetch's entries has the bits used
// so far in its low five bits — exactly what v_bfe_u32 takes as its shift: one v_bfe per window, one add per look-up.
template <int LONG>
__device__ __forceinline__ void scan_multi(const lds_u8 *tab, const u16 *lt, const u32 KW, u32 &q, const u32 qe, u32 &cnt, ScanWin &sw)
{
    const u32 q0 = q;
    u32 acc = 0;                                        // sum of the entries = bits walked + 32 * codes
    auto multi = [&](auto nlook) {
        constexpr u32 N = decltype(nlook)::value;
        const u32 in_row = 32u * SC_PHW - 1u - (LONG ? (N - 1u) * KW : 0u);
        const u32 qs = qe - N * KW < in_row ? qe - N * KW : in_row;
        while (q <= qs) {
            const u32 w = sw.at(q);
            u32 sum = 0, e = 0;
#pragma unroll
            for (u32 i = 0; i < N; ++i) {
                e = tab[__builtin_amdgcn_ubfe(w, sum, KW)];
                sum += e;
            }
            acc += sum;
            q += sum & 31u;
            if (LONG && __builtin_expect((e & 31u) == 0u, 0)) {      // a long code stopped the look-ups: it starts at q
                const u32 l = scan_long_len<LONG>(lt, sw, q);
                q += l;
                acc += l + 32u;
            }
        }
    };
    if (KW <= 10) multi(std::integral_constant<u32, 3>{});
    else multi(std::integral_constant<u32, 2>{});
    cnt += (acc - (q - q0)) >> 5;
}

// single codes from row bit q to the first code start >= qe.  LAST: the stream ends at row bit `qlimit` (may lie in
// front of q); a code that does not end inside it is not a symbol and nothing starts after it: returns true ("cut")
template <bool LAST, int LONG>
__device__ __forceinline__ bool scan_single(const lds_u8 *tab, const u16 *lt, const u32 KW, u32 &q, const u32 qe, const int qlimit,
                                            u32 &cnt, ScanWin &sw)
{
    const u32 mask = (1u << KW) - 1u;
    const lds_u8 *len0 = tab + (1u << KW);
    while (q < qe) {
        u32 l0 = len0[sw.at(q) & mask];
        if (LONG && __builtin_expect(l0 == 0u, 0)) l0 = scan_long_len<LONG>(lt, sw, q);
        if (LAST && (int)(q + l0) > qlimit) return true;
        q += l0;
        ++cnt;
    }
    return false;
}

// What a wave's lanes share while they load and walk 64 strips from byte `wave_off` of the block's stream (lane = strip =
// column).  The phase's words are fetched by lane PAIRS: pair px loads strips px and px + 32, 32 contiguous bytes per
// pair and load, and stores them into those strips' columns.
#define SC_QUAD 0
#define SC_WAVES 6                                  // waves per SIMD the register allocation aims at (71 registers: seven fit)
#define SC_NT 0
struct ScanIO {
    const u8 *in;
    u64 in_n;
    long long grp_off;                                  // the lane group's first byte of phase 0, first strip
    u32 wbase, ph, cb;
    // lanes per strip and load (SC_QUAD: four lanes fetch the 64 bytes of a strip's phase as one contiguous piece — one
    // request — and store into one column: a 4-way store conflict, twice the cycles of ds_write_b32; else pairs, 32 bytes
    // per request, 2-way stores = free)
    static constexpr u32 G = SC_QUAD ? 4u : 2u, SPL = 64u / G;        // strips per load instruction
    __device__ __forceinline__ void init(const DecBlk &blk, const u64 wave_off)
    {
        const u32 tid = threadIdx.x, lane = tid & 63u, px = lane / G;
        in = blk.in;
        in_n = blk.in_n;
        ph = lane % G;
        grp_off = (long long)wave_off + (long long)px * SC_SB + 16ll * ph;
        wbase = 4u * ((tid & ~63u) + px);
        cb = 4u * tid;
    }
    // 16 stream bytes at `off`; CHECKED: zeros outside the stream (the piece the stream ends in byte by byte, rolled:
    // one per block)
    template <bool CHECKED>
    __device__ __forceinline__ uint4 fetch16(const long long off) const
    {
        if (!CHECKED || (off >= 0 && (u64)off + 16 <= in_n)) return SC_NT ? gload_nt<uint4>(in + off) : gload<uint4>(in + off);
        u32 w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        if (off >= 0 && (u64)off < in_n) {
            const int nv = (int)(in_n - (u64)off);
#pragma clang loop unroll(disable)
            for (int b = 0; b < nv; ++b) {
                const u32 v = (u32)gload<u8>(in + off + b) << (8 * (b & 3));
                const int wi = b >> 2;
                w0 |= wi == 0 ? v : 0u;
                w1 |= wi == 1 ? v : 0u;
                w2 |= wi == 2 ? v : 0u;
                w3 |= wi == 3 ? v : 0u;
            }
        }
        return make_uint4(w0, w1, w2, w3);
    }
    // load t of a phase: strips px + SPL * sgrp(t), 16 bytes at piece(t) of the phase's 64
    static __device__ __forceinline__ constexpr u32 sgrp(const int t) { return SC_QUAD ? (u32)t : (u32)t >> 1; }
    static __device__ __forceinline__ constexpr u32 piece(const int t) { return SC_QUAD ? 0u : 32u * ((u32)t & 1u); }
    template <bool CHECKED>
    __device__ __forceinline__ void load_phase(const int k, uint4 (&R)[4]) const
    {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            R[t] = fetch16<CHECKED>(grp_off + (long long)sgrp(t) * SPL * SC_SB + 64ll * k + piece(t));
    }
    // phases 2 j and 2 j + 1 at once: the two halves of every strip's 128-byte line are asked for back to back (a phase
    // alone asks for half a line, and the other half — a walk later — finds the line evicted: every line fetched twice,
    // tools/ubench/fetch_calib.hip)
    template <bool CHECKED>
    __device__ __forceinline__ void load_line(const int j, uint4 (&R)[8]) const
    {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = 2 * g + u;
                    R[4 * h + t] = fetch16<CHECKED>(grp_off + (long long)sgrp(t) * SPL * SC_SB + 64ll * (2 * j + h) + piece(t));
                }
    }
    // the 32 bytes in front of every strip (the block's first strip has none: it is never guessed; reads its own head)
    static constexpr int NRU = SC_QUAD ? 4 : 2;        // loads of the run-up (SC_QUAD: half of the lanes idle)
    template <bool CHECKED>
    __device__ __forceinline__ void load_runup(uint4 (&RU)[NRU]) const
    {
#pragma unroll
        for (int t = 0; t < NRU; ++t) {
            const long long o = grp_off - 32ll - (SC_QUAD ? 16ll * (ph & 2u) : 0ll) + (long long)t * SPL * SC_SB;
            RU[t] = fetch16<CHECKED>(o < 0 ? o + 32 : o);
        }
    }
    // row word r of the column of strip px + SPL * xhi
    __device__ __forceinline__ void put(const u32 r, const u32 xhi, const u32 v) const
    {
        *(lds_u32 *)(size_t)(wbase + r * SC_COLB + 4u * SPL * xhi) = rev_bytes(v);
    }
    __device__ __forceinline__ void wave_sync() const  // LDS operations of one wave execute in order: only the compiler is told
    {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void put_runup(const uint4 (&RU)[NRU]) const    // row words 9..16
    {
#pragma unroll
        for (int t = 0; t < NRU; ++t) {
            const u32 r = 9u + 4u * (ph & 1u);
            if (!SC_QUAD || ph < 2u) {
                put(r, t, RU[t].x);
                put(r + 1, t, RU[t].y);
                put(r + 2, t, RU[t].z);
                put(r + 3, t, RU[t].w);
            }
        }
        wave_sync();
    }
    // the row of a phase: word 0 = the old row's last word, words 1..16 = the phase
    __device__ __forceinline__ void put_phase(const uint4 *R) const
    {
        const u32 carry = *(const lds_u32 *)(size_t)(cb + 16u * SC_COLB);
        wave_sync();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32 r = 1u + 4u * ph + piece(t) / 4u;
            put(r, sgrp(t), R[t].x);
            put(r + 1, sgrp(t), R[t].y);
            put(r + 2, sgrp(t), R[t].z);
            put(r + 3, sgrp(t), R[t].w);
        }
        *(lds_u32 *)(size_t)cb = carry;
        wave_sync();
    }
    // The strip's last chunk ends in the NEXT strip's first word.  Asked for at the end of the walk it costs a line per strip
    // (that line was loaded — by the lanes that load the next strip — when the walk began: long evicted).  So it is taken at
    // the beginning, from those lanes' registers: every loader lane that holds word 0 of a strip puts it (bit-reversed) into
    // row word 0 of the column in front of it, and every walker keeps its own.  The wave's last strip has its successor in
    // another wave: one 4-byte load.  (R: the registers of load_line(0) / load_phase(0).)
    __device__ __forceinline__ u32 next_strip_word(const uint4 *R, const u64 wave_off) const
    {
        const u32 lane = threadIdx.x & 63u;
        if (ph == 0) {                                  // this lane holds word 0 of strips px (R[0]) and px + SPL (R[2])
            if (lane) *(lds_u32 *)(size_t)(wbase - 4u) = rev_bytes(R[0].x);
            *(lds_u32 *)(size_t)(wbase - 4u + 4u * SPL) = rev_bytes(R[2].x);
        }
        wave_sync();
        u32 nw = *(const lds_u32 *)(size_t)cb;
        if (lane == 63) nw = rev_bytes(gload<u32>(in + wave_off + (u64)64 * SC_SB));
        wave_sync();
        return nw;
    }
    __device__ __forceinline__ void put_next_word(const u32 nw_rev) const
    {
        const u32 carry = *(const lds_u32 *)(size_t)(cb + 16u * SC_COLB);
        *(lds_u32 *)(size_t)cb = carry;
        *(lds_u32 *)(size_t)(cb + SC_COLB) = nw_rev;
    }
    // the same from memory (the rolled walk: bounded loads near the stream's end)
    template <bool CHECKED>
    __device__ __forceinline__ void put_next_strip(const u64 wave_off) const
    {
        u32 nw = 0;
        const long long off = (long long)wave_off + (long long)((threadIdx.x & 63u) + 1u) * SC_SB;
        if (!CHECKED || (u64)off + 4 <= in_n) nw = gload<u32>(in + off);
        else {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if ((u64)off + b < in_n) nw |= (u32)gload<u8>(in + off + b) << (8 * b);
        }
        const u32 carry = *(const lds_u32 *)(size_t)(cb + 16u * SC_COLB);
        *(lds_u32 *)(size_t)cb = carry;
        *(lds_u32 *)(size_t)(cb + SC_COLB) = rev_bytes(nw);
    }
};

// entries (a byte each) and counts (16 bits each) of a strip's eight chunks, packed as they are stored
struct ScanOut {
    u32 e[2], c[4];
    __device__ __forceinline__ void clear() { e[0] = e[1] = 0; c[0] = c[1] = c[2] = c[3] = 0; }
    // compile-time index, fields still zero: one v_lshl_or each
    template <int J> __device__ __forceinline__ void put_ent(const u32 v) { e[J >> 2] |= v << (8 * (J & 3)); }
    template <int J> __device__ __forceinline__ void put_cnt(const u32 v) { c[J >> 1] |= v << (16 * (J & 1)); }
    // run-time index (the rolled walk below)
    __device__ __forceinline__ u32 ent(const u32 j) const { return ((j < 4 ? e[0] : e[1]) >> (8u * (j & 3u))) & 0xFFu; }
    __device__ __forceinline__ void set_ent(const u32 j, const u32 v)
    {
        const u32 sh = 8u * (j & 3u), m = ~(0xFFu << sh), x = v << sh;
        if (j < 4) e[0] = (e[0] & m) | x; else e[1] = (e[1] & m) | x;
    }
    __device__ __forceinline__ void set_cnt(const u32 j, const u32 v)
    {
        const u32 sh = 16u * (j & 1u), m = ~(0xFFFFu << sh), x = v << sh, i = j >> 1;
        c[0] = i == 0 ? (c[0] & m) | x : c[0];
        c[1] = i == 1 ? (c[1] & m) | x : c[1];
        c[2] = i == 2 ? (c[2] & m) | x : c[2];
        c[3] = i == 3 ? (c[3] & m) | x : c[3];
    }
    __device__ __forceinline__ u32 total() const
    {
        u32 t = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) t += (c[i] & 0xFFFFu) + (c[i] >> 16);
        return t;
    }
};
static_assert(SC_CH == 8, "ScanOut packs eight chunks");

// The walk of a wave whose strips lie inside the stream with room to spare (all but a block's last wave): no bounds, no
// stream end, every lane walks; the phases unrolled, entries and counts in registers.
template <int LONG>
__device__ __forceinline__ void scan_strip_fast(const lds_u8 *tab, const u16 *lt, const u32 KW, const DecBlk &blk, const u64 wave_off,
                                                const u32 e_forced, const bool forced, ScanOut &o, u32 &exit_)
{
    o.clear();
    ScanIO io;
    io.init(blk, wave_off);
    ScanWin sw;
    sw.init(io.cb);
#ifndef SC_LINE
#define SC_LINE 1
#endif
    uint4 R[SC_LINE ? 8 : 4];
    u32 q, pc = 0;
    // the chunk in progress up to where the row ends (q >= 512 afterwards): the last step of a phase, and the whole run-up
    auto to_row_end = [&]() {
        scan_multi<LONG>(tab, lt, KW, q, 32u * SC_PHW + 32u, pc, sw);
        if (LONG) scan_single<false, LONG>(tab, lt, KW, q, 32u * SC_PHW, 0, pc, sw);
    };
    {
        uint4 RU[ScanIO::NRU];
        io.load_runup<false>(RU);
#if SC_LINE
        io.load_line<false>(0, R);
#else
        io.load_phase<false>(0, R);
#endif
        io.put_runup(RU);
    }
    static_assert(!SC_QUAD, "next_strip_word: pair mapping");
    const u32 nextw = io.next_strip_word(R, wave_off);  // (row word 0 is free until the first phase's carry goes there)
    q = 32u * 9u;
    to_row_end();
    q -= 32u * SC_PHW;
    auto phase = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
#if SC_LINE
        io.put_phase(R + 4 * (k & 1));
        sw.flush();
        if ((k & 1) && k + 1 < SC_M) io.load_line<false>((k + 1) / 2, R);      // both halves are free: the next line, a phase ahead
#else
        io.put_phase(R);
        sw.flush();
        if (k + 1 < SC_M) io.load_phase<false>(k + 1, R);
#endif
        // 1. finish the chunk in progress: its count, and the entry of chunk 2 k
        scan_single<false, LONG>(tab, lt, KW, q, 32u, 0, pc, sw);
        if (k > 0) o.put_cnt<(k > 0 ? 2 * k - 1 : 0)>(pc);
        if (k == 0 && forced) q = 32u + e_forced;
        o.put_ent<2 * k>(q - 32u);
        // 2. chunk 2 k
        u32 c = 0;
        scan_multi<LONG>(tab, lt, KW, q, 288u, c, sw);
        scan_single<false, LONG>(tab, lt, KW, q, 288u, 0, c, sw);
        o.put_cnt<2 * k>(c);
        o.put_ent<2 * k + 1>(q - 288u);
        // 3. chunk 2 k + 1 as far as the row reaches
        pc = 0;
        to_row_end();
        q -= 32u * SC_PHW;
    };
    phase(std::integral_constant<int, 0>{});
    phase(std::integral_constant<int, 1>{});
    phase(std::integral_constant<int, 2>{});
    phase(std::integral_constant<int, 3>{});
    static_assert(SC_M == 4, "four phases written out");
    io.put_next_word(nextw);
    sw.flush();
    scan_single<false, LONG>(tab, lt, KW, q, 32u, 0, pc, sw);
    o.put_cnt<SC_CH - 1>(pc);
    exit_ = (q - 32u) & spec_emask<LONG>();
}

// The walk again, phases rolled (a few waves per launch: registers matter, speed does not): the lanes with `walking` walk from
// entry `e_forced` until an entry equals the one recorded in `o` (back on the first walk's path: the rest of o / exit stands);
// the other lanes only help loading.  Strips inside the stream only (sfd_scan's units; the stream's end is sfd_ends').
template <int LONG>
__device__ __forceinline__ void scan_strip_redo(const lds_u8 *tab, const u16 *lt, const u32 KW, const DecBlk &blk, const u64 wave_off,
                                                const u32 e_forced, bool walking, ScanOut &o, u32 &exit_)
{
    ScanIO io;
    io.init(blk, wave_off);
    ScanWin sw;
    sw.init(io.cb);
    uint4 R[4];
    u32 q = 0, pc = 0;
    // (a walk that repeats loads each phase when it gets there: no registers held across the look-up loops)
#pragma clang loop unroll(disable)
    for (int k = 0; k < SC_M; ++k) {
        io.load_phase<false>(k, R);
        io.put_phase(R);
        sw.flush();
        if (walking) {
            if (k == 0) {
                q = 32u + e_forced;
                o.set_ent(0, e_forced);
            } else {
                scan_single<false, LONG>(tab, lt, KW, q, 32u, 0, pc, sw);
                o.set_cnt(2 * k - 1, pc);
                const u32 e = q - 32u;
                if (e == o.ent(2 * k)) walking = false; else o.set_ent(2 * k, e);
            }
        }
        if (walking) {
            u32 c = 0;
            scan_multi<LONG>(tab, lt, KW, q, 288u, c, sw);
            scan_single<false, LONG>(tab, lt, KW, q, 288u, 0, c, sw);
            o.set_cnt(2 * k, c);
            const u32 e = q - 288u;
            if (e == o.ent(2 * k + 1)) walking = false; else o.set_ent(2 * k + 1, e);
        }
        if (walking) {
            pc = 0;
            scan_multi<LONG>(tab, lt, KW, q, 32u * SC_PHW + 32u, pc, sw);
            if (LONG) scan_single<false, LONG>(tab, lt, KW, q, 32u * SC_PHW, 0, pc, sw);
            q -= 32u * SC_PHW;
        }
        if (!__any(walking)) return;                    // (uniform) every lane is back on its first path
    }
    io.put_next_strip<false>(wave_off);
    sw.flush();
    if (walking) {
        scan_single<false, LONG>(tab, lt, KW, q, 32u, 0, pc, sw);
        o.set_cnt(SC_CH - 1, pc);
        exit_ = (q - 32u) & spec_emask<LONG>();
    }
}

// the wave's unit: tiles wtile, wtile + 1 of the block
template <int LONG>
__device__ __forceinline__ void scan_unit(u8 *smem, const DecBlk &blk, const u32 wtile, u8 *__restrict__ chunk_entry,
                                          u16 *__restrict__ chunk_cnt, u32 *__restrict__ tile_cnt, u8 *__restrict__ tile_guess,
                                          u8 *__restrict__ tile_exit, const u32 tab_bytes)
{
    const u32 lane = threadIdx.x & 63u;
    const size_t gtw = (size_t)blk.tile_base + wtile;
    const u32 KW = blk.KW;
    const lds_u8 *tab = (const lds_u8 *)(size_t)SC_LDS_ROWS;      // a literal: the dynamic segment starts at 0 (checked by the kernel)
    const u16 *lt = (const u16 *)(smem + SC_LDS_ROWS + tab_bytes + SC_MISC);
    const u64 wave_off = (u64)wtile * DTILE;
    const u64 left = blk.in_n > wave_off ? blk.in_n - wave_off : 0;
    if (left < (u64)64 * SC_SB + 4) return;             // (uniform) the stream ends inside the wave's strips: the unit is sfd_ends'
    const bool forced = wtile == 0 && lane == 0;        // the block's first bit: entry 0, no guess
    ScanOut o;
    u32 exit_ = 0;
    scan_strip_fast<LONG>(tab, lt, KW, blk, wave_off, 0u, forced, o, exit_);
    // lanes whose guess differs from the exit in front of them walk again from that exit
    bool bad = false;
    for (int round = 0; round < 3; ++round) {
        const u32 prev = (u32)__shfl_up((int)exit_, 1, 64);
        bad = lane > 0 && (o.e[0] & 0xFFu) != prev;
        if (!__any(bad) || round == 2) break;
        scan_strip_redo<LONG>(tab, lt, KW, blk, wave_off, prev, bad, o, exit_);
    }
    if (__any(bad)) {                                   // did not settle: the block takes the exact kernels
        if (lane == 0) __hip_atomic_store(blk.run_dp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const u32 my_tile = wtile + lane / SC_STRIPS_TILE;
    const u32 incl = dpp_scan_add(o.total());
    const u32 half = (u32)__builtin_amdgcn_readlane((int)incl, 31);
    if (my_tile < blk.n_tiles) {
        const size_t c0 = gtw * DEC_THREADS + (size_t)lane * SC_CH;
        gstore<uint2>(chunk_entry + c0, make_uint2(o.e[0], o.e[1]));
        gstore<uint4>(chunk_cnt + c0, make_uint4(o.c[0], o.c[1], o.c[2], o.c[3]));
    }
    const bool two = wtile + 1 < blk.n_tiles;
    if (lane == 0) tile_guess[gtw] = (u8)(o.e[0] & 0xFFu);
    if (lane == 31) {
        tile_cnt[gtw] = half;
        if (two) { tile_exit[gtw] = 0; tile_guess[gtw + 1] = 0; }      // the link inside the unit was checked above
    }
    if (lane == 63) {
        if (two) tile_cnt[gtw + 1] = incl - half;
        tile_exit[gtw + (two ? 1 : 0)] = (u8)exit_;
    }
}

// dynamic LDS: rows (SC_LDS_ROWS) | cnt3 + len0 (tab_bytes) | flags (SC_MISC) | long-code table (LONG)
template <int LONG>
__global__ __launch_bounds__(DEC_THREADS) __attribute__((amdgpu_waves_per_eu(SC_WAVES, 8))) void sfd_scan(const DecBlk *__restrict__ blks, u8 *__restrict__ chunk_entry,
                                                        u16 *__restrict__ chunk_cnt, u32 *__restrict__ tile_cnt,
                                                        u8 *__restrict__ tile_guess, u8 *__restrict__ tile_exit, u32 tab_bytes, u32 long_bytes)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const DecBlk blk = blks[blockIdx.y];
    if (!blk.run_dp || __hip_atomic_load(blk.run_dp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // exact path
    if (lds_addr(smem) != 0) __builtin_trap();
    const u32 wv = threadIdx.x >> 6;
    const u32 t_lo = blockIdx.x * SC_TILES;
    if (t_lo >= blk.n_tiles) return;
    fill_lds16((void *)(smem + SC_LDS_ROWS), (const void *)blk.cnt3, 2u << blk.KW);
    if (LONG) {
        u16 *lt = (u16 *)(smem + SC_LDS_ROWS + tab_bytes + SC_MISC);
        const u16 *src = LONG == 1 ? blk.longtab : blk.long32;
        if (src) fill_lds16((void *)lt, src, long_bytes);
        else if (threadIdx.x == 0) *lt = 0;
    }
    lds_barrier();
    const u32 wtile = t_lo + wv * SC_WTILES;
    if (wtile < blk.n_tiles) scan_unit<LONG>(smem, blk, wtile, chunk_entry, chunk_cnt, tile_cnt, tile_guess, tile_exit, tab_bytes);
}

// ------------------------------------------------------------------------------------------------
// The unit(s) a block's stream ENDS in, one chunk pair per lane (sl_region, a role of sfd_ends below).
// sfd_scan walks 256-byte strips, one wave per 16 KiB unit: right for a grid that fills the chip, but the unit the stream
// ends in (bounded loads, the cut at the last code) would be ONE wave that walks its 2 K bits per lane alone, and a dependent
// look-up costs a lonely wave ~120 cycles: it used to take 40-50 us at the end of every launch, whatever the launch's size.
// Here that unit is a whole workgroup: 256 lanes with 64 bytes each (two chunks) and the 256 bits in front as run-up — a third
// of the look-ups per chain, all loads in flight at once — verified inside the workgroup like sfd_scan's units (a lane whose
// guess differs from the exit in front of it walks again from that exit); its first lane needs no guess: it runs behind
// sfd_scan and enters at the exit of the unit in front.  (The shape of round 3's sfd_spec, which sfd_scan replaced for the
// bulk of the stream.)
// ------------------------------------------------------------------------------------------------
constexpr int SL_SW = 16;                          // stream words per lane (two chunks)
constexpr int SL_ROW = SL_SW + 1;                  // LDS words per lane: its strip, then a copy of the next strip's first word
constexpr int SL_STRIPS = DEC_THREADS + 1;         // strip 0 = the strip in front of the unit (its last chunk is lane 0's run-up)
constexpr int SL_LDS_DATA = (SL_STRIPS * SL_ROW + 3) / 4 * 16;
constexpr int SL_MISC = DEC_THREADS + 16 + 32;     // exits[256] | wsum[4] | flags[8]

struct SlWin {
    u32 lo, hi, wa;                                     // words at LDS address wa, wa + 4
    __device__ __forceinline__ void init() { wa = 0xFFFFFFFFu; hi = lo = 0; }
    __device__ __forceinline__ u32 at(u32 q)            // the 32 stream bits from LDS bit address q on
    {
        const u32 a = (q >> 3) & ~3u;
        if (a != wa) {
            const lds_u32 *pa = (const lds_u32 *)(size_t)a;
            lo = pa[0];
            hi = pa[1];
            wa = a;
        }
        return __builtin_amdgcn_alignbit(hi, lo, q);
    }
};

// one walk: from LDS bit address q (a code start, by assumption) to the first code start >= qe, counting the codes started on
// the way; the stream ends at bit address `qlimit` (may lie in front of q): a code that does not end inside it is not a symbol
// and nothing starts after it (q = qe + emask: "cut").  tab: sfd_scan's tables (bits | codes << 5 per window, then the length
// of a window's first code).
template <int LONG>
__device__ __forceinline__ void sl_walk(const lds_u8 *tab, const u16 *lt, const u32 KW, u32 &q, const u32 qe, const int qlimit, u32 &cnt, SlWin &sw)
{
    const u32 mask = (1u << KW) - 1u;
    auto long_len = [&](const u32 qq) -> u32 {          // the code of more than KW bits at qq (at least 1: the walk must move)
        const u32 win = __builtin_bitreverse32(sw.at(qq));
        const u32 l = (LONG == 1 ? long_code(lt, win) : long_code32(lt, win)) >> 8;
        return l ? l : 1u;
    };
    // whole fetches while every code taken ends in front of the stream's end (three or two windows of KW bits each)
    const u32 nlook = KW <= 10 ? 3u : 2u;
    while (q + nlook * KW <= qe && (int)(q + nlook * KW) <= qlimit) {
        const u32 w = sw.at(q);
        u32 sum = 0, e = 0;
        for (u32 i = 0; i < nlook; ++i) {
            e = tab[(w >> (sum & 31u)) & mask];
            sum += e;
        }
        q += sum & 31u;
        cnt += sum >> 5;
        if (LONG && __builtin_expect((e & 31u) == 0u, 0)) {      // a long code stopped the look-ups: it starts at q
            const u32 l = long_len(q);
            if ((int)(q + l) > qlimit) { q = qe + spec_emask<LONG>(); return; }
            q += l;
            ++cnt;
        }
    }
    const lds_u8 *len0 = tab + (1u << KW);
    while (q < qe) {
        u32 l0 = len0[sw.at(q) & mask];
        if (LONG && __builtin_expect(l0 == 0u, 0)) l0 = long_len(q);
        if ((int)(q + l0) > qlimit) { q = qe + spec_emask<LONG>(); break; }   // cut by the end of the stream
        q += l0;
        ++cnt;
    }
}

// a lane's two chunks (LDS bit address qrow) from entry `ent0` of the first: entries, counts, the exit of the second.
// HAVE_OLD: ent[] holds the entries of an earlier walk: once this walk meets it the rest is unchanged.
template <bool HAVE_OLD, int LONG>
__device__ __forceinline__ void sl_strip(const lds_u8 *tab, const u16 *lt, const u32 KW, const u32 qrow, const int qlimit, const u32 ent0,
                                         u32 (&ent)[2], u32 (&cnt)[2], u32 &exit_)
{
    u32 q = qrow + ent0;
    SlWin sw;
    sw.init();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const u32 r = q - (qrow + 256u * k);
        if (HAVE_OLD && k > 0 && r == ent[k]) return;   // back on the earlier walk's path
        ent[k] = r;
        u32 c = 0;                                      // (a chunk the stream ended in front of: entry emask, cut at once, count 0)
        sl_walk<LONG>(tab, lt, KW, q, qrow + 256u * (k + 1), qlimit, c, sw);
        cnt[k] = c;
    }
    exit_ = (q - (qrow + 512u)) & spec_emask<LONG>();
}

// One unit the stream ends in (tiles tile0, tile0 + 1), the tables already in LDS.  `have_e0`: lane 0 enters at e0 (the exit of the
// unit in front, which is final by now: sfd_ends runs behind sfd_scan), else at the block's first bit.  Returns the unit's exit.
// dynamic LDS: stream frame (SL_LDS_DATA) | cnt3 + len0 (tab_bytes) | exits[256] wsum[4] flags[8] (SL_MISC) | long-code table (LONG)
template <int LONG>
__device__ __forceinline__ u32 sl_region(u8 *smem, const DecBlk &blk, const u32 tile0, const u32 e0_forced, u8 *__restrict__ chunk_entry,
                                         u16 *__restrict__ chunk_cnt, u32 *__restrict__ tile_cnt, u8 *__restrict__ tile_guess,
                                         u8 *__restrict__ tile_exit, const u32 tab_bytes)
{
    const u64 start = (u64)tile0 * DTILE;
    const u64 left = blk.in_n > start ? blk.in_n - start : 0;
    const size_t gt0 = (size_t)blk.tile_base + tile0;
    const u32 ntl = blk.n_tiles - tile0 < (u32)SC_WTILES ? blk.n_tiles - tile0 : (u32)SC_WTILES;   // tiles of the unit
    u32 *data = (u32 *)smem;
    const u32 tab_off = SL_LDS_DATA;
    u8 *ex = smem + SL_LDS_DATA + tab_bytes;
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 KW = blk.KW;
    const u16 *lt = (const u16 *)(smem + SL_LDS_DATA + tab_bytes + SL_MISC);
    {   // the unit's stream from one strip before it, 16 bytes a lane; frame word f -> LDS word f + f / SL_SW.  All of a lane's
        // pieces are requested before the first is used; bytes outside the stream read as zero.
        const long long base = (long long)start - 4 * SL_SW;
        constexpr u32 UNITS = (u32)(SL_STRIPS * SL_SW / 4 + 1), NIT = (UNITS + DEC_THREADS - 1) / DEC_THREADS;
        uint4 v[NIT];
#pragma unroll
        for (u32 it = 0; it < NIT; ++it) {
            const u32 i = tid + it * DEC_THREADS;
            const long long off = base + (long long)i * 16;
            v[it] = make_uint4(0, 0, 0, 0);
            if (i < UNITS && off >= 0 && (u64)off + 16 <= blk.in_n) v[it] = gload<uint4>(blk.in + off);
        }
#pragma unroll
        for (u32 it = 0; it < NIT; ++it) {
            const u32 i = tid + it * DEC_THREADS;
            if (i >= UNITS) break;
            const long long off = base + (long long)i * 16;
            u32 w[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
            if (off >= 0 && (u64)off < blk.in_n && (u64)off + 16 > blk.in_n) {      // the piece the stream ends in
                const int nv = (int)(blk.in_n - (u64)off);
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (q < nv) w[q >> 2] |= (u32)gload<u8>(blk.in + off + q) << (8 * (q & 3));
            }
            const u32 f = 4 * i, at = f + f / SL_SW;
            const u32 w0 = rev_bytes(w[0]);
            if (f % SL_SW == 0 && f > 0) data[at - 1] = w0;         // the previous row's look-ahead word
            if (i < UNITS - 1) {
                data[at] = w0;
                data[at + 1] = rev_bytes(w[1]);
                data[at + 2] = rev_bytes(w[2]);
                data[at + 3] = rev_bytes(w[3]);
            }
        }
    }
    __syncthreads();
    // lane tid owns frame strip tid + 1; the stream ends at bit `limit` of that strip (<= 0: in front of it)
    const int limit = (int)(left * 8) - (int)(512u * tid);
    const u32 qrow = 8u * ((4u * SL_ROW) * (tid + 1));  // LDS bit address of the strip's first bit (the segment starts at 0)
    const int qlimit = (int)qrow + limit;
    const lds_u8 *tab = (const lds_u8 *)(size_t)tab_off;
    u32 ent[2] = {0, 0}, cnt[2] = {0, 0}, exit_ = 0;
    {
        u32 e0;
        if (tid == 0) e0 = e0_forced;                   // known: the block's first bit, or the exit of the unit in front
        else {                                          // run-up: the last chunk of the strip in front, from its first bit
            const u32 qe = qrow;
            u32 q = qe - 256u, dummy = 0;
            SlWin sw;
            sw.init();
            sl_walk<LONG>(tab, lt, KW, q, qe, qlimit, dummy, sw);
            e0 = (q - qe) & spec_emask<LONG>();
        }
        sl_strip<false, LONG>(tab, lt, KW, qrow, qlimit, e0, ent, cnt, exit_);
    }
    ex[tid] = (u8)exit_;
    __syncthreads();
    // lanes whose guess differs from the exit in front of them walk again from that exit, until the walk meets the old one
    bool bad = false;
    u32 *const wsum = (u32 *)(ex + DEC_THREADS), *const flags = wsum + 4;
    u32 turn = 0;
    for (int round = 0; round < 4; ++round) {
        bad = tid > 0 && ent[0] != (u32)ex[tid - 1];
        if (!wg_any(bad, flags, turn)) break;
        if (bad) sl_strip<true, LONG>(tab, lt, KW, qrow, qlimit, (u32)ex[tid - 1], ent, cnt, exit_);
        __syncthreads();                                // every lane has read the exit in front of it
        ex[tid] = (u8)exit_;
        __syncthreads();
        bad = tid > 0 && ent[0] != (u32)ex[tid - 1];
    }
    if (wg_any(bad, flags, turn)) {                     // did not settle: the block takes the exact kernels
        if (tid == 0) __hip_atomic_store(blk.run_dp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // chunks 2 tid, 2 tid + 1 of the unit; a tile is 128 consecutive lanes (two waves)
    const u32 my_tile = tid / 128u;
    const u32 wtot = dpp_scan_add(cnt[0] + cnt[1]);     // lane 63: the wave's codes
    if (lane == 63) wsum[wv] = wtot;
    __syncthreads();
    if (my_tile < ntl) {
        const size_t c0 = gt0 * DEC_THREADS + (size_t)tid * 2;
        gstore<u16>(chunk_entry + c0, (u16)((ent[0] & 0xFFu) | (ent[1] << 8)));
        gstore<u32>(chunk_cnt + c0, cnt[0] | (cnt[1] << 16));
        const u32 in_tile = tid & 127u;
        if (in_tile == 0) {
            tile_cnt[gt0 + my_tile] = wsum[2 * my_tile] + wsum[2 * my_tile + 1];
            tile_guess[gt0 + my_tile] = (u8)ent[0];
        }
        if (in_tile == 127u) tile_exit[gt0 + my_tile] = (u8)exit_;
    }
    const u32 unit_exit = ex[ntl * 128u - 1u];          // the exit of the unit's last existing tile
    __syncthreads();                                    // (the frame and ex[] may be rewritten by the next unit)
    return unit_exit;
}

// ------------------------------------------------------------------------------------------------
// sfd_fix: the links BETWEEN units, checked and repaired in one launch (one workgroup per block).
// A unit's first lane guesses its entry from a 256-bit run-up; the unit in front of it knows the truth: its exit.  Where the
// two differ (0.2 % of the links of Zipf-like data: a handful per 64 MiB block) the unit's walk was wrong only until it met
// the true parse — a few dozen bits.  So ONE lane per bad link walks again from the exit in front, code by code, rewriting the
// entries and counts of the unit's first chunks until its position at a chunk boundary equals the entry recorded there: from
// then on everything the unit wrote stands, its own exit included — which is why one round is enough and no link depends on
// another's repair.  (This used to be two rounds of a check kernel and a repair launch of sfd_scan, where a whole 16 KiB unit was walked
// again by one lonely wave: 40-50 us per round, whatever the launch's size.)  A link that does not heal within SF_CHUNKS
// chunks, or lies within that distance of the stream's end, sends its block to the exact kernels; when every link of a
// block holds, its entries ARE the true parse (the argument at the top of this file).
// ------------------------------------------------------------------------------------------------
constexpr u32 SF_CHUNKS = 8;                       // chunks a repair may walk before it gives up (2048 bits)

// the links into tiles t_begin .. t_end - 1, at most SF_LINKS of them (the tiles of the units the stream ends in are the ends role's).
// Checked first — two loads a link, all in flight at once; a slice whose links all hold (nearly every slice) is done then —
// and only a slice with a link to repair brings the tables into LDS (at smem).
constexpr u32 SF_LINKS = 2 * DEC_THREADS;          // links per workgroup: a 64 MiB block is 16 workgroups
template <int LONG>
__device__ __forceinline__ void sf_links(u8 *smem, const DecBlk &blk, const u32 t_begin, const u32 t_end, u8 *__restrict__ chunk_entry,
                                         u16 *__restrict__ chunk_cnt, u32 *__restrict__ tile_cnt, u8 *__restrict__ tile_guess,
                                         const u8 *__restrict__ tile_exit, const u32 tab_bytes, const u32 long_bytes)
{
    const u32 tid = threadIdx.x, KW = blk.KW, mask = (1u << KW) - 1u;
    const u16 *lt = (const u16 *)(smem + tab_bytes + 48);      // tables | wg_any's flags (32 bytes) | gave_up | long-code table
    const u8 *len0 = smem + (1u << KW);
    const u64 end_bit = blk.in_n * 8;
    u32 turn = 0;
    u32 xs[SF_LINKS / DEC_THREADS];
    bool any_bad = false;
#pragma unroll
    for (u32 i = 0; i < SF_LINKS / DEC_THREADS; ++i) {
        const u32 t = t_begin + tid + i * DEC_THREADS;
        xs[i] = 0xFFFFFFFFu;                            // holds / not this slice's
        if (t < t_end) {
            const size_t gt = (size_t)blk.tile_base + t;
            const u32 x = tile_exit[gt - 1];
            if (tile_guess[gt] != x) { xs[i] = x; any_bad = true; }
        }
    }
    if (!wg_any(any_bad, (u32 *)(smem + tab_bytes), turn)) return;
    __syncthreads();                                    // (the flag words lie behind the tables: read by everybody before the fill)
    fill_lds16((void *)smem, (const void *)blk.cnt3, 2u << KW);
    if (LONG) {
        const u16 *src = LONG == 1 ? blk.longtab : blk.long32;
        if (src) fill_lds16((void *)lt, src, long_bytes);
        else if (tid == 0) *(u16 *)lt = 0;
    }
    bool failed = false;
    // (one link that does not heal sends the whole block to the exact kernels: the others stop looking — a table that does not
    // re-synchronise, forced to speculate, would otherwise walk eight chunks behind most of its links)
    volatile u32 *const gave_up = (volatile u32 *)(smem + tab_bytes + 32);
    if (tid == 0) *gave_up = 0;
    __syncthreads();
#pragma unroll 1
    for (u32 i = 0; i < SF_LINKS / DEC_THREADS; ++i) {
        if (*gave_up) break;
        const u32 x = xs[i];
        if (x == 0xFFFFFFFFu) continue;
        const u32 t = t_begin + tid + i * DEC_THREADS;
        const size_t gt = (size_t)blk.tile_base + t;
        const u64 tile_bit0 = (u64)t * DTILE * 8;
        // a "stream ended" mark in front, or the stream's end within reach of the walk: not repaired here
        if (x >= spec_emask<LONG>() || tile_bit0 + (u64)(SF_CHUNKS + 1) * CH_BITS + 64 > end_bit) { failed = true; *gave_up = 1; continue; }
        u32 q = x, wi = 0xFFFFFFFFu, lo = 0, hi = 0;    // q: bit position in the tile; (lo, hi): stream words wi, wi + 1 of the tile, LSB first
        auto window = [&](const u32 qq) -> u32 {
            const u32 i = qq >> 5;
            if (i != wi) {
                const u8 *p = blk.in + (u64)t * DTILE + 4u * i;
                lo = rev_bytes(gload<u32>(p));
                hi = rev_bytes(gload<u32>(p + 4));
                wi = i;
            }
            return __builtin_amdgcn_alignbit(hi, lo, qq);
        };
        int delta = 0;
        bool merged = false;
        const size_t c0 = gt * DEC_THREADS;
        for (u32 c = 0; c < SF_CHUNKS; ++c) {
            const u32 qe = CH_BITS * (c + 1);
            u32 cnt = 0;
            while (q < qe) {
                const u32 w = window(q);
                u32 l = len0[w & mask];
                if (LONG && l == 0u) {
                    const u32 win = __builtin_bitreverse32(w);
                    l = (LONG == 1 ? long_code(lt, win) : long_code32(lt, win)) >> 8;
                    if (!l) l = 1u;
                }
                q += l;
                ++cnt;
            }
            delta += (int)cnt - (int)chunk_cnt[c0 + c];
            chunk_cnt[c0 + c] = (u16)cnt;
            const u32 e = q - qe;
            if (e == (u32)chunk_entry[c0 + c + 1]) { merged = true; break; }
            chunk_entry[c0 + c + 1] = (u8)e;
        }
        if (!merged) { failed = true; *gave_up = 1; continue; }
        chunk_entry[c0] = (u8)x;
        tile_guess[gt] = (u8)x;
        tile_cnt[gt] = (u32)((int)tile_cnt[gt] + delta);
    }
    if (wg_any(failed, (u32 *)(smem + tab_bytes), turn) && tid == 0)
        __hip_atomic_store(blk.run_dp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sfd_ends, behind sfd_scan: grid (1 + slices of SF_LINKS links, blocks).  Workgroup 0 of a block: the unit(s) its stream ends in
// (sl_region; sfd_scan leaves them out), the first lane entered at the exit of the unit in front — final by now, so these units'
// links hold by construction.  Workgroups 1 ..: the other links of the block, a slice each (sf_links; no link depends on
// another's repair).  Any of them sets *run_dp when something does not hold.
template <int LONG>
__global__ __launch_bounds__(DEC_THREADS) void sfd_ends(const DecBlk *__restrict__ blks, u8 *__restrict__ chunk_entry,
                                                        u16 *__restrict__ chunk_cnt, u32 *__restrict__ tile_cnt,
                                                        u8 *__restrict__ tile_guess, u8 *__restrict__ tile_exit, u32 tab_bytes, u32 long_bytes)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const DecBlk blk = blks[blockIdx.y];
    if (!blk.n_tiles || !blk.run_dp || __hip_atomic_load(blk.run_dp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // exact path
    if (lds_addr(smem) != 0) __builtin_trap();
    // the units sfd_scan left out: those with fewer than 64 strips + 4 bytes of stream from their start (the last one; the one
    // before it too when the stream ends one to three bytes into the last)
    const u32 units = (blk.n_tiles + SC_WTILES - 1) / SC_WTILES;
    u32 first_end = units - 1;
    if (units >= 2 && blk.in_n - (u64)(units - 2) * SC_WTILES * DTILE < (u64)64 * SC_SB + 4) first_end = units - 2;
    if (blockIdx.x > 0) {                               // a slice of the block's links
        const u32 t_end = first_end * SC_WTILES, t_begin = 1u + (blockIdx.x - 1u) * SF_LINKS;
        if (t_begin < t_end)
            sf_links<LONG>(smem, blk, t_begin, t_end, chunk_entry, chunk_cnt, tile_cnt, tile_guess, tile_exit, tab_bytes, long_bytes);
        return;
    }
    fill_lds16((void *)(smem + SL_LDS_DATA), (const void *)blk.cnt3, 2u << blk.KW);
    if (LONG) {
        u16 *lt = (u16 *)(smem + SL_LDS_DATA + tab_bytes + SL_MISC);
        const u16 *src = LONG == 1 ? blk.longtab : blk.long32;
        if (src) fill_lds16((void *)lt, src, long_bytes);
        else if (threadIdx.x == 0) *lt = 0;
    }
    __syncthreads();
    u32 e0 = first_end ? (u32)tile_exit[(size_t)blk.tile_base + first_end * SC_WTILES - 1] : 0u;
    for (u32 u = first_end; u < units; ++u)
        e0 = sl_region<LONG>(smem, blk, u * SC_WTILES, e0, chunk_entry, chunk_cnt, tile_cnt, tile_guess, tile_exit, tab_bytes);
}

// ------------------------------------------------------------------------------------------------
// sfd_wstage: the symbol pass for complete codes (three codes per look-up), staged through LDS.
// A lane decodes one chunk from its entry, exactly the symbols counted for it.  The workgroup's symbols are one
// contiguous run of the output, so they are collected in an LDS image of that run (aligned like the output address)
// and leave as aligned 16-byte stores, fully coalesced.  A lane gathers symbols in a 32-bit word and ORs each finished
// word into the image (ds_or_b32: the words at both ends of a lane's run are shared with its neighbours; only lanes
// with a finished word issue it — an LDS atomic costs bank cycles per active lane, and two steps in three finish none).
// The stream sits in LDS one row per chunk (8 words + the next row's first: rows 9 words apart, conflict free), LSB
// first (bytes bit-reversed): the window at LDS bit address q is alignbit(W[(q>>5)+1], W[q>>5], q), no bit buffer,
// and the 30 bits it delivers serve two or three look-ups.
// dynamic LDS: rows | sym3 (tab_bytes) | long table (LONG) | image (cap bytes) | wsum[4], next
// The host sizes the image for the launch's average symbols per tile plus a margin (LDS is what limits the waves per
// CU); a tile with more symbols than it holds goes in several rounds of consecutive lanes.
// ------------------------------------------------------------------------------------------------
// LDS words per chunk row: the chunk and the words behind it that the last fetch of the walk may read — two, or three when a
// code longer than the window (ESC: read as 32 bits) may start in them.  Rows 10 words apart would put lanes 16 apart on one
// bank: every 16 rows they move on by a word (row r at word 10 r + r / 16), so that the 32 lanes of a half wave that are at
// the same word of their rows are on 32 banks; 11 is odd as it is.
constexpr int ws_row(bool esc) { return CH_BYTES / 4 + (esc ? 3 : 2); }
constexpr int ws_rows_bytes(bool esc)               // 16 in front: the window at a row's bit 0 reads the word before it
{
    return 16 + (DEC_THREADS * ws_row(esc) + (esc ? 0 : DEC_THREADS / 16)) * 4;
}
constexpr int WS_MISC = 32;

// ESC: some code of the launch may be longer than its block's sym3 window (then a look-up can return no symbol)
template <int LONG, bool ESC>
__global__ __launch_bounds__(DEC_THREADS) void sfd_wstage(const DecBlk *__restrict__ blks, const u8 *__restrict__ chunk_entry,
                                                          const u16 *__restrict__ chunk_cnt, const u64 *__restrict__ tile_off,
                                                          u32 tpw, u32 tab_bytes, u32 cap, u32 long_bytes)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const DecBlk blk = blks[blockIdx.y];
    const u32 first_tile = blockIdx.x * tpw;
    if (first_tile >= blk.n_tiles) return;
    if (tile_off[(size_t)blk.tile_base + first_tile] >= blk.n_sym) return;   // all padding / past the end
    // LONG == 1: the table of the 13..16-bit codes is kept up to the launch's largest group count only (`long_bytes` = header
    // + prefixes + that many groups: a few hundred bytes for a real file's rare symbols, 4.3 KB in full — the difference
    // decides whether four workgroups share a CU or two, LABNOTES.md §3.2)
    const u32 LONGB = LONG ? long_bytes : 0u;           // (LONG == 2: header, prefixes, roots and the sub-trie nodes in use)
    u32 *rows = (u32 *)(smem + 16);
    constexpr u32 WS_ROW = (u32)ws_row(ESC), tab_off = (u32)ws_rows_bytes(ESC);
    const u16 *lt = (const u16 *)(smem + tab_off + tab_bytes);
    const u32 img_off = tab_off + tab_bytes + LONGB;
    u32 *wsum = (u32 *)(smem + img_off + cap);
    u32 *next = wsum + 4;
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 K1 = blk.K1, K3 = sym3_window(K1);
    const u32 sh = 32 - K1;
    fill_lds16(smem + tab_off, (const void *)blk.sym3, 4u << K3);
    if (LONG) {
        const u16 *src = LONG == 1 ? blk.longtab : blk.long32;
        if (src) fill_lds16((void *)lt, src, LONGB);
        else if (tid == 0) *(u16 *)lt = 0;
    }
    for (u32 i = tid; i < cap / 16; i += DEC_THREADS) *(uint4 *)(smem + img_off + 16 * i) = make_uint4(0, 0, 0, 0);
    bool bad = false;
    // LDS addresses as literals: the kernel has no static LDS, so its dynamic segment starts at 0 (checked), and a
    // literal goes into the ds offset field where "smem + offset" leaves an add per look-up
    if (lds_addr(smem) != 0) __builtin_trap();
    const u32 q2row = 8u * (16u + 4u * (WS_ROW * tid + (ESC ? 0u : tid >> 4))) - 2u;     // LDS bit address of the lane's row, minus 2 (see step)
    const u32 mask4 = ((1u << K3) - 1u) << 2;
    // 16 stream bytes at `off` (zeros past the end)
    auto fetch16 = [&](const u64 off) -> uint4 {
        if (off + 16 <= blk.in_n) return gload_nt<uint4>(blk.in + off);
        u32 w[4] = {0, 0, 0, 0};
        if (off < blk.in_n) {
            const int nv = (int)(blk.in_n - off);
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < nv) w[q >> 2] |= (u32)gload<u8>(blk.in + off + q) << (8 * (q & 3));
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    };
    // a tile's inputs travel through registers: they are requested while the tile before is being decoded
    uint4 pf0, pf1, pf2 = make_uint4(0, 0, 0, 0);
    u32 pf_entry = 0, pf_cnt = 0, pf_next = 0;
    auto prefetch = [&](const u32 tile) {
        const u64 base = (u64)tile * DTILE;
        pf0 = fetch16(base + 16ull * tid);
        pf1 = fetch16(base + 16ull * (tid + DEC_THREADS));
        if (tid == 0) pf2 = fetch16(base + DTILE);      // the last row's look-ahead word
        const size_t g = ((size_t)blk.tile_base + tile) * DEC_THREADS + tid;
        pf_entry = chunk_entry[g];
        pf_cnt = chunk_cnt[g];
        // the entry of the chunk behind this one = where this chunk's codes end (not asked for in a block's last tile)
        pf_next = (tile + 1 < blk.n_tiles || tid + 1 < (u32)DEC_THREADS) ? chunk_entry[g + 1] : 0u;
    };
    // tile word f -> word f % 8 of row f / 8; a row's first two (ESC: three) words are also the look-ahead words of the row before
    auto put16 = [&](const u32 i, const uint4 v) {
        const u32 f = 4 * i, at = f + (WS_ROW - 8u) * (f >> 3) + (ESC ? 0u : f >> 7);
        const u32 w0 = rev_bytes(v.x), w1 = rev_bytes(v.y), w2 = rev_bytes(v.z);
        if ((f & 7u) == 0 && f > 0) {                  // (not ESC: the row in front ends a word earlier when this row is the first of its 16)
            const u32 lk = at - (WS_ROW - 8u) - (!ESC && (f & 127u) == 0 ? 1u : 0u);
            rows[lk] = w0;
            rows[lk + 1] = w1;
            if (ESC) rows[lk + 2] = w2;
        }
        if (i < (u32)(DTILE / 16)) {
            rows[at] = w0;
            rows[at + 1] = w1;
            rows[at + 2] = w2;
            rows[at + 3] = rev_bytes(v.w);
        }
    };
    // bytes [lo, hi) of a 16-byte piece, one by one (the ends of a workgroup's run of the output)
    auto store_bytes = [&](u8 *ga, const uint4 v, const u32 lo, const u32 hi) {
        const u32 wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (u32 q = 0; q < 16; ++q)
            if (q >= lo && q < hi) gstore<u8>(ga + q, (u8)(wds[q >> 2] >> (8 * (q & 3))));
    };
    u32 carry_n = 0, carry_lo = 0;                      // (uniform) piece 0 of the image holds bytes [carry_lo, carry_n) of the
    u8 *carry_g = nullptr;                              //   output piece at carry_g, kept from the round before
    prefetch(first_tile);
    for (u32 it = 0; it < tpw; ++it) {
        const u32 tile = first_tile + it;
        if (tile >= blk.n_tiles) break;
        const size_t gt = (size_t)blk.tile_base + tile;
        const u64 toff = tile_off[gt];
        if (toff >= blk.n_sym) break;                   // the rest is padding / past the end
        lds_barrier();                                // the previous tile's rows and image are done with
        put16(tid, pf0);
        put16(tid + DEC_THREADS, pf1);
        if (tid == 0) put16(DTILE / 16, pf2);
        const u32 entry = pf_entry, cnt = pf_cnt, nent = pf_next;
        if (it + 1 < tpw && tile + 1 < blk.n_tiles) prefetch(tile + 1);
        const u32 incl = wave_incl_scan_add<u32>(cnt);
        if (lane == 63) wsum[wv] = incl;
        lds_barrier();
        u32 pre = incl - cnt, total = 0;
        for (u32 w = 0; w < 4; ++w) { if (w < wv) pre += wsum[w]; total += wsum[w]; }
        const u64 room = blk.n_sym - toff;              // symbols of this tile that exist in the block
        const u32 tot_c = room < (u64)total ? (u32)room : total;
        u32 want = pre >= tot_c ? 0u : (tot_c - pre < cnt ? tot_c - pre : cnt);
        u32 q2 = q2row + entry;
        // A lane that emits all of its chunk's codes, in a tile behind which the stream goes on, knows where they end: the
        // next chunk's entry.  Its main loop then runs on the POSITION (whole fetches while every code taken starts in front
        // of that end) instead of counting symbols down: no count bookkeeping per fetch, and a shorter tail of single steps
        // (on average 15 bits are left instead of 4.5 symbols).
        // (not when the next entry is the "stream ended in front of this chunk" mark — a truncated stream: the walk would
        // run on zero fill past the lane's count; the counted loop is exact there)
        const bool by_pos = want == cnt && tile + 1 < blk.n_tiles && nent != spec_emask<LONG>();
        const u32 q2end = q2row + 256u + nent;
        for (u32 done = 0; done < tot_c;) {
            u8 *gout = blk.out + toff + done;
            const u32 mis = (u32)((uintptr_t)gout & 15u);
            if (carry_g && (carry_n != mis || carry_g != gout - mis)) {      // (uniform) the kept pieces are not where this round
                if (tid == 0) {                                              //   starts (cannot happen while tile offsets are a
                    uint4 *ip = (uint4 *)(smem + img_off);                   //   scan of the counts): they leave first
                    store_bytes(carry_g, *ip, carry_lo, carry_n);
                    ip[0] = make_uint4(0, 0, 0, 0);
                    ip[1] = make_uint4(0, 0, 0, 0);
                }
                carry_n = 0;
                carry_g = nullptr;
                lds_barrier();
            }
            const u32 capw = cap - 32;
            u32 nxt = tot_c;
            if (mis + (tot_c - done) > capw) {          // (uniform) more than the image holds: consecutive lanes that fit
                if (tid == 0) *next = tot_c;
                lds_barrier();
                if (want && pre >= done && mis + (pre - done) + want > capw) atomicMin(next, pre);
                lds_barrier();
                nxt = *next;
                lds_barrier();
            }
            if (want && pre >= done && pre + want <= nxt) {
                const u32 x = mis + (pre - done);
                // LDS addresses are absolute from here on (the dynamic segment's base folded into the constants): an
                // address that is "base + variable" costs an add per look-up that the ds instructions cannot absorb
                // the bytes already in the word being gathered are counted in the TOP two bits of nbx: adding a look-up's
                // count (the top two bits of its entry) carries out exactly when the word is finished
                u32 wp = img_off + (x & ~3u), nbx = x << 30, acc = 0;
                // the symbols of entry e (its low bytes; count in bits 30..31) join the word being gathered; a finished word
                // is ORed into the image
                auto emit = [&](const u32 e) {
                    const u32 sh = nbx >> 27, syms = e & 0xFFFFFFu;
                    acc |= syms << sh;
                    if (__builtin_add_overflow(nbx, e & 0xC0000000u, &nbx)) {   // only the lanes with a finished word touch the image
                        __hip_atomic_fetch_or((lds_u32 *)(size_t)wp, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        wp += 4u;
                        acc = syms >> (32u - sh);       // the bytes that did not fit (sh >= 8 here: at most three symbols)
                    }
                };
                // the rows hold the stream LSB first; the 64 bits around the position shifted down to two bits in
                // front of it are the window times four, the byte offset of its table entry: alignbit, and, read
                auto window4 = [&]() -> u32 {
                    const lds_u32 *pa = (const lds_u32 *)(size_t)((q2 >> 3) & ~3u);
                    return __builtin_amdgcn_alignbit(pa[1], pa[0], q2);
                };
                auto step = [&](const bool tail) {
                    u32 e = *(const lds_u32 *)(size_t)(tab_off + (window4() & mask4));
                    if (ESC && __builtin_expect((e >> 30) == 0, 0)) {      // first code longer than the window: one code
                        const u32 q = q2 + 2u;                             // its 32 bits, first bit at the top
                        const lds_u32 *pb = (const lds_u32 *)(size_t)((q >> 3) & ~3u);
                        const u32 win = __builtin_bitreverse32(__builtin_amdgcn_alignbit(pb[1], pb[0], q));
                        u32 e1 = LONG == 1 ? long_code(lt, win) : LONG == 2 ? long_code32(lt, win)
                                           : (u32)gload<u16>(blk.lut13 + (win >> sh));
                        if (e1 == 0) { bad = true; e1 = 1u << 8; }  // not a code (complete tables never get here)
                        e = (e1 & 0xFFu) | ((e1 >> 8) << 24) | (1u << 30);
                    }
                    u32 n = e >> 30, syms = e & 0xFFFFFFu;
                    if (tail) {
                        n = n < want ? n : want;
                        syms &= (1u << (8 * n)) - 1u;
                    }
                    emit(syms | n << 30);
                    q2 += (e >> 24) & 63u;
                    want -= n;
                };
                // N look-ups per stream read: the 30 window bits fetched hold further windows behind the first one's
                // codes as long as N windows fit (a look-up uses at most K3 bits).  ESC: a window that starts with a code
                // longer than K3 bits has the entry 0 — it emits nothing and uses no bits, so the look-ups behind it in
                // the same round see the same window and do nothing either; the long code is then taken by one step
                auto multi = [&](auto nlook) {
                    constexpr u32 N = decltype(nlook)::value;
                    // (the top bytes of a fetch's entries — bits used, symbols << 6 — are summed: at most 30 bits, so the
                    // sum's low five bits are the next look-up's shift as they stand)
                    if (by_pos) {
                        // Whole fetches while the fetch STARTS in front of the end: the last one runs into the next chunk by up
                        // to 29 bits and nine symbols — the very symbols the next lane (round, tile) places at the very bytes
                        // this lane's word pointer has reached, and an OR of equal bytes changes nothing.  No tail of clipped
                        // single steps, no count.  (A look-up reads its window from at most 19 bits behind the end, a long
                        // code 32 bits from there: inside the row's two — ESC: three — look-ahead words.  The bytes this leaves behind the
                        // round's end stay in the image with the piece the round ends in, below.)
                        while ((int)q2 < (int)q2end) {
                            const u32 w4 = window4();
                            u32 ua = 0, e = 0;
#pragma unroll
                            for (u32 i = 0; i < N; ++i) {
                                e = *(const lds_u32 *)(size_t)(tab_off + ((w4 >> (ua & 31u)) & mask4));
                                emit(e);
                                ua += e >> 24;
                            }
                            q2 += ua & 63u;
                            if (ESC && __builtin_expect((e >> 30) == 0, 0)) step(false);     // a long code stopped the look-ups: it starts at q2
                        }
                        want = 0;
                    }
                    while (want >= 3 * N) {
                        const u32 w4 = window4();
                        u32 ua = 0, e = 0;
#pragma unroll
                        for (u32 i = 0; i < N; ++i) {
                            e = *(const lds_u32 *)(size_t)(tab_off + ((w4 >> (ua & 31u)) & mask4));
                            emit(e);
                            ua += e >> 24;
                        }
                        q2 += ua & 63u;
                        want -= ua >> 6;
                        if (ESC && __builtin_expect((e >> 30) == 0, 0)) step(false);
                    }
                };
                if (K3 <= 10) multi(std::integral_constant<u32, 3>{});
                else if (K3 <= 15) multi(std::integral_constant<u32, 2>{});
                while (want) step(true);
                if (nbx) __hip_atomic_fetch_or((lds_u32 *)(size_t)wp, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            lds_barrier();
            // Image bytes [mis, end) are this round's symbols.  Whole 16-byte pieces leave as aligned stores; the piece the round
            // ends in STAYS: it becomes piece 0 of the workgroup's next round or tile, whose output continues exactly there
            // (tile offsets are a scan of the counts), so inside a workgroup's run of tiles no piece is ever partial.  Only the
            // run's first piece (bytes in front of it belong to another workgroup) and its last go out byte by byte.
            const u32 end = mis + (nxt - done);
            const u32 lo0 = (carry_n == mis && carry_n) ? carry_lo : mis;     // piece 0: bytes [lo0, 16) are this workgroup's
            const u32 nfull = end >> 4;
            u8 *const g0 = gout - mis;
            for (u32 u = tid; u < nfull; u += DEC_THREADS) {
                uint4 *ip = (uint4 *)(smem + img_off + 16 * u);
                const uint4 v = *ip;
                *ip = make_uint4(0, 0, 0, 0);
                if (u == 0 && lo0) store_bytes(g0, v, lo0, 16u);
                else gstore_nt<uint4>(g0 + 16 * u, v);
            }
            if (tid == 0 && nfull) {                    // the piece the round ends in and the one behind it (symbols of the next
                uint4 *ip = (uint4 *)(smem + img_off);  //   chunk that the round's last lane has already placed there) move to
                const uint4 a = ip[nfull], b = ip[nfull + 1];           //   the front of the image
                ip[nfull] = make_uint4(0, 0, 0, 0);
                ip[nfull + 1] = make_uint4(0, 0, 0, 0);
                ip[0] = a;
                ip[1] = b;
            }
            carry_lo = nfull ? 0u : lo0;
            carry_n = end & 15u;
            carry_g = g0 + 16 * nfull;
            done = nxt;
            if (done < tot_c) lds_barrier();          // the image is in place again before the next round's ORs
        }
    }
    if (tid == 0 && carry_n) {                          // the last piece of the workgroup's run of tiles
        const uint4 v = *(const uint4 *)(smem + img_off);
        store_bytes(carry_g, v, carry_lo, carry_n);
    }
    if (bad) set_error(blk.err, SHAFA_FILE_UNRECOGNIZABLE);
}

// ------------------------------------------------------------------------------------------------
// sfd_offsets: per block, exclusive scan of the tile counts; stream too short => error
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DEC_THREADS) void sfd_offsets(const DecBlk *__restrict__ blks,
                                                           const u32 *__restrict__ tile_cnt, u64 *__restrict__ tile_off)
{
    __shared__ u64 wtot[4];
    __shared__ u64 carry;
    const DecBlk blk = blks[blockIdx.x];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    // a thread takes OFF_EPT consecutive tiles a round, all of its loads issued before the first is used (one tile per
    // thread and round: 32 rounds of a load and three barriers for a 64 MiB block, 28 us)
    constexpr u32 OFF_EPT = 16;
    for (u32 r0 = 0; r0 < blk.n_tiles; r0 += DEC_THREADS * OFF_EPT) {
        const u32 t0 = r0 + tid * OFF_EPT;
        u32 x[OFF_EPT];
#pragma unroll
        for (u32 j = 0; j < OFF_EPT; ++j) x[j] = t0 + j < blk.n_tiles ? tile_cnt[blk.tile_base + t0 + j] : 0u;
        u64 mine = 0;
#pragma unroll
        for (u32 j = 0; j < OFF_EPT; ++j) mine += x[j];
        const u64 incl = wave_incl_scan_add<u64>(mine);
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        u64 base = carry + incl - mine;
        for (u32 w = 0; w < wv; ++w) base += wtot[w];
#pragma unroll
        for (u32 j = 0; j < OFF_EPT; ++j) {
            if (t0 + j < blk.n_tiles) tile_off[blk.tile_base + t0 + j] = base;
            base += x[j];
        }
        __syncthreads();
        if (tid == 0) carry += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
    if (tid == 0 && carry < blk.n_sym) set_error(blk.err, SHAFA_FILE_UNRECOGNIZABLE);   // ran out of bits
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host: validate the table (prefix-free), build LUT + trie, launch the five kernels
// ------------------------------------------------------------------------------------------------
#include "sfd_host_tables.hpp"

// fn(b) for every block of a launch, on the caller and up to seven helper threads (a helper is worth starting for sixteen
// blocks or more; blocks are handed out by a counter, so the call does not depend on how many helpers could be started)
template <class F> static void for_blocks_parallel(int nblocks, F fn)
{
    std::atomic<int> next{0};
    auto work = [&]() {
        for (;;) {
            const int b = next.fetch_add(1, std::memory_order_relaxed);
            if (b >= nblocks) break;
            fn(b);
        }
    };
    int helpers = nblocks / 16;
    const int hw = (int)std::thread::hardware_concurrency();
    if (helpers > 7) helpers = 7;
    if (hw > 0 && helpers > hw - 1) helpers = hw - 1;
    std::vector<std::thread> th;
    for (int i = 0; i < helpers; ++i) {
        try { th.emplace_back(work); } catch (...) { break; }
    }
    work();
    for (auto &t : th) t.join();
}

static int g_sfd_speculate = 1;                    // 0 never, 1 where spec_worthwhile() says so, 2 wherever the kernels apply
void sfdec_configure(int speculate) { g_sfd_speculate = speculate; }
// test knob ("sf_decode_path"): 0 = the fastest kernels the launch's tables allow; 1 = treat every table as if it were not
// a complete code and 2 = the same: the generic byte-map kernels (sfd_sync / sfd_tiles / sfd_count / sfd_write) that
// otherwise serve incomplete tables and codes of more than 32 bits (the single-code kernels 1 used to select are gone)
static int g_sfd_path = 0;
void sfdec_configure_path(int path) { g_sfd_path = path; }

// Does a decoder that starts 256 bits early agree with the true parse when it reaches the chunk?  Answered per table by
// simulation on random bits (any bit string is a concatenation of code words of a complete code, distributed as the
// code's own lengths imply): the true parser starts at bit 0, a second one at a random offset 1..15; they "merge" when
// the second lands on a start of the first.  Tables whose parsers fail to merge within 256 bits in more than 6 of 32
// trials (a wrong guess costs a lane one more walk of its strip; the exact kernels cost three times the speculative ones, so
// the attempt pays up to a failure rate of a fifth, and a verdict that is too strict sends whole blocks of ordinary data
// to the exact kernels: with "more than 1 of 32" one or two of 128 Zipf blocks did, 0.25 ms per launch) (Zipf-like data: ~0.2 %; uniform bytes, 8/9-bit codes: ~85 %) do not take the speculative kernels.
// Verdicts are cached by a hash of the table (a launch usually repeats the previous launch's tables).
static bool spec_worthwhile(const shafa_code_table &t, const HostTab &h, bool force)
{
    if (!h.ok || !(h.complete || h.complete16 || h.complete32) || h.lmax > 32 || h.lmax < 2) return false;
    if (force) return true;
    u64 key = 1469598103934665603ull;
    for (int s2 = 0; s2 < 256; ++s2) {
        key = (key ^ t.len[s2]) * 1099511628211ull;
        for (int q = 0; q < (t.len[s2] + 7) / 8; ++q) key = (key ^ t.bits[s2][q]) * 1099511628211ull;
    }
    static u64 ckey[1024];
    static signed char cval[1024];                      // 0 = empty, 1 = no, 2 = yes
    static std::mutex cmu;                              // decodes run on several host threads (pipes, layer 1 callers)
    const u32 slot = (u32)(key >> 17) & 1023u;
    {
        std::lock_guard<std::mutex> lk(cmu);
        if (cval[slot] && ckey[slot] == key) return cval[slot] == 2;
    }
    const u32 K1 = h.K1;
    u64 rs = key | 1ull;
    auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
    int fails = 0;
    for (int trial = 0; trial < 32 && fails <= 6; ++trial) {
        u64 w[6];                                       // 384 random bits, bit i = bit (63 - i % 64) of w[i / 64]
        for (u64 &x : w) x = rnd();
        auto window = [&](u32 pos) -> u32 {             // the K1 bits at pos, MSB first
            const u32 wi = pos >> 6, r = pos & 63;
            u64 v = w[wi] << r;
            if (r && wi + 1 < 6) v |= w[wi + 1] >> (64 - r);
            return (u32)(v >> (64 - K1));
        };
        u64 starts[5] = {0, 0, 0, 0, 0};                // code starts of the true parse, bits 0..319
        // (a window that starts a code of more than K1 bits — one random window in 2^K1 — counts as K1 + 1 bits)
        auto len_at = [&](u32 pos) -> u32 { const u32 l = h.lenlut[window(pos)]; return l ? l : K1 + 1; };
        for (u32 pos = 0; pos < 300;) { starts[pos >> 6] |= 1ull << (pos & 63); pos += len_at(pos); }
        u32 pos = 1 + (u32)(rnd() % 15);
        bool merged = false;
        while (pos <= 256) {
            if ((starts[pos >> 6] >> (pos & 63)) & 1ull) { merged = true; break; }
            pos += len_at(pos);
        }
        if (!merged) ++fails;
    }
    const bool yes = fails <= 6;
    std::lock_guard<std::mutex> lk(cmu);
    ckey[slot] = key;
    cval[slot] = yes ? 2 : 1;
    return yes;
}

int sfdec_launch(Batch *bt, hipStream_t st, int nblocks, const u8 *d_in, const u64 *h_in_off,
                 const u64 *h_in_n, const shafa_code_table *h_tables, const u64 *h_n_symbols, u8 *d_out,
                 const u64 *h_out_off)
{
    if (nblocks <= 0) return SHAFA_SUCCESS;
    if (nblocks > bt->max_blocks) return SHAFA_LACK_OF_MEMORY;

    std::vector<HostTab> tabs(nblocks);
    u32 lmax_all = 1, max_tiles = 0, max_l2 = 0;
    u64 total_tiles = 0;
    size_t tab_bytes = 0;
    std::vector<u32> ntiles(nblocks, 0);
    bool pair_all = true;
    bool c16_all = true;
    bool c32_all = true;
    for (int b = 0; b < nblocks; ++b)
        if ((h_in_off[b] & 15) || (h_out_off[b] & 15)) return SHAFA_OUTSIDE_MODULE;
    {   // The tables of a launch are prepared by a few helper threads next to the caller: 10 us a table (prefix check and
        // trie, four look-up tables) is 2 ms for 128 blocks — hidden behind the kernels of 64 MiB blocks, but twice the
        // kernels' time for 8 MiB blocks of compressible data.  Blocks are handed out by a counter, so the launch does
        // not depend on how many helpers could be started.
        std::atomic<bool> nomem{false};
        for_blocks_parallel(nblocks, [&](int b) {
            try { build_host_tab(h_tables[b], tabs[b]); } catch (...) { nomem.store(true); }   // no exception leaves a thread
        });
        if (nomem.load()) return SHAFA_LACK_OF_MEMORY;
    }
    for (int b = 0; b < nblocks; ++b) {
        HostTab &h = tabs[b];
        bool run = h_n_symbols[b] > 0;
        if (run && (!h.ok || h.empty)) {          // malformed table, or single-symbol block
            bt->h_hosterr[b] = SHAFA_FILE_UNRECOGNIZABLE;
            run = false;
        }
        if (run && h_in_n[b] == 0) { bt->h_hosterr[b] = SHAFA_FILE_UNRECOGNIZABLE; run = false; }
        if (!run) continue;
        ntiles[b] = (u32)ceil_div_u64(h_in_n[b], DTILE);
        total_tiles += ntiles[b];
        if (ntiles[b] > max_tiles) max_tiles = ntiles[b];
        if (h.lmax > lmax_all) lmax_all = h.lmax;
        tab_bytes += ((h.lut.size() * 2 + 15) & ~(size_t)15) + ((h.trie.size() * 4 + 15) & ~(size_t)15) +
                     ((h.lenlut.size() + 15) & ~(size_t)15) + ((h.lut2.size() * 2 + 16 + 15) & ~(size_t)15) +
                     ((h.lut13.size() * 2 + 15) & ~(size_t)15) + (h.longtab.empty() ? 0 : (size_t)LONG_BYTES) +
                     (h.long32.empty() ? 0 : (size_t)((LONG32_BYTES + 15) & ~15)) + ((h.lenlut32.size() + 15) & ~(size_t)15);
        if (h.lut2.size() > max_l2) max_l2 = (u32)h.lut2.size();
        pair_all = pair_all && h.complete;
        c16_all = c16_all && h.complete16;
        c32_all = c32_all && (h.complete16 || h.complete32);
    }
    if (!total_tiles) return SHAFA_SUCCESS;
    u32 R = 16;
    while (R < lmax_all) R <<= 1;
    if (g_sfd_path >= 1) pair_all = c16_all = c32_all = false;
    // The kernels of a launch, one family per class of codes:
    //   complete codes of <= 13 bits (what Module T makes)          packed DP sfd_sync16<true> / sfd_countfsm, three codes per look-up
    //   complete codes of 14..16 bits                               the same, the rare long codes from a small LDS table (long_all)
    //   complete codes of 17..32 bits (a real file at -b M)          sfd_sync32 / sfd_countfsm32 (mid32)
    //   anything else (incomplete or > 32 bits: hand-made .cod)      the byte-map kernels sfd_sync / sfd_tiles / sfd_count / sfd_write
    // and in front of the first three the speculative entries (sfd_scan) for tables that re-synchronise.
    const bool long_all = R == 16 && !pair_all && c16_all && lmax_all > (u32)LEN_MAXK;
    const bool packed = R == 16 && ((pair_all && lmax_all <= (u32)LEN_MAXK) || long_all);
    const bool mid32 = !packed && R == 32 && c32_all;
    if (!packed && !mid32 && R < 32) R = 32;           // the byte-map kernels keep 32 entries per chunk at least

    // workspace layout
    size_t off = 0;
    const size_t o_blk = off; off += ((size_t)nblocks * sizeof(DecBlk) + 15) & ~(size_t)15;
    const size_t o_rundp = off; off += ((size_t)nblocks * 4 + 15) & ~(size_t)15;       // staged with the records: initial values
    const size_t o_tab = off; off += tab_bytes;
    const size_t stage_bytes = off;
    const size_t o_tilefn = off; off += (size_t)total_tiles * (R < 8 ? 8 : R); off = (off + 15) & ~(size_t)15;
    const size_t o_tent = off; off += (size_t)total_tiles; off = (off + 15) & ~(size_t)15;
    const size_t o_tcnt = off; off += (size_t)total_tiles * 4; off = (off + 15) & ~(size_t)15;
    const size_t o_toff = off; off += (size_t)total_tiles * 8;
    const size_t o_cent = off; off += (size_t)total_tiles * DEC_THREADS; off = (off + 15) & ~(size_t)15;
    const size_t o_ccnt = off; off += (size_t)total_tiles * DEC_THREADS * 2; off = (off + 15) & ~(size_t)15;
    const bool need_tabs = packed || mid32;
    const size_t o_pair = off; off += (packed && pair_all) ? (size_t)nblocks * (2u << LEN_MAXK) : 0;
    const size_t o_cnt3 = off; off += need_tabs ? (size_t)nblocks * (2u << LEN_MAXK) : 0;
    const size_t o_sym3 = off; off += need_tabs ? (size_t)nblocks * (4u << LEN_MAXK) : 0;
    const size_t o_fsm4 = off; off += need_tabs ? (size_t)nblocks * 16384 : 0;
    const size_t o_fsm1 = off; off += need_tabs ? (size_t)nblocks * 2048 : 0;
    const size_t o_cfn = off; off += packed ? (size_t)total_tiles * DEC_THREADS * 8 : (size_t)total_tiles * R * DEC_THREADS;
    // speculative entries (complete codes, Lmax <= 32, tables that re-synchronise): per-tile guess / exit / redo flag
    // (long_all and mid32 launches too: a code of more than 13 bits is an escape inside the walk)
    const bool spec_path = (packed || mid32) && g_sfd_speculate != 0;
    const int spec_long = mid32 ? 2 : long_all ? 1 : 0;
    // window of sfd_scan's counting tables.  A 13-bit table in a launch with the table of long codes (which holds every code
    // of more than 12 bits, by 12-bit prefix) may count with 12-bit windows: the 13-bit codes become escapes like the 14..16-
    // bit ones, 8 KB of tables instead of 16 fit six workgroups on a CU instead of four (LABNOTES.md §3.2).  Worth it while the
    // 13-bit codes are few (an escape is a binary search that the whole wave waits for): at most eight of them, 0.1 % of the
    // symbols of a block coded near its entropy.
    std::vector<u32> hblk_kw(nblocks, 0);
    for (int b = 0; b < nblocks; ++b) {
        if (!ntiles[b]) continue;
        hblk_kw[b] = spec_window(tabs[b].K1);
        if (spec_long && tabs[b].K1 == 13) {
            u32 n13 = 0;
            for (int sy = 0; sy < 256; ++sy) n13 += h_tables[b].len[sy] == 13;
            if (n13 <= 8) hblk_kw[b] = 12;
        }
    }
    u32 long_used = 0;                                 // bytes of the 13..16-bit codes' table that the launch's blocks fill
    if (long_all) {
        u32 gmax = 0;
        for (int b = 0; b < nblocks; ++b)
            if (ntiles[b] && !tabs[b].longtab.empty() && tabs[b].longtab[0] > gmax) gmax = tabs[b].longtab[0];
        long_used = (16u + (u32)LONG_PFX * 2u + gmax * 32u + 15u) & ~15u;
    } else if (mid32) {                                // header, prefixes, roots, the sub-trie nodes in use (not the DP's root13)
        u32 nmax = 0;
        for (int b = 0; b < nblocks; ++b)
            if (ntiles[b] && !tabs[b].long32.empty() && tabs[b].long32[1] > nmax) nmax = tabs[b].long32[1];
        long_used = (16u + (u32)LONG_PFX * 4u + nmax * 4u + 15u) & ~15u;
    }
    std::vector<char> spec_blk(nblocks, 0);
    bool any_spec = false;
    for (int b = 0; spec_path && b < nblocks; ++b)
        if (ntiles[b] && spec_worthwhile(h_tables[b], tabs[b], g_sfd_speculate == 2)) { spec_blk[b] = 1; any_spec = true; }
    const size_t o_tguess = off; off += any_spec ? (size_t)total_tiles : 0; off = (off + 15) & ~(size_t)15;
    const size_t o_texit = off; off += any_spec ? (size_t)total_tiles : 0; off = (off + 15) & ~(size_t)15;
    int rc = batch_reserve(bt, st, off);
    if (rc) return rc;
    u8 *ws = (u8 *)bt->d_ws;

    // the records, the blocks' run_dp words and the tables travel on the side stream into a parameter buffer of their own
    // (batch_params_*): the copy overlaps the launch before this one instead of sitting in front of sfd_tables
    u8 *dpar = batch_params_begin(bt, stage_bytes);
    if (!dpar) return SHAFA_LACK_OF_MEMORY;
    ParamsScope pscope(bt, st);                        // every return below records the buffer's last reader
    u8 *hs = (u8 *)batch_stage(bt, bt->par_inline ? st : bt->copy_st, stage_bytes);
    if (!hs) return SHAFA_LACK_OF_MEMORY;
    DecBlk *hb = (DecBlk *)hs;
    size_t tpos = o_tab;
    u32 tbase = 0;
    std::vector<size_t> tab_pos(nblocks, 0);
    for (int b = 0; b < nblocks; ++b) {
        DecBlk &e = hb[b];
        memset(&e, 0, sizeof(e));
        e.in = d_in + h_in_off[b];
        e.out = d_out + h_out_off[b];
        e.in_n = h_in_n[b];
        e.n_sym = h_n_symbols[b];
        e.err = bt->d_err + b;
        e.n_tiles = ntiles[b];
        e.tile_base = tbase;
        e.run_dp = any_spec ? (u32 *)(dpar + o_rundp) + b : nullptr;
        ((u32 *)(hs + o_rundp))[b] = spec_blk[b] ? 0u : 1u;               // 1: straight to the exact kernels
        e.pairlut = (packed && pair_all) ? ws + o_pair + (size_t)b * (2u << LEN_MAXK) : nullptr;
        e.cnt3 = need_tabs ? (u16 *)(ws + o_cnt3 + (size_t)b * (2u << LEN_MAXK)) : nullptr;
        e.sym3 = need_tabs ? (u32 *)(ws + o_sym3 + (size_t)b * (4u << LEN_MAXK)) : nullptr;
        e.fsm4 = need_tabs ? (u32 *)(ws + o_fsm4 + (size_t)b * 16384) : nullptr;
        e.fsm1 = need_tabs ? (u32 *)(ws + o_fsm1 + (size_t)b * 2048) : nullptr;
        tbase += ntiles[b];
        if (!ntiles[b]) continue;
        HostTab &h = tabs[b];
        e.K = h.K;
        e.K1 = h.K1;
        e.KW = hblk_kw[b];
        e.lmax = h.lmax;
        e.n_states = (u32)(h.trie.size() / 2);
        tab_pos[b] = tpos;                                  // where this block's tables start in the staging arena
        e.lut2 = (const u16 *)(dpar + tpos);
        e.n_l2 = (u32)h.lut2.size();
        tpos += (h.lut2.size() * 2 + 16 + 15) & ~(size_t)15;
        e.lut13 = (const u16 *)(dpar + tpos);
        tpos += (h.lut13.size() * 2 + 15) & ~(size_t)15;
        if (!h.longtab.empty()) {
            e.longtab = (const u16 *)(dpar + tpos);
            tpos += LONG_BYTES;
        }
        if (!h.long32.empty()) {
            e.long32 = (const u16 *)(dpar + tpos);
            tpos += (LONG32_BYTES + 15) & ~15;
        }
        if (!h.lenlut32.empty()) {
            e.lenlut32 = dpar + tpos;
            tpos += (h.lenlut32.size() + 15) & ~(size_t)15;
        }
        e.lenlut = dpar + tpos;
        if (!e.lenlut32) e.lenlut32 = e.lenlut;            // no code longer than 13 bits: the plain table
        tpos += (h.lenlut.size() + 15) & ~(size_t)15;
        e.lut = (const u16 *)(dpar + tpos);
        tpos += (h.lut.size() * 2 + 15) & ~(size_t)15;
        e.trie = (const u32 *)(dpar + tpos);
        tpos += (h.trie.size() * 4 + 15) & ~(size_t)15;
    }
    // the tables themselves (8-40 KB a block) are copied into the arena by the same helpers that built them: on the caller
    // alone the copies of a launch of 512 small blocks took longer than its kernels
    for_blocks_parallel(nblocks, [&](int b) {
        if (!ntiles[b]) return;
        const HostTab &h = tabs[b];
        size_t q = tab_pos[b];
        if (h.lut2.size()) memcpy(hs + q, h.lut2.data(), h.lut2.size() * 2);
        q += (h.lut2.size() * 2 + 16 + 15) & ~(size_t)15;
        memcpy(hs + q, h.lut13.data(), h.lut13.size() * 2);
        q += (h.lut13.size() * 2 + 15) & ~(size_t)15;
        if (!h.longtab.empty()) { memcpy(hs + q, h.longtab.data(), LONG_BYTES); q += LONG_BYTES; }
        if (!h.long32.empty()) { memcpy(hs + q, h.long32.data(), LONG32_BYTES); q += (LONG32_BYTES + 15) & ~15; }
        if (!h.lenlut32.empty()) { memcpy(hs + q, h.lenlut32.data(), h.lenlut32.size()); q += (h.lenlut32.size() + 15) & ~(size_t)15; }
        memcpy(hs + q, h.lenlut.data(), h.lenlut.size());
        q += (h.lenlut.size() + 15) & ~(size_t)15;
        memcpy(hs + q, h.lut.data(), h.lut.size() * 2);
        q += (h.lut.size() * 2 + 15) & ~(size_t)15;
        memcpy(hs + q, h.trie.data(), h.trie.size() * 4);
    });
    if ((rc = batch_params_commit(bt, st, hs, stage_bytes))) return rc;

    const DecBlk *dblk = (const DecBlk *)(dpar + o_blk);
    const size_t lds_data = (size_t)(DATA_WORDS + DATA_WORDS / 8 + 8) * 4;
    const size_t lds_lut = (size_t)(1u << LUT_MAXK) * 2 + (size_t)((max_l2 + 8) & ~7u) * 2;
    const size_t lds_sync = lds_data + (size_t)R * DEC_THREADS + lds_lut + 4 * R + 64;
    const size_t lds_count = lds_data + (size_t)R * DEC_THREADS + lds_lut + 4 * R + DEC_THREADS + 64;
    const size_t lds_write = lds_data + lds_lut + 64;
    const size_t lds_tiles = (size_t)R * DEC_THREADS + DEC_THREADS;
    if (lds_tiles > 65536 || lds_sync > 65536) {     // long codes (R = 256): more than the default 64 KiB
        HIP_TRY(hipFuncSetAttribute((const void *)sfd_sync, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sync));
        HIP_TRY(hipFuncSetAttribute((const void *)sfd_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_count));
        HIP_TRY(hipFuncSetAttribute((const void *)sfd_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tiles));
    }
    const u32 l2cap = (max_l2 + 8) & ~7u;             // level-2 LDS entries reserved after level 1
    const dim3 grid_t(max_tiles, (u32)nblocks), grid_b((u32)nblocks);
    u32 tpw = 4;                                       // tiles per workgroup of the fast kernels (one table load)
    while (tpw > 1 && (u64)ceil_div_u64(max_tiles, tpw) * nblocks < 2048) tpw >>= 1;     // keep the chip full
    const dim3 grid_f((u32)ceil_div_u64(max_tiles, tpw), (u32)nblocks);
    constexpr int CSUBS = 4;                           // 256-lane groups per workgroup of sfd_countfsm
    const dim3 grid_c((u32)ceil_div_u64(max_tiles, tpw * CSUBS), (u32)nblocks);
    // speculative entries: the guesses of every unit (sfd_scan), then the units the streams end in, the links between all units and
    // the verdict per block (sfd_ends)
    auto launch_spec = [&]() {
        u8 *tg = ws + o_tguess, *tx = ws + o_texit;
        u32 k1_max = 1;
        for (int b = 0; b < nblocks; ++b) if (spec_blk[b] && hblk_kw[b] > k1_max) k1_max = hblk_kw[b];
        const u32 tabb = 2u << k1_max;
        const size_t lds_scan = (size_t)SC_LDS_ROWS + tabb + SC_MISC + (spec_long ? (size_t)long_used : 0);
        const dim3 grid_c((u32)ceil_div_u64(max_tiles, SC_TILES), (u32)nblocks);
        if (spec_long == 2)
            hipLaunchKernelGGL((sfd_scan<2>), grid_c, dim3(DEC_THREADS), lds_scan, st, dblk, ws + o_cent, (u16 *)(ws + o_ccnt),
                               (u32 *)(ws + o_tcnt), tg, tx, tabb, long_used);
        else if (spec_long == 1)
            hipLaunchKernelGGL((sfd_scan<1>), grid_c, dim3(DEC_THREADS), lds_scan, st, dblk, ws + o_cent, (u16 *)(ws + o_ccnt),
                               (u32 *)(ws + o_tcnt), tg, tx, tabb, long_used);
        else
            hipLaunchKernelGGL((sfd_scan<0>), grid_c, dim3(DEC_THREADS), lds_scan, st, dblk, ws + o_cent, (u16 *)(ws + o_ccnt),
                               (u32 *)(ws + o_tcnt), tg, tx, tabb, long_used);
        {   // the units the streams end in and the links between all units: checked, repaired, the blocks' verdicts
            const size_t lds_ends = (size_t)SL_LDS_DATA + tabb + SL_MISC + (spec_long ? (size_t)long_used : 0);
            const dim3 grid_e(1u + (u32)ceil_div_u64(max_tiles, SF_LINKS), (u32)nblocks);
            if (spec_long == 2)
                hipLaunchKernelGGL((sfd_ends<2>), grid_e, dim3(DEC_THREADS), lds_ends, st, dblk, ws + o_cent, (u16 *)(ws + o_ccnt),
                                   (u32 *)(ws + o_tcnt), tg, tx, tabb, long_used);
            else if (spec_long == 1)
                hipLaunchKernelGGL((sfd_ends<1>), grid_e, dim3(DEC_THREADS), lds_ends, st, dblk, ws + o_cent, (u16 *)(ws + o_ccnt),
                                   (u32 *)(ws + o_tcnt), tg, tx, tabb, long_used);
            else
                hipLaunchKernelGGL((sfd_ends<0>), grid_e, dim3(DEC_THREADS), lds_ends, st, dblk, ws + o_cent, (u16 *)(ws + o_ccnt),
                                   (u32 *)(ws + o_tcnt), tg, tx, tabb, long_used);
        }
    };
    // when every block of the launch speculates, the exact kernels are fall-backs that normally return at once:
    // fat workgroups (256 tiles each) make that a launch of a few thousand workgroups instead of a few hundred thousand
    bool all_spec = any_spec;
    for (int b = 0; b < nblocks; ++b) if (ntiles[b] && !spec_blk[b]) all_spec = false;
    const u32 tpw_dp = all_spec ? 256u : tpw;
    const dim3 grid_fd((u32)ceil_div_u64(max_tiles, tpw_dp), (u32)nblocks);
    if (packed) {
        hipLaunchKernelGGL(sfd_tables, dim3((u32)nblocks, 8), dim3(DEC_THREADS), 0, st, dblk);
        if (any_spec) launch_spec();
        const dim3 grid_cd((u32)ceil_div_u64(max_tiles, tpw_dp * CSUBS), (u32)nblocks);
        u32 k1_all = 0;                                // common K1 of the running blocks, 0 when they differ
        for (int b = 0; b < nblocks; ++b)
            if (ntiles[b]) k1_all = (k1_all == 0 || k1_all == tabs[b].K1) ? tabs[b].K1 : 0xFFFFFFFFu;
        if (long_all)
            hipLaunchKernelGGL((sfd_sync16<false, true>), grid_fd, dim3(DEC_THREADS), 0, st, dblk,
                               (u64 *)(ws + o_cfn), (u64 *)(ws + o_tilefn), tpw_dp);
        else if (k1_all == 12)
            hipLaunchKernelGGL((sfd_sync16<true, false, 12>), grid_fd, dim3(DEC_THREADS), 0, st, dblk,
                               (u64 *)(ws + o_cfn), (u64 *)(ws + o_tilefn), tpw_dp);
        else if (k1_all == 13)
            hipLaunchKernelGGL((sfd_sync16<true, false, 13>), grid_fd, dim3(DEC_THREADS), 0, st, dblk,
                               (u64 *)(ws + o_cfn), (u64 *)(ws + o_tilefn), tpw_dp);
        else
            hipLaunchKernelGGL((sfd_sync16<true, false>), grid_fd, dim3(DEC_THREADS), 0, st, dblk,
                               (u64 *)(ws + o_cfn), (u64 *)(ws + o_tilefn), tpw_dp);
        hipLaunchKernelGGL(sfd_tiles16, grid_b, dim3(DEC_THREADS), 0, st, dblk, (const u64 *)(ws + o_tilefn),
                           ws + o_tent);
        hipLaunchKernelGGL((sfd_countfsm<CSUBS>), grid_cd, dim3(DEC_THREADS * CSUBS), 0, st, dblk,
                           (const u64 *)(ws + o_cfn), (const u8 *)(ws + o_tent), ws + o_cent, (u16 *)(ws + o_ccnt),
                           (u32 *)(ws + o_tcnt), tpw_dp);
    } else if (mid32) {
        hipLaunchKernelGGL(sfd_tables, dim3((u32)nblocks, 8), dim3(DEC_THREADS), 0, st, dblk);
        if (any_spec) launch_spec();
        hipLaunchKernelGGL(sfd_sync32, grid_fd, dim3(DEC_THREADS), 0, st, dblk, ws + o_cfn, ws + o_tilefn, tpw_dp);
        hipLaunchKernelGGL(sfd_tiles, grid_b, dim3(DEC_THREADS), lds_tiles, st, dblk, R, (const u8 *)(ws + o_tilefn),
                           ws + o_tent);
        hipLaunchKernelGGL(sfd_countfsm32, grid_fd, dim3(DEC_THREADS), 0, st, dblk, (const u8 *)(ws + o_cfn),
                           (const u8 *)(ws + o_tent), ws + o_cent, (u16 *)(ws + o_ccnt), (u32 *)(ws + o_tcnt), tpw_dp);
    } else {
        hipLaunchKernelGGL(sfd_sync, grid_t, dim3(DEC_THREADS), lds_sync, st, dblk, R, l2cap, ws + o_cfn, ws + o_tilefn);
        hipLaunchKernelGGL(sfd_tiles, grid_b, dim3(DEC_THREADS), lds_tiles, st, dblk, R, (const u8 *)(ws + o_tilefn),
                           ws + o_tent);
        hipLaunchKernelGGL(sfd_count, grid_t, dim3(DEC_THREADS), lds_count, st, dblk, R, l2cap,
                           (const u8 *)(ws + o_cfn), (const u8 *)(ws + o_tent), ws + o_cent, (u16 *)(ws + o_ccnt),
                           (u32 *)(ws + o_tcnt));
    }
    hipLaunchKernelGGL(sfd_offsets, grid_b, dim3(DEC_THREADS), 0, st, dblk, (const u32 *)(ws + o_tcnt),
                       (u64 *)(ws + o_toff));
    u32 ws_tab = 16;                                   // sym3 bytes of the widest table of the launch
    for (int b = 0; b < nblocks; ++b) {
        const u32 k3 = sym3_window(tabs[b].K1);
        if ((4u << k3) > ws_tab) ws_tab = 4u << k3;
    }
    // image: the densest block's average symbols per tile + 1/8 + 1 KiB, 4 .. 40 KiB — unless a smaller margin (1/32 +
    // 256 bytes; a tile that exceeds the image goes in rounds) lets more workgroups share a CU: the symbol pass loses time
    // almost in proportion to the waves it loses, and a CU's LDS is handed out as two halves of 80 KiB (8 workgroups up to
    // 20 KiB each, 6 up to 26.25, 4 up to 40, 2 above: DESIGN.md §3.2).  Codes of up to 12 bits on run-heavy data: 41.7 KB
    // with the wide margin = 2 workgroups per CU, 40 KB with the narrow one = 4.
    const bool ws_esc = mid32 || (packed && (long_all || lmax_all > (u32)SYM3_MAXK));       // the form launched below
    const u32 ws_rows = (u32)ws_rows_bytes(ws_esc);
    u32 ws_cap = 4096, ws_tight = 4096;
    for (int b = 0; b < nblocks; ++b) {
        if (!ntiles[b]) continue;
        const u64 per_tile = ceil_div_u64(h_n_symbols[b], ntiles[b]);
        const u64 c = per_tile + per_tile / 8 + 1024, t = per_tile + per_tile / 32 + 256;
        if (c > ws_cap) ws_cap = (u32)(c > 40960 ? 40960 : c);
        if (t > ws_tight) ws_tight = (u32)(t > 40960 ? 40960 : t);
    }
    {
        const u32 longb = long_used;
        const u32 base = ws_rows + ws_tab + longb + (u32)WS_MISC;
        const u32 most = 65536u - base;                                           // 64 KiB of dynamic LDS
        if (ws_cap > most) ws_cap = most;
        constexpr u32 steps[3] = {20480u, 26880u, 40960u};                        // 8, 6, 4 workgroups per CU as launches show them
                                                                                  //   (tools/ubench/occ_lds.hip: 27136 bytes are 5 already)
        for (u32 k = 0; k < 3; ++k) {
            if (base + ((ws_tight + 15u) & ~15u) <= steps[k]) {                   // this step is within reach of the narrow margin:
                ws_cap = steps[k] - base;                                         //   the image gets all that the step leaves
                if (ws_cap > most) ws_cap = most;                                 //   (more than the wide margin costs nothing:
                break;                                                            //   zeroed once, stored as far as it is used)
            }
        }
    }
    ws_cap &= ~15u;
    const size_t lds_ws = (size_t)ws_rows + ws_tab + ws_cap + WS_MISC;
    // the staged symbol pass takes 16 tiles per workgroup (table fill, image zeroing and the prefetch pipeline's start are
    // paid once per workgroup: 7.8 -> 7.7 ms on the headline data against 4; 32: the same), the chip kept full as above
    u32 tpw_ws = 16;
    while (tpw_ws > 1 && (u64)ceil_div_u64(max_tiles, tpw_ws) * nblocks < 2048) tpw_ws >>= 1;
    const dim3 grid_ws((u32)ceil_div_u64(max_tiles, tpw_ws), (u32)nblocks);
    if (mid32) {
        hipLaunchKernelGGL((sfd_wstage<2, true>), grid_ws, dim3(DEC_THREADS), lds_ws + long_used, st, dblk,
                           (const u8 *)(ws + o_cent), (const u16 *)(ws + o_ccnt), (const u64 *)(ws + o_toff), tpw_ws, ws_tab, ws_cap, long_used);
    } else if (packed) {
        if (long_all)
            hipLaunchKernelGGL((sfd_wstage<1, true>), grid_ws, dim3(DEC_THREADS), lds_ws + long_used, st, dblk,
                               (const u8 *)(ws + o_cent), (const u16 *)(ws + o_ccnt), (const u64 *)(ws + o_toff), tpw_ws, ws_tab, ws_cap, long_used);
        else if (lmax_all > (u32)SYM3_MAXK)
            hipLaunchKernelGGL((sfd_wstage<0, true>), grid_ws, dim3(DEC_THREADS), lds_ws, st, dblk,
                               (const u8 *)(ws + o_cent), (const u16 *)(ws + o_ccnt), (const u64 *)(ws + o_toff), tpw_ws, ws_tab, ws_cap, 0u);
        else
            hipLaunchKernelGGL((sfd_wstage<0, false>), grid_ws, dim3(DEC_THREADS), lds_ws, st, dblk,
                               (const u8 *)(ws + o_cent), (const u16 *)(ws + o_ccnt), (const u64 *)(ws + o_toff), tpw_ws, ws_tab, ws_cap, 0u);
    } else {
        hipLaunchKernelGGL(sfd_write, grid_t, dim3(DEC_THREADS), lds_write, st, dblk, l2cap, (const u8 *)(ws + o_cent),
                           (const u16 *)(ws + o_ccnt), (const u64 *)(ws + o_toff));
    }
    HIP_TRY(hipGetLastError());
    return pscope.done();
}
