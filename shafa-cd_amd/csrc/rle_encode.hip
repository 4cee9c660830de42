// rle_encode.hip — Module F hot path: RLE tokenisation (reference f.c:29-55 block_compression).
//
// Closed form per element, so a run of any length — up to the whole 64 MiB block —
// is tokenised in parallel.  With head(i)/end(i) the bounds of i's maximal run,
//     r = (i - head(i)) mod 255          e = min(255, end(i) - i)
//     r == 0 : emit {0, s, e} if s == 0 or e >= 4, else the literal s
//     r  > 0 : emit the literal s if s != 0 and r + e < 4, else nothing
// r needs the run length that enters a tile from the left, e needs at most 255 bytes of look-ahead (a halo read), the
// output offset of a tile the sizes of all tiles before it.  Two passes over the input by independent workgroups and
// three tiny launches in between deliver them (rle3_* below: no tickets, no look-backs); tokens are staged in LDS and
// stored as aligned 16-byte pieces.  (A single chained pass — one read of the input — was built twice, rounds 1 and 2: on inputs that
// take the mask code it is no faster, a workgroup spends 10 of its 15 us waiting for its ticket, its loads and its two
// look-backs, and on long runs its general tile code is 40 times slower; it needs the deferred look-backs of a
// persistent pipeline like sfe4's to pay, LABNOTES.md §3.)
//
// Algorithmic HBM bytes per block: n read + rle_n written.
#include "common.hpp"
#include "internal.hpp"

#include <stdlib.h>
#include <vector>

namespace {

constexpr int RLE_THREADS = 256;
constexpr int RLE_TILE = RLE_THREADS * 16;
constexpr int RLE_STAGE = 2 * RLE_TILE + 64;

struct RleBlk {
    const u8 *in;
    u8 *out;
    u64 n;
    u64 out_cap;
    u64 *out_n;
    int *err;
    u32 desc_base;
    u32 n_tiles;
    u32 ticket;
    u32 force_general;     // test knob ("rle_encode_general"): every tile takes the per-element general code
    uint2 *masks;          // per 32-byte granule of the block: {E, Z} masks of the first pass for the emit pass (pairs of tiles)
    // one-pass form (rle4_kernel): the block's first super-tile descriptor, its super-tiles, its packed tile histograms
    u32 st_base;
    u32 n_st;
    u32 *th32;             // 128 words per 32 KiB tile of the OUTPUT: counts of bytes 2 j (low half) and 2 j + 1 (high half)
    u64 *freq;             // the block's 256 bins (NULL: no histogram asked for)
    u64 th_bytes;          // bytes of th32 that a launch may touch (zeroed by rle4_zero)
};

struct Seg { u32 f, v; };

// forward: combine(left a, right b) — b's run absorbs a's when b is "all one run that continues"
__device__ __forceinline__ Seg comb_f(Seg a, Seg b) { return Seg{a.f & b.f, b.f ? a.v + b.v : b.v}; }
// backward: combine(left a, right b) — a's run absorbs b's
__device__ __forceinline__ Seg comb_b(Seg a, Seg b) { return Seg{a.f & b.f, a.f ? a.v + b.v : a.v}; }

struct RleShared {
    u32 stage[RLE_STAGE / 4];
    Seg wf[4], wb[4];
    u32 wsum[4];
    u32 R;          // run length entering the tile from the left
    u32 H;          // bytes after the tile equal to its last byte (<= 255)
    u16 E[260];     // fast path: equality masks of the threads, [0] = the 16 bytes before the tile, [257] = after
    u16 lastb[258]; // fast path: last byte of every thread, [0] = the byte before the tile (0x100: none)
    u32 dump[RLE_THREADS];      // fast path: where the byte writes of bytes that emit nothing go
    u32 slow;                   // fast path: some wave asks for the general tile code
};

// General tile code: any input (runs of any length, ragged last tile).  Per element closed form.
// MODE 1: sizes only, carry Rin given, tile total to *Tout; 2: emit, carry Rin and output offset Gin given.
template <int MODE>
__device__ __forceinline__ void rle_tile_general(RleShared &sh, const RleBlk &blk, const int k, u64 Rin, u64 Gin, u32 *Tout)
{
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const u64 n = blk.n;
    const u64 tile_start = (u64)k * RLE_TILE;
    const u64 tile_end = (tile_start + RLE_TILE < n) ? tile_start + RLE_TILE : n;
    const u64 pos = tile_start + (u64)tid * 16;

    // ---- load 16 bytes; bytes past the block end become sentinels that never equal a neighbour ----
    u32 x[16];
    int nvalid = 0;
    if (pos + 16 <= n) {
        const uint4 v = *(const uint4 *)(blk.in + pos);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        nvalid = 16;
    } else {
        nvalid = pos < n ? (int)(n - pos) : 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = (j < nvalid) ? (u32)blk.in[pos + j] : (0x100u | (j & 1));
    }
    const u32 pb = (pos > 0 && pos <= n) ? (u32)blk.in[pos - 1] : 0x200u;
    const u32 nb = (pos + 16 < n) ? (u32)blk.in[pos + 16] : 0x300u;

    // ---- per-thread summaries ------------------------------------------------------------------------
    bool uni = true;
#pragma unroll
    for (int j = 1; j < 16; ++j) uni &= (x[j] == x[0]);
    u32 suf = 1, pre = 1;
    {
        bool go = true;
#pragma unroll
        for (int j = 14; j >= 0; --j) { go &= (x[j] == x[15]); suf += go ? 1u : 0u; }
        go = true;
#pragma unroll
        for (int j = 1; j < 16; ++j) { go &= (x[j] == x[0]); pre += go ? 1u : 0u; }
    }
    const bool eq0 = (x[0] == pb), eq15 = (x[15] == nb);

    // ---- segmented scans over the 256 threads: forward (run length entering from the left),
    //      backward (run length continuing to the right) -------------------------------------------------
    Seg fi = Seg{(u32)(uni && eq0), suf}, bi = Seg{(u32)(uni && eq15), pre};
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        Seg y = Seg{(u32)__shfl_up(fi.f, d, 64), (u32)__shfl_up(fi.v, d, 64)};
        if (lane >= d) fi = comb_f(y, fi);
        Seg z = Seg{(u32)__shfl_down(bi.f, d, 64), (u32)__shfl_down(bi.v, d, 64)};
        if (lane + d < 64) bi = comb_b(bi, z);
    }
    if (lane == 63) sh.wf[wv] = fi;
    if (lane == 0) sh.wb[wv] = bi;
    // exclusive values inside the wave
    Seg fe = Seg{(u32)__shfl_up(fi.f, 1, 64), (u32)__shfl_up(fi.v, 1, 64)};
    if (lane == 0) fe = Seg{1u, 0u};
    Seg be = Seg{(u32)__shfl_down(bi.f, 1, 64), (u32)__shfl_down(bi.v, 1, 64)};
    if (lane == 63) be = Seg{1u, 0u};
    __syncthreads();
    Seg fcar = Seg{1u, 0u}, bcar = Seg{1u, 0u}, ftot = Seg{1u, 0u};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wv) fcar = comb_f(fcar, sh.wf[w]);
        ftot = comb_f(ftot, sh.wf[w]);
    }
#pragma unroll
    for (int w = 3; w >= 0; --w)
        if (w > wv) bcar = comb_b(sh.wb[w], bcar);
    const Seg fex = comb_f(fcar, fe);     // run ending just before this thread (inside the tile)
    const Seg bex = comb_b(be, bcar);     // run starting just after this thread (inside the tile)

    // ---- tile-level: run entering from the previous tile (look-back #1) and halo -----------------------
    if (wv == 0) {
        const bool need_R = __shfl((int)eq0, 0, 64) != 0;   // first byte equals the byte before the tile
        const u64 R = (need_R && k > 0) ? Rin : 0;
        if (tid == 0) sh.R = (u32)(R % 255u);
    } else if (wv == 1) {
        u32 H = 0;
        if (tile_end < n && tile_end == tile_start + RLE_TILE) {
            const u32 lastb = (u32)blk.in[tile_end - 1];
            const u64 q = tile_end + (u64)lane * 4;
            u32 cnt = 0;
            bool go = true;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32 c = (q + j < n) ? (u32)blk.in[q + j] : 0x400u;
                go &= (c == lastb);
                cnt += go ? 1u : 0u;
            }
            const u64 full = __ballot(cnt == 4);
            const int l0 = (~full) ? (__ffsll((unsigned long long)~full) - 1) : 64;
            const u32 c0 = (l0 < 64) ? (u32)__shfl((int)cnt, l0, 64) : 0u;
            H = (u32)l0 * 4 + c0;
            if (H > 255) H = 255;
        }
        if (lane == 0) sh.H = H;
    }
    __syncthreads();

    // ---- per-element r, e, emit size ----------------------------------------------------------------------
    const u32 cin = fex.v + (fex.f ? sh.R : 0u);          // run length (mod-255-safe) before this thread
    const u32 aout = bex.v + (bex.f ? sh.H : 0u);         // run length after this thread
    u32 r[16], e[16];
    r[0] = eq0 ? (cin % 255u) : 0u;
#pragma unroll
    for (int j = 1; j < 16; ++j) {
        const u32 nx = (r[j - 1] + 1 == 255u) ? 0u : r[j - 1] + 1;
        r[j] = (x[j] == x[j - 1]) ? nx : 0u;
    }
    e[15] = 1 + (eq15 ? aout : 0u);
#pragma unroll
    for (int j = 14; j >= 0; --j) e[j] = (x[j] == x[j + 1]) ? e[j + 1] + 1 : 1u;
    u32 emit[16];
    u32 tot = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const u32 ee = e[j] > 255u ? 255u : e[j];
        e[j] = ee;
        u32 m;
        if (r[j] == 0) m = (x[j] == 0 || ee >= 4) ? 3u : 1u;
        else m = (x[j] != 0 && r[j] + ee < 4) ? 1u : 0u;
        if (j >= nvalid) m = 0;
        emit[j] = m;
        tot += m;
    }

    // ---- output offsets: workgroup scan + look-back #2 ------------------------------------------------------
    const u32 incl = wave_incl_scan_add<u32>(tot);
    if (lane == 63) sh.wsum[wv] = incl;
    __syncthreads();
    u32 run = 0, off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w == wv) off = run + incl - tot;
        run += sh.wsum[w];
    }
    const u32 T = run;
    if (MODE == 1) {
        if (tid == 0) *Tout = T;
        return;
    }
    const u64 G = Gin;
    const u32 shift = (u32)G & 3;

    // ---- tokens into LDS (byte offset keeps the global 4-byte phase), then aligned word stores -----------
    u8 *st8 = (u8 *)sh.stage;
    u32 o = shift + off;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (emit[j] == 3) {
            st8[o] = 0; st8[o + 1] = (u8)x[j]; st8[o + 2] = (u8)e[j];
            o += 3;
        } else if (emit[j] == 1) {
            st8[o] = (u8)x[j];
            o += 1;
        }
    }
    __syncthreads();
    const u64 end_b = G + T;
    if (end_b > blk.out_cap) {
        if (tid == 0) set_error(blk.err, SHAFA_LACK_OF_MEMORY);
    } else {
        const u32 nwords = (shift + T + 3) >> 2;
        const u64 gw0 = G >> 2;
        for (u32 w = tid; w < nwords; w += RLE_THREADS) {
            const u32 val = sh.stage[w];
            const u64 byte0 = (gw0 + w) * 4;
            if (byte0 >= G && byte0 + 4 <= end_b) {
                ((u32 *)blk.out)[gw0 + w] = val;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (byte0 + q >= G && byte0 + q < end_b) blk.out[byte0 + q] = (u8)(val >> (8 * q));
            }
        }
    }
}


// bit i of nibble q of mask -> 0x01 in byte i
__device__ __forceinline__ u32 nibf(u32 mask, int q) { return __umul24((mask >> (4 * q)) & 15u, 0x00204081u) & 0x01010101u; }

// 4-bit mask of the bytes of a that equal the bytes of b (exact SWAR zero-byte test of a ^ b)
__device__ __forceinline__ u32 eqmask4(u32 a, u32 b)
{
    return zmask4(a ^ b);
}

// ================================================================================================
// Independent passes (no tickets, no look-backs): every workgroup of every pass is independent.
//   rle3_first   : per tile {is the whole tile one run that continues the previous byte, trailing run length} AND the
//                  emitted bytes of the mask code (which do not depend on the entering run), flagged
//   rle3_carry   : per block, segmented scan -> run length that ends at the last byte before every tile
//   rle3_fix     : emitted bytes of the few tiles that need the general code (flags + carries)
//   rle3_offsets : per block, exclusive scan + block size
//   rle3_emit    : tokens -> LDS image -> aligned 16-byte stores
// rle3_first and rle3_emit work on pairs of tiles with 32 bytes per lane when the mask code is valid for the pair, else
// tile by tile with 16 (rle3_first_tile / rle3_pass_tile<2>); all loads are issued before the first tile is worked on,
// barriers order LDS only.
// The passes use the mask-based tile code when every run around the tile is short enough that the 255-byte
// segmentation cannot matter inside it, the per-element general code otherwise.
// ================================================================================================
struct Rle3Ctx {            // what a thread of the mask-based tile code knows after its loads
    u32 w[4];
    u32 E16, Z16;
};

// loads + masks; returns false when the tile has to take the general code (ragged end of the block).
// sh.E / sh.lastb filled; barrier inside.
// a tile's loads, issued for both tiles of a workgroup before either is worked on (the second tile's memory latency
// passes while the first is processed; nothing is loaded after the first store, so no wave drains its stores for a load)
struct Rle3Pre {
    uint4 v;                    // the thread's 16 bytes
    u32 qb, qa;                 // thread 0: the four bytes before the tile; thread 255: the four bytes after it
    u32 hb[4];                  // wave 1 (lookahead): bytes tile_end + 4 lane + 0..3, 0x400 past the end of the block
    bool full;                  // the tile (and, with lookahead, four bytes more) lies inside the block: mask code possible
};
__device__ __forceinline__ Rle3Pre rle3_preload(const RleBlk &blk, const int k, const bool lookahead)
{
    Rle3Pre p;
    p.v = make_uint4(0, 0, 0, 0);
    p.qb = p.qa = 0;
    p.hb[0] = p.hb[1] = p.hb[2] = p.hb[3] = 0x400u;
    const u64 tile_start = (u64)k * RLE_TILE, tile_end = tile_start + RLE_TILE;
    p.full = (u32)k < blk.n_tiles && tile_end + (lookahead ? 4 : 0) <= blk.n;    // uniform
    if (p.full) {
        p.v = *(const uint4 *)(blk.in + tile_start + (u64)threadIdx.x * 16);
        if (threadIdx.x == 0 && k > 0) p.qb = *(const u32 *)(blk.in + tile_start - 4);
        if (threadIdx.x == RLE_THREADS - 1 && lookahead) p.qa = *(const u32 *)(blk.in + tile_end);
        if (lookahead && (threadIdx.x >> 6) == 1) {     // the 256 bytes after the tile: the count of a run that leaves it
            const u64 q = tile_end + (u64)(threadIdx.x & 63) * 4;
            if (q + 4 <= blk.n) {
                const u32 v = *(const u32 *)(blk.in + q);
                p.hb[0] = v & 0xFFu; p.hb[1] = (v >> 8) & 0xFFu; p.hb[2] = (v >> 16) & 0xFFu; p.hb[3] = v >> 24;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (q + j < blk.n) p.hb[j] = (u32)blk.in[q + j];
            }
        }
    }
    return p;
}

__device__ __forceinline__ bool rle3_masks(RleShared &sh, const int k, Rle3Ctx &c, const bool lookahead, const Rle3Pre &pre,
                                           const u32 force_general = 0)
{
    const int tid = threadIdx.x;
    if (!pre.full) return false;                                         // uniform
    c.w[0] = pre.v.x; c.w[1] = pre.v.y; c.w[2] = pre.v.z; c.w[3] = pre.v.w;
    sh.lastb[tid + 1] = (u16)(c.w[3] >> 24);
    if (tid == 0) {
        u32 e0 = 0, lb = 0x100u;                       // first tile: nothing before it
        if (k > 0) {
            const u32 q = pre.qb;                      // bytes -4 .. -1
            e0 = (eqmask4(q, q << 8) >> 1) << 13;      // E of positions -3, -2, -1 in bits 13..15
            lb = q >> 24;
        }
        sh.E[0] = (u16)e0;
        sh.lastb[0] = (u16)lb;
        sh.slow = force_general;
    }
    if (tid == RLE_THREADS - 1) {
        u32 e = 0;
        if (lookahead) {
            const u32 q = pre.qa;                      // bytes 4096 .. 4099 of the tile's frame
            e = eqmask4(q, (q << 8) | (c.w[3] >> 24)) & 7u;
        }
        sh.E[RLE_THREADS + 1] = (u16)e;
    }
    lds_barrier();
    const u32 pb = sh.lastb[tid];
    u32 E16 = eqmask4(c.w[0], (c.w[0] << 8) | (pb & 0xFFu)) | (eqmask4(c.w[1], __builtin_amdgcn_alignbit(c.w[1], c.w[0], 24)) << 4) |
              (eqmask4(c.w[2], __builtin_amdgcn_alignbit(c.w[2], c.w[1], 24)) << 8) |
              (eqmask4(c.w[3], __builtin_amdgcn_alignbit(c.w[3], c.w[2], 24)) << 12);
    if (pb > 0xFFu) E16 &= ~1u;
    c.E16 = E16;
    c.Z16 = eqmask4(c.w[0], 0) | (eqmask4(c.w[1], 0) << 4) | (eqmask4(c.w[2], 0) << 8) | (eqmask4(c.w[3], 0) << 12);
    sh.E[tid + 1] = (u16)E16;
    return true;
}

__device__ __forceinline__ void rle3_summary_tile(RleShared &sh, const RleBlk &blk, const int k, const Rle3Pre &pre,
                                                  u32 *__restrict__ tsum)
{
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    Rle3Ctx c;
    if (!rle3_masks(sh, k, c, false, pre)) {           // ragged last tile: nothing follows that needs it
        if (tid == 0) tsum[blk.desc_base + k] = 0;
        return;
    }
    const u32 H16 = ~c.E16 & 0xFFFFu;
    const u64 hm = __ballot(H16 != 0);
    int last = -1;                                     // tile-local position of the wave's last run head
    if (hm) {
        const int l = 63 - __builtin_clzll((unsigned long long)hm);
        const u32 hv = (u32)__shfl((int)H16, l, 64);
        last = (wv * 64 + l) * 16 + (31 - __builtin_clz(hv));
    }
    if (lane == 0) sh.wsum[wv] = (u32)last;
    lds_barrier();
    if (tid == 0) {
        int lp = -1;
        for (int ww = 0; ww < 4; ++ww) if ((int)sh.wsum[ww] >= 0) lp = (int)sh.wsum[ww];
        tsum[blk.desc_base + k] = lp < 0 ? (0x80000000u | (u32)RLE_TILE) : (u32)(RLE_TILE - lp);
    }
}

// First pass, summary and sizes in one read of the input: the mask code's emitted size of a tile does not depend on the
// run that enters it — only WHETHER the mask code may be used does (entering run >= 60 bytes) — so the size is computed
// here, before the carries exist, and flagged: bit 31 = the tile needs the general code whatever enters (ragged, long
// runs inside), bit 30 = a run enters (rle3_fix recomputes the size once the carry says it is long).
// bit 29 (pairs only): every byte of the pair is a literal — no zero byte, no run of four reaches into it: the emit pass copies
constexpr u32 T_GENERAL = 0x80000000u, T_ENTERS = 0x40000000u, T_LITERAL = 0x20000000u, T_SIZE = 0xFFFFu;

__device__ __forceinline__ void rle3_first_tile(RleShared &sh, const RleBlk &blk, const int k, const Rle3Pre &pre,
                                                u32 *__restrict__ tsum, u32 *__restrict__ Tarr)
{
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    Rle3Ctx c;
    if (!rle3_masks(sh, k, c, true, pre, blk.force_general)) {   // ragged tile, or the last full one: summary alone, size by rle3_fix
        const Rle3Pre q = rle3_preload(blk, k, false);
        rle3_summary_tile(sh, blk, k, q, tsum);
        if (tid == 0) Tarr[blk.desc_base + k] = T_GENERAL;
        return;
    }
    const u32 E16 = c.E16, Z16 = c.Z16, H16 = ~E16 & 0xFFFFu;
    const u64 hm = __ballot(H16 != 0);
    int last = -1;                                      // tile-local position of the wave's last run head
    if (hm) {
        const int l = 63 - __builtin_clzll((unsigned long long)hm);
        const u32 hv = (u32)__shfl((int)H16, l, 64);
        last = (wv * 64 + l) * 16 + (31 - __builtin_clz(hv));
    }
    const u32 transparent = (u32)__builtin_popcountll((unsigned long long)__ballot(E16 == 0xFFFFu));
    if (lane == 0) {
        sh.wsum[wv] = (u32)last;
        if (transparent > 5) sh.slow = 1;               // a run could reach 255 bytes inside the tile
    }
    lds_barrier();                                      // the neighbours' E masks, the waves' last heads and `slow` are in LDS
    if (tid == 0) {
        int lp = -1;
        for (int ww = 0; ww < 4; ++ww) if ((int)sh.wsum[ww] >= 0) lp = (int)sh.wsum[ww];
        tsum[blk.desc_base + k] = lp < 0 ? (0x80000000u | (u32)RLE_TILE) : (u32)(RLE_TILE - lp);
    }
    const bool slow = sh.slow != 0;
    const bool enters = (sh.E[1] & 1u) != 0;
    const u32 B = ((u32)sh.E[tid] >> 13) | (E16 << 3) | (((u32)sh.E[tid + 2] & 7u) << 19);   // positions -3 .. 18
    const u32 Tm = B & (B >> 1) & (B >> 2);
    const u32 LC = (((Tm >> 1) | Tm | (Tm << 1) | (Tm << 2)) >> 3) & 0xFFFFu;                 // bytes of runs of >= 4
    const u32 Lit = ~Z16 & ~LC & 0xFFFFu;
    const u32 T3 = H16 & (Z16 | LC);
    const u32 tot = (u32)__builtin_popcount(Lit) + 3u * (u32)__builtin_popcount(T3);
    const u32 wtot = wave_reduce_add<u32>(tot);
    lds_barrier();                                      // wsum is read: reuse it for the sizes
    if (lane == 0) sh.wsum[wv] = wtot;
    lds_barrier();
    if (tid == 0)
        Tarr[blk.desc_base + k] = slow ? T_GENERAL : ((sh.wsum[0] + sh.wsum[1] + sh.wsum[2] + sh.wsum[3]) | (enters ? T_ENTERS : 0u));
}

// R[t] = length of the run that ends at the last byte of tile t - 1
constexpr u32 SCAN_EPT = 16;                           // tiles per thread and round of the two per-block scans
constexpr u32 SCAN_THREADS = 1024;                     // one workgroup per block: a 64 MiB block's 16 K tiles are one round (four rounds
                                                       //   of 256 threads — a load and three barriers each — took 18 us per kernel)
constexpr u32 SCAN_WAVES = SCAN_THREADS / 64;
__global__ __launch_bounds__(SCAN_THREADS) void rle3_carry(const RleBlk *__restrict__ blks, const u32 *__restrict__ tsum,
                                                          u32 *__restrict__ R)
{
    // one workgroup per block; a thread takes SCAN_EPT consecutive tiles a round, all of its loads issued before the
    // first is used (64 rounds of one tile per thread, a load and three barriers each, took 60 us for 16 K tiles)
    __shared__ Seg wtot[SCAN_WAVES];
    __shared__ Seg carry;
    const RleBlk blk = blks[blockIdx.x];
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    if (tid == 0) carry = Seg{1u, 0u};
    lds_barrier();
    for (u32 r0 = 0; r0 < blk.n_tiles; r0 += SCAN_THREADS * SCAN_EPT) {
        const u32 t0 = r0 + (u32)tid * SCAN_EPT;
        u32 x[SCAN_EPT];
#pragma unroll
        for (u32 j = 0; j < SCAN_EPT; ++j) x[j] = t0 + j < blk.n_tiles ? tsum[blk.desc_base + t0 + j] : 0x80000000u;
        Seg ex[SCAN_EPT], f = Seg{1u, 0u};              // ex[j]: the tiles of this thread in front of tile j
#pragma unroll
        for (u32 j = 0; j < SCAN_EPT; ++j) {
            ex[j] = f;
            f = comb_f(f, Seg{x[j] >> 31, x[j] & 0x7FFFFFFFu});
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            Seg y = Seg{(u32)__shfl_up(f.f, d, 64), (u32)__shfl_up(f.v, d, 64)};
            if (lane >= d) f = comb_f(y, f);
        }
        if (lane == 63) wtot[wv] = f;
        lds_barrier();
        Seg pre = carry;                                // everything before this wave
        for (int ww = 0; ww < wv; ++ww) pre = comb_f(pre, wtot[ww]);
        Seg lanes = Seg{(u32)__shfl_up(f.f, 1, 64), (u32)__shfl_up(f.v, 1, 64)};      // the lanes in front, this wave
        if (lane == 0) lanes = Seg{1u, 0u};
        pre = comb_f(pre, lanes);
#pragma unroll
        for (u32 j = 0; j < SCAN_EPT; ++j)              // the run that ends just before tile t0 + j
            if (t0 + j < blk.n_tiles) R[blk.desc_base + t0 + j] = comb_f(pre, ex[j]).v;
        lds_barrier();
        if (tid == 0) {
            Seg c2 = carry;
            for (int ww = 0; ww < (int)SCAN_WAVES; ++ww) c2 = comb_f(c2, wtot[ww]);
            carry = c2;
        }
        lds_barrier();
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void rle3_offsets(const RleBlk *__restrict__ blks, const u32 *__restrict__ T,
                                                            u64 *__restrict__ G)
{
    __shared__ u64 wtot[SCAN_WAVES];
    __shared__ u64 carry;
    const RleBlk blk = blks[blockIdx.x];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry = 0;
    lds_barrier();
    for (u32 r0 = 0; r0 < blk.n_tiles; r0 += SCAN_THREADS * SCAN_EPT) {      // as rle3_carry: SCAN_EPT tiles per thread
        const u32 t0 = r0 + tid * SCAN_EPT;
        u32 x[SCAN_EPT];
#pragma unroll
        for (u32 j = 0; j < SCAN_EPT; ++j) x[j] = t0 + j < blk.n_tiles ? T[blk.desc_base + t0 + j] & T_SIZE : 0u;   // flags of rle3_first masked off
        u32 ex[SCAN_EPT], mine = 0;
#pragma unroll
        for (u32 j = 0; j < SCAN_EPT; ++j) {
            ex[j] = mine;
            mine += x[j];
        }
        const u64 incl = wave_incl_scan_add<u64>((u64)mine);
        if (lane == 63) wtot[wv] = incl;
        lds_barrier();
        u64 base = carry + incl - mine;
        for (u32 ww = 0; ww < wv; ++ww) base += wtot[ww];
#pragma unroll
        for (u32 j = 0; j < SCAN_EPT; ++j)
            if (t0 + j < blk.n_tiles) G[blk.desc_base + t0 + j] = base + ex[j];
        lds_barrier();
        if (tid == 0) {
            u64 c2 = carry;
            for (u32 ww = 0; ww < SCAN_WAVES; ++ww) c2 += wtot[ww];
            carry = c2;
        }
        lds_barrier();
    }
    if (tid == 0) *blk.out_n = carry;
}

template <int MODE>                                     // 1: sizes -> T, 2: emit at G
__device__ __forceinline__ void rle3_pass_tile(RleShared &sh, const RleBlk &blk, const int k, const Rle3Pre &pre, const u32 Rin,
                                               const u64 Gin, u32 *__restrict__ Tarr)
{
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    Rle3Ctx c;
    bool fast = rle3_masks(sh, k, c, true, pre, blk.force_general);
    if (fast) {
        // mask code only when no run around the tile can reach 255 bytes inside it: at most 5 threads per wave lie
        // wholly inside a run (<= 10 across a wave edge: 192 bytes with both ends) and the entering run is < 60 bytes
        const u32 transparent = (u32)__builtin_popcountll((unsigned long long)__ballot(c.E16 == 0xFFFFu));
        const bool enters = (sh.E[1] & 1u) != 0;
        // (a barrier that does not drain the second tile's loads: flag in LDS instead of __syncthreads_or)
        if (lane == 0 && ((transparent > 5) || (enters && Rin >= 60))) sh.slow = 1;
        lds_barrier();
        if (sh.slow) fast = false;
    }
    if (!fast) {
        __syncthreads();
        rle_tile_general<MODE>(sh, blk, k, Rin, Gin, Tarr + blk.desc_base + k);
        return;
    }
    const u32 E16 = c.E16, Z16 = c.Z16, H16 = ~E16 & 0xFFFFu;
    const u32 B = ((u32)sh.E[tid] >> 13) | (E16 << 3) | (((u32)sh.E[tid + 2] & 7u) << 19);   // positions -3 .. 18
    const u32 Tm = B & (B >> 1) & (B >> 2);
    const u32 LC = (((Tm >> 1) | Tm | (Tm << 1) | (Tm << 2)) >> 3) & 0xFFFFu;                 // bytes of runs of >= 4
    const u32 Lit = ~Z16 & ~LC & 0xFFFFu;
    const u32 T3 = H16 & (Z16 | LC);
    const u32 tot = (u32)__builtin_popcount(Lit) + 3u * (u32)__builtin_popcount(T3);
    const u32 incl = wave_incl_scan_add<u32>(tot);
    if (lane == 63) sh.wsum[wv] = incl;
    // distance from the end of this thread's bytes to the next run head (for the count of a run that leaves the thread)
    u32 after = 0;
    if (MODE == 2) {
        const u64 hm = __ballot(H16 != 0);              // never 0: at most 5 threads of a wave have no head
        const u64 above = lane == 63 ? 0ull : hm & ~((2ull << lane) - 1);
        const int l2 = above ? __builtin_ctzll((unsigned long long)above) : 0;
        const u32 hv = (u32)__shfl((int)H16, l2, 64);
        if (above) after = 16u * (u32)(l2 - lane - 1) + (u32)__builtin_ctz(hv);
        if (lane == 0) {
            const int l0 = __builtin_ctzll((unsigned long long)hm);
            sh.wf[wv].v = 16u * (u32)l0;                // + ctz of that lane's heads, added below
        }
        const int l0 = __builtin_ctzll((unsigned long long)hm);
        const u32 h0 = (u32)__shfl((int)H16, l0, 64);
        if (lane == 0) sh.wf[wv].v += (u32)__builtin_ctz(h0);          // first head of this wave, in bytes from its start
        if (wv == 1) {                                  // halo: bytes after the tile equal to its last byte (<= 255)
            const u32 lastb = sh.lastb[RLE_THREADS];
            u32 cnt = 0;
            bool go = true;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                go &= (pre.hb[j] == lastb);
                cnt += go ? 1u : 0u;
            }
            const u64 full = __ballot(cnt == 4);
            const int f0 = (~full) ? (__ffsll((unsigned long long)~full) - 1) : 64;
            const u32 c0 = (f0 < 64) ? (u32)__shfl((int)cnt, f0, 64) : 0u;
            u32 H = (u32)f0 * 4 + c0;
            if (H > 255) H = 255;
            if (lane == 0) sh.H = H;
        }
        (void)above;
    }
    lds_barrier();
    u32 run = 0, off = 0;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
        if (ww == wv) off = run + incl - tot;
        run += sh.wsum[ww];
    }
    const u32 Tt = run;
    if (MODE == 1) {
        if (tid == 0) Tarr[blk.desc_base + k] = Tt;
        return;
    }
    if (MODE == 2) {
        const u64 hm = __ballot(H16 != 0);
        const bool has_above = lane != 63 && (hm & ~((2ull << lane) - 1)) != 0;
        if (!has_above) after = 16u * (u32)(63 - lane) + (wv < 3 ? sh.wf[wv + 1].v : sh.H);
    }
    const u64 G = Gin;
    const u32 shift = (u32)((uintptr_t)(blk.out + G) & 15u);   // the staging buffer is aligned like the output address
    u8 *st8 = (u8 *)sh.stage;
    const u32 p0 = shift + off;
    {   // every byte is written somewhere, no branches: a literal at the running position, the symbol of a triple in the
        // triple's second place, a byte that emits nothing into the lane's dump word
        u32 p = p0;
        u8 *dp = (u8 *)&sh.dump[tid];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 ft = nibf(T3, q), sz = nibf(Lit, q) | ft | (ft << 1);      // per byte: 1 at a literal, 3 at a triple head
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = 4 * q + b;
                u8 *at = (Lit >> j) & 1u ? st8 + p : ((T3 >> j) & 1u ? st8 + p + 1 : dp);
                *at = (u8)(c.w[q] >> (8 * b));
                p += (sz >> (8 * b)) & 0xFFu;
            }
        }
    }
    for (u32 t = T3; t; t &= t - 1) {                   // triples: the lane's heads in order
        const u32 j = (u32)__builtin_ctz(t), below = (1u << j) - 1u;
        const u32 hn = j < 15 ? H16 >> (j + 1) : 0u;
        u32 L = hn ? (u32)__builtin_ctz(hn) + 1u : (16u - j) + after;   // to the next head: in this thread / beyond it
        L = L > 255u ? 255u : L;
        const u32 at = p0 + (u32)__builtin_popcount(Lit & below) + 3u * (u32)__builtin_popcount(T3 & below);
        st8[at] = 0;
        st8[at + 2] = (u8)L;
    }
    lds_barrier();
    const u64 end_b = G + Tt;
    if (end_b > blk.out_cap) {
        if (tid == 0) set_error(blk.err, SHAFA_LACK_OF_MEMORY);
    } else {                                            // aligned 16-byte pieces, non-temporal
        u8 *gbase = blk.out + G - shift;
        const u32 npieces = (shift + Tt + 15) >> 4;
        for (u32 u = tid; u < npieces; u += RLE_THREADS) {
            const uint4 v = ((const uint4 *)sh.stage)[u];
            const u32 lo = 16 * u;
            if (lo >= shift && lo + 16 <= shift + Tt) {
                gstore_nt<uint4>(gbase + lo, v);
            } else {
                const u32 wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (u32 q = 0; q < 16; ++q)
                    if (lo + q >= shift && lo + q < shift + Tt) gbase[lo + q] = (u8)(wds[q >> 2] >> (8 * (q & 3)));
            }
        }
    }
}

// ================================================================================================
// rle3_emit: the emit pass on PAIRS of tiles.  When the pair (8 KiB) is full, at most two 32-byte lanes of a wave lie
// wholly inside a run and the run that enters is shorter than 60 bytes (no run can reach 255 bytes inside the pair), the
// mask code runs on 32 bytes per lane — half the scans, ballots and flush bookkeeping per byte of the 16-byte code:
// E / Z masks, runs of >= 4 from three consecutive E bits, per-byte sizes by SWAR, literals as byte writes at running
// positions, the symbol of a triple written by the same loop, {0, length} by a loop over the lane's triple heads, the
// image aligned like the output address, aligned non-temporal 16-byte stores.  Otherwise the two tiles take
// rle3_pass_tile<2> one after the other.
// ================================================================================================
constexpr int R8_BPL = 32;
constexpr int R8_TILE = RLE_THREADS * R8_BPL;      // a pair of tiles
constexpr int R8_IMG = 2 * R8_TILE + 64;           // the mask code emits at most two bytes per byte; + alignment shift

struct R8Fast {
    u8 img[R8_IMG];
    u32 E[RLE_THREADS + 2];        // E masks of the lanes; [0] bits 29..31: the three bytes before the pair, [257] bits 0..2: after
    u32 dump[RLE_THREADS];
    u32 wsum[4];
    u32 wfirst[4];                 // bytes from the start of a wave to its first run head
    u32 wlast[4];                  // last byte of each wave
    u32 wpure[4];                  // first pass: every byte of the wave is a literal
    u32 slow;
    u32 H;                         // bytes after the pair equal to its last byte (<= 255)
};
union R8Shared {
    RleShared gen;
    R8Fast f;
};

__device__ __forceinline__ bool rle3_emit8k(R8Fast &sh, const RleBlk &blk, const int kp, const u32 Rin, const u64 G)
{
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const u64 n = blk.n;
    const u64 tile_start = (u64)kp * R8_TILE, tile_end = tile_start + R8_TILE;
    if (tile_end + 4 > n) return false;                // ragged / last pair (uniform)
    u8 *smem = (u8 *)&sh;
    const u64 pos = tile_start + (u64)tid * R8_BPL;
    u32 w[8];
    {
        const uint4 v0 = gload_nt<uint4>(blk.in + pos), v1 = gload_nt<uint4>(blk.in + pos + 16);
        w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w; w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
    }
    const uint2 mz = gload_nt<uint2>(blk.masks + (size_t)kp * RLE_THREADS + tid);
    // everything else the pair needs from memory is requested now as well
    u32 hq[4] = {0x400u, 0x400u, 0x400u, 0x400u};      // wave 2: bytes tile_end + 4 lane + 0..3 (0x400 past the block)
    u32 qb = 0, qa = 0;
    if (wv == 2) {
        const u64 q = tile_end + (u64)lane * 4;
        if (q + 4 <= n) {
            const u32 v = *(const u32 *)(blk.in + q);
            hq[0] = v & 0xFFu; hq[1] = (v >> 8) & 0xFFu; hq[2] = (v >> 16) & 0xFFu; hq[3] = v >> 24;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (q + j < n) hq[j] = (u32)blk.in[q + j];
        }
    }
    if (tid == 0 && kp > 0) qb = *(const u32 *)(blk.in + tile_start - 4);          // bytes -4 .. -1
    if (tid == RLE_THREADS - 1) qa = *(const u32 *)(blk.in + tile_end);            // the four bytes after the pair
    if (lane == 63) sh.wlast[wv] = w[7] >> 24;
    if (tid == 0) {
        u32 e0 = 0;
        if (kp > 0) e0 = (zmask4(qb ^ (qb << 8)) >> 1) << 29;      // E of positions -3, -2, -1 in bits 29..31
        sh.E[0] = e0;
        sh.slow = blk.force_general;
    }
    if (tid == RLE_THREADS - 1) sh.E[RLE_THREADS + 1] = zmask4(qa ^ ((qa << 8) | (w[7] >> 24))) & 7u;
    lds_barrier();
    const u32 E = mz.x, Z = mz.y;                       // "equals the byte before" and "is zero" per byte: from the first pass
    sh.E[tid + 1] = E;
    const u32 H = ~E;                                   // run heads
    const u64 hm = __ballot(H != 0);
    {
        // lanes wholly inside a run: more than two in a wave and a run could reach 255 bytes here
        const u32 transparent = (u32)__builtin_popcountll((unsigned long long)__ballot(E == 0xFFFFFFFFu));
        const bool enters = wv == 0 && (__shfl((int)E, 0, 64) & 1) != 0;
        if (lane == 0 && (transparent > 2 || (enters && Rin >= 60))) sh.slow = 1;
        if (hm) {
            const int l0 = __builtin_ctzll((unsigned long long)hm);
            const u32 h0 = (u32)__shfl((int)H, l0, 64);
            if (lane == 0) sh.wfirst[wv] = 32u * (u32)l0 + (u32)__builtin_ctz(h0);
        }
        if (wv == 2) {                                  // halo: bytes after the pair equal to its last byte (<= 255)
            const u32 lastb = sh.wlast[3];
            u32 cnt = 0;
            bool go = true;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                go &= (hq[j] == lastb);
                cnt += go ? 1u : 0u;
            }
            const u64 full = __ballot(cnt == 4);
            const int f0 = (~full) ? (__ffsll((unsigned long long)~full) - 1) : 64;
            const u32 c0 = (f0 < 64) ? (u32)__shfl((int)cnt, f0, 64) : 0u;
            u32 Hh = (u32)f0 * 4 + c0;
            if (Hh > 255) Hh = 255;
            if (lane == 0) sh.H = Hh;
        }
    }
    lds_barrier();
    if (sh.slow) return false;

    // ---- classification: bytes of runs of >= 4, literals, triple heads ----------------------------------------
    const u64 B = (u64)(sh.E[tid] >> 29) | ((u64)E << 3) | ((u64)(sh.E[tid + 2] & 7u) << 35);   // positions -3 .. 34
    const u64 T = B & (B >> 1) & (B >> 2);
    const u32 LC = (u32)(((T >> 1) | T | (T << 1) | (T << 2)) >> 3);
    const u32 Lit = ~Z & ~LC;
    const u32 T3 = H & (Z | LC);
    const u32 tot = (u32)__builtin_popcount(Lit) + 3u * (u32)__builtin_popcount(T3);
    const u32 incl = wave_incl_scan_add<u32>(tot);
    if (lane == 63) sh.wsum[wv] = incl;
    // bytes from the end of this lane to the next run head (for the count of a run that leaves the lane)
    u32 after;
    {
        const u64 above = lane == 63 ? 0ull : hm & ~((2ull << lane) - 1);
        const int l2 = above ? __builtin_ctzll((unsigned long long)above) : 0;
        const u32 hv = (u32)__shfl((int)H, l2, 64);
        after = above ? 32u * (u32)(l2 - lane - 1) + (u32)__builtin_ctz(hv) : 0xFFFFFFFFu;
    }
    lds_barrier();
    u32 run = 0, off = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q == wv) off = run + incl - tot;
        run += sh.wsum[q];
    }
    const u32 Tt = run;
    if (after == 0xFFFFFFFFu) after = 32u * (u32)(63 - lane) + (wv < 3 ? sh.wfirst[wv + 1] : sh.H);

    // ---- tokens into the image, which is aligned like the output address ---------------------------------------
    const u32 shift = (u32)((uintptr_t)(blk.out + G) & 15u);
    const u32 p0 = (u32)offsetof(R8Fast, img) + shift + off;
    {   // every byte is written somewhere, no branches: a literal at the running position p, the symbol of a triple in
        // the triple's second place, and a byte that emits nothing ALSO at p — where the lane's next token will
        // overwrite it (the symbol and, below, the 0 and the count of a triple are written later in program order) —
        // unless no token follows in this lane (p has reached the lane's end): then into the lane's dump word
        u32 p = p0;
        const u32 p_end = p0 + tot;
        const u32 dump = (u32)offsetof(R8Fast, dump) + 4u * (u32)tid;
#pragma unroll
        for (int q = 0; q < R8_BPL / 4; ++q) {
            const u32 ft = nibf(T3, q), sz = nibf(Lit, q) | ft | (ft << 1);      // per byte: 1 at a literal, 3 at a triple head
            u32 a;
            a = add_byte<0>(p, ft); smem[a < p_end ? a : dump] = (u8)w[q];         p = add_byte<0>(p, sz);
            a = add_byte<1>(p, ft); smem[a < p_end ? a : dump] = (u8)(w[q] >> 8);  p = add_byte<1>(p, sz);
            a = add_byte<2>(p, ft); smem[a < p_end ? a : dump] = (u8)(w[q] >> 16); p = add_byte<2>(p, sz);
            a = add_byte<3>(p, ft); smem[a < p_end ? a : dump] = (u8)(w[q] >> 24); p = add_byte<3>(p, sz);
        }
    }
    u32 p3 = p0;                                        // p0 + 3 * (triples of the lane so far): the same count in every lane
    for (u32 t = T3; t; t &= t - 1, p3 += 3u) {         // triples: the lane's heads in order
        const u32 j = (u32)__builtin_ctz(t), below = (1u << j) - 1u;
        const u32 hn = j < 31 ? H >> (j + 1) : 0u;
        u32 L = hn ? (u32)__builtin_ctz(hn) + 1u : (32u - j) + after;   // to the next head: in this lane / beyond it
        L = L > 255u ? 255u : L;
        const u32 at = p3 + (u32)__builtin_popcount(Lit & below);
        smem[at] = 0;
        smem[at + 2] = (u8)L;
    }
    lds_barrier();
    const u64 end_b = G + Tt;
    if (end_b > blk.out_cap) {
        if (tid == 0) set_error(blk.err, SHAFA_LACK_OF_MEMORY);
        return true;
    }
    u8 *gbase = blk.out + G - shift;                    // aligned 16-byte pieces, non-temporal
    const u32 npieces = (shift + Tt + 15) >> 4;
    for (u32 u = tid; u < npieces; u += RLE_THREADS) {
        const uint4 v = ((const uint4 *)sh.img)[u];
        const u32 lo = 16 * u;
        if (lo >= shift && lo + 16 <= shift + Tt) {
            gstore_nt<uint4>(gbase + lo, v);
        } else {
            const u32 wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (u32 q = 0; q < 16; ++q)
                if (lo + q >= shift && lo + q < shift + Tt) gbase[lo + q] = (u8)(wds[q >> 2] >> (8 * (q & 3)));
        }
    }
    return true;
}

// A pair whose every byte is a literal (T_LITERAL of the first pass: text-like data) leaves as it came: 32 bytes per lane
// into LDS at their own alignment, out again as 16-byte pieces aligned like the OUTPUT address (five dwords funnel-shifted
// by the misalignment of G) — no masks, no classification, no token loop.
__device__ __forceinline__ void rle3_copy8k(R8Fast &sh, const RleBlk &blk, const int kp, const u64 G)
{
    const u32 tid = threadIdx.x;
    const u64 pos = (u64)kp * R8_TILE + (u64)tid * R8_BPL;
    const uint4 v0 = gload_nt<uint4>(blk.in + pos), v1 = gload_nt<uint4>(blk.in + pos + 16);
    u8 *smem = (u8 *)&sh;
    constexpr u32 base = (u32)offsetof(R8Fast, img) + 16u;             // 16 bytes in front: a piece may start before the pair
    *(uint4 *)(smem + base + 32u * tid) = v0;
    *(uint4 *)(smem + base + 32u * tid + 16u) = v1;
    lds_barrier();
    if (G + R8_TILE > blk.out_cap) {
        if (tid == 0) set_error(blk.err, SHAFA_LACK_OF_MEMORY);
        return;
    }
    u8 *gout = blk.out + G;
    const u32 mis = (u32)((uintptr_t)gout & 15u);
    for (u32 u = tid; 16 * u < mis + (u32)R8_TILE; u += RLE_THREADS) {
        const u32 s0 = base + 16 * u - mis, sb = s0 & ~3u, sf = s0 & 3u;
        const u32 d0 = *(const u32 *)__builtin_assume_aligned(smem + sb, 4), d1 = *(const u32 *)__builtin_assume_aligned(smem + sb + 4, 4),
                  d2 = *(const u32 *)__builtin_assume_aligned(smem + sb + 8, 4), d3 = *(const u32 *)__builtin_assume_aligned(smem + sb + 12, 4),
                  d4 = *(const u32 *)__builtin_assume_aligned(smem + sb + 16, 4);
        const u32 wds[4] = {__builtin_amdgcn_alignbyte(d1, d0, sf), __builtin_amdgcn_alignbyte(d2, d1, sf),
                            __builtin_amdgcn_alignbyte(d3, d2, sf), __builtin_amdgcn_alignbyte(d4, d3, sf)};
        u8 *ga = gout - mis + 16 * u;
        const u32 lo = 16 * u;                          // the piece is bytes [lo - mis, lo - mis + 16) of the pair
        if (lo >= mis && lo + 16 <= mis + (u32)R8_TILE) {
            gstore_nt<uint4>(ga, make_uint4(wds[0], wds[1], wds[2], wds[3]));
        } else {
#pragma unroll
            for (u32 q = 0; q < 16; ++q)
                if (lo + q >= mis && lo + q < mis + (u32)R8_TILE) ga[q] = (u8)(wds[q >> 2] >> (8 * (q & 3)));
        }
    }
}

__global__ __launch_bounds__(RLE_THREADS) void rle3_emit(const RleBlk *__restrict__ blks, const u32 *__restrict__ Rarr,
                                                         u32 *__restrict__ Tarr, const u64 *__restrict__ Garr)
{
    __shared__ __attribute__((aligned(16))) R8Shared sh;
    const RleBlk blk = blks[blockIdx.y];
    const int k0 = 2 * (int)blockIdx.x;
    if ((u32)k0 >= blk.n_tiles) return;
    const bool two = (u32)(k0 + 1) < blk.n_tiles;
    const u32 R0 = Rarr[blk.desc_base + k0];
    const u64 G0 = Garr[blk.desc_base + k0];
    if (!blk.force_general && (Tarr[blk.desc_base + k0] & T_LITERAL)) {    // (uniform) set for full pairs only
        rle3_copy8k(sh.f, blk, (int)blockIdx.x, G0);
        return;
    }
    if (rle3_emit8k(sh.f, blk, (int)blockIdx.x, R0, G0)) return;
    // the pair needs the 16-byte code (or the general code): tile by tile
    lds_barrier();
    const Rle3Pre p0 = rle3_preload(blk, k0, true), p1 = rle3_preload(blk, k0 + 1, true);
    const u32 R1 = two ? Rarr[blk.desc_base + k0 + 1] : 0u;
    const u64 G1 = two ? Garr[blk.desc_base + k0 + 1] : 0ull;
    rle3_pass_tile<2>(sh.gen, blk, k0, p0, R0, G0, Tarr);
    if (two) {
        lds_barrier();
        rle3_pass_tile<2>(sh.gen, blk, k0 + 1, p1, R1, G1, Tarr);
    }
}

// the first pass on a pair of tiles, 32 bytes per lane: both tiles' run summaries and mask-code sizes (same validity rule
// as rle3_emit8k, minus the entering run, which is not known yet: that is what T_ENTERS is for)
struct R8In {
    u32 w[8];
    u32 qb, qa;
    bool full;
};
__device__ __forceinline__ R8In rle3_preload8k(const RleBlk &blk, const int kp)
{
    R8In p;
    const u64 tile_start = (u64)kp * R8_TILE, tile_end = tile_start + R8_TILE;
    p.full = tile_end + 4 <= blk.n;                     // uniform
    p.qb = p.qa = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) p.w[i] = 0;
    if (p.full) {
        const u64 pos = tile_start + (u64)threadIdx.x * R8_BPL;
        const uint4 v0 = gload_nt<uint4>(blk.in + pos), v1 = gload_nt<uint4>(blk.in + pos + 16);
        p.w[0] = v0.x; p.w[1] = v0.y; p.w[2] = v0.z; p.w[3] = v0.w; p.w[4] = v1.x; p.w[5] = v1.y; p.w[6] = v1.z; p.w[7] = v1.w;
        if (threadIdx.x == 0 && kp > 0) p.qb = *(const u32 *)(blk.in + tile_start - 4);
        if (threadIdx.x == RLE_THREADS - 1) p.qa = *(const u32 *)(blk.in + tile_end);
    }
    return p;
}

__device__ __forceinline__ bool rle3_first8k(R8Fast &sh, const RleBlk &blk, const int kp, const R8In &in, u32 *__restrict__ tsum,
                                             u32 *__restrict__ Tarr)
{
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    if (!in.full) return false;                         // uniform
    const u32 *w = in.w;
    if (lane == 63) sh.wlast[wv] = w[7] >> 24;
    u32 pb0 = 0x100u;
    if (tid == 0) {
        u32 e0 = 0;
        if (kp > 0) {
            e0 = (zmask4(in.qb ^ (in.qb << 8)) >> 1) << 29;
            pb0 = in.qb >> 24;
        }
        sh.E[0] = e0;
        sh.slow = blk.force_general;
    }
    if (tid == RLE_THREADS - 1) sh.E[RLE_THREADS + 1] = zmask4(in.qa ^ ((in.qa << 8) | (w[7] >> 24))) & 7u;
    lds_barrier();
    u32 pb = (u32)__shfl_up((int)(w[7] >> 24), 1, 64);
    if (lane == 0) pb = wv ? sh.wlast[wv - 1] : pb0;
    u32 dif[8];                                         // every byte xor the byte before it
    dif[0] = w[0] ^ ((w[0] << 8) | (pb & 0xFFu));
#pragma unroll
    for (int i = 1; i < 8; ++i) dif[i] = w[i] ^ __builtin_amdgcn_alignbit(w[i], w[i - 1], 24);
    u32 E = zmask32(dif);
    const u32 Z = zmask32(in.w);
    if (pb > 0xFFu) E &= ~1u;
    sh.E[tid + 1] = E;
    // the two masks are a third of this pass's arithmetic and the emit pass needs the same ones: 8 bytes per 32 of
    // input through the workspace cost it less than computing them again
    gstore<uint2>(blk.masks + (size_t)kp * RLE_THREADS + tid, make_uint2(E, Z));
    const u32 H = ~E;
    const u64 hm = __ballot(H != 0);
    const u32 transparent = (u32)__builtin_popcountll((unsigned long long)__ballot(E == 0xFFFFFFFFu));
    if (lane == 0) {
        if (transparent > 2) sh.slow = 1;
        int last = -1;                                  // position of the wave's last run head inside its 4 KiB tile
        if (hm) {
            const int l = 63 - __builtin_clzll((unsigned long long)hm);
            last = l;                                   // lane; its head position is added below
        }
        sh.wfirst[wv] = (u32)last;
    }
    {
        const int l = hm ? 63 - __builtin_clzll((unsigned long long)hm) : 0;
        const u32 hv = (u32)__shfl((int)H, l, 64);
        if (lane == 0 && hm) sh.wfirst[wv] = (u32)(((wv & 1) * 64 + l) * R8_BPL + (31 - __builtin_clz(hv)));
    }
    lds_barrier();                                      // the neighbours' E masks, the waves' last heads and `slow` are in LDS
    if (sh.slow) return false;
    const u64 B = (u64)(sh.E[tid] >> 29) | ((u64)E << 3) | ((u64)(sh.E[tid + 2] & 7u) << 35);   // positions -3 .. 34
    const u64 T = B & (B >> 1) & (B >> 2);
    const u32 LC = (u32)(((T >> 1) | T | (T << 1) | (T << 2)) >> 3);
    const u32 Lit = ~Z & ~LC;
    const u32 T3 = H & (Z | LC);
    const u32 wtot = wave_reduce_add<u32>((u32)__builtin_popcount(Lit) + 3u * (u32)__builtin_popcount(T3));
    const bool wpure = __ballot(Lit != 0xFFFFFFFFu) == 0ull;          // text-like data: nearly every pair
    if (lane == 0) { sh.wsum[wv] = wtot; sh.wpure[wv] = wpure ? 1u : 0u; }
    const bool ent0 = (sh.E[1] & 1u) != 0, ent1 = (sh.E[129] & 1u) != 0;
    lds_barrier();
    if (tid < 2) {                                      // thread t writes the summary and the size of tile 2 kp + t
        const int k = 2 * kp + tid;
        const int lp = (int)sh.wfirst[2 * tid + 1] >= 0 ? (int)sh.wfirst[2 * tid + 1] : (int)sh.wfirst[2 * tid];
        const u32 pure = (sh.wpure[0] & sh.wpure[1] & sh.wpure[2] & sh.wpure[3]) ? T_LITERAL : 0u;   // (a run of >= 60 that enters
        tsum[blk.desc_base + k] = lp < 0 ? (0x80000000u | (u32)RLE_TILE) : (u32)(RLE_TILE - lp);     //  would cover byte 0: not pure)
        Tarr[blk.desc_base + k] = (sh.wsum[2 * tid] + sh.wsum[2 * tid + 1]) | ((tid ? ent1 : ent0) ? T_ENTERS : 0u) | pure;
    }
    return true;
}

__global__ __launch_bounds__(RLE_THREADS) void rle3_first(const RleBlk *__restrict__ blks, u32 *__restrict__ tsum, u32 *__restrict__ Tarr)
{
    __shared__ __attribute__((aligned(16))) R8Shared sh;
    const RleBlk blk = blks[blockIdx.y];
    const int kp0 = 2 * (int)blockIdx.x;                // two pairs of tiles per workgroup, both loaded up front
    if ((u32)(2 * kp0) >= blk.n_tiles) return;
    const R8In a = rle3_preload8k(blk, kp0), b = rle3_preload8k(blk, kp0 + 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kp = kp0 + i, k0 = 2 * kp;
        if ((u32)k0 >= blk.n_tiles) break;
        if (i) lds_barrier();
        if (rle3_first8k(sh.f, blk, kp, i ? b : a, tsum, Tarr)) continue;
        lds_barrier();                                  // the pair takes the 16-byte code, tile by tile
        const Rle3Pre p0 = rle3_preload(blk, k0, true), p1 = rle3_preload(blk, k0 + 1, true);
        rle3_first_tile(sh.gen, blk, k0, p0, tsum, Tarr);
        if ((u32)(k0 + 1) < blk.n_tiles) {
            lds_barrier();
            rle3_first_tile(sh.gen, blk, k0 + 1, p1, tsum, Tarr);
        }
    }
}

// Sizes of the tiles that rle3_first left to the general code: one workgroup looks at 64 tiles' flags (almost none set)
__global__ __launch_bounds__(RLE_THREADS) void rle3_fix(const RleBlk *__restrict__ blks, const u32 *__restrict__ Rarr,
                                                        u32 *__restrict__ Tarr)
{
    __shared__ __attribute__((aligned(16))) RleShared sh;
    __shared__ u64 todo;
    const RleBlk blk = blks[blockIdx.y];
    const u32 k0 = blockIdx.x * 64u;
    if (k0 >= blk.n_tiles) return;
    if (threadIdx.x < 64) {
        const u32 k = k0 + threadIdx.x;
        bool need = false;
        if (k < blk.n_tiles) {
            const u32 t = Tarr[blk.desc_base + k];
            need = (t & T_GENERAL) != 0 || ((t & T_ENTERS) != 0 && Rarr[blk.desc_base + k] >= 60);
        }
        const u64 m = __ballot(need);
        if (threadIdx.x == 0) todo = m;
    }
    __syncthreads();
    for (u64 m = todo; m; m &= m - 1) {                 // uniform
        const int k = (int)k0 + __builtin_ctzll((unsigned long long)m);
        const Rle3Pre pre = rle3_preload(blk, k, true);
        __syncthreads();
        rle3_pass_tile<1>(sh, blk, k, pre, Rarr[blk.desc_base + k], 0ull, Tarr);
    }
}

// ================================================================================================
// rle4_kernel: block_compression (f.c:29-55) and make_freq of its output (f.c:63-79, as f.c:310 calls it) in ONE pass
// over the input: n read + rle_n written, the output counted while it is still in LDS.
//
// A workgroup takes a SUPER-TILE of 32 KiB (four 8 KiB units, the pairs of rle3_emit8k) by a per-block ticket — whoever
// holds a later ticket started after its predecessors, so the two chains below cannot deadlock whatever is resident —
// and requests everything it will ever load at once.  Then, without waiting for anybody:
//   * E / Z masks, run heads and the mask code's emitted sizes of its four units (what rle3_first computed and passed
//     through the workspace);
//   * the run that ENTERS the super-tile, from the 64 bytes in front of it: a shorter run is known exactly, only a run that
//     fills them all needs the STATE chain — one descriptor per super-tile, PREFIX = length of the run that ends at its last
//     byte (published at once by every super-tile that holds a run head: all of them on ordinary data), AGG = "one run from
//     end to end, 32 KiB more" — and such a run sends its unit to the general per-element code anyway;
//   * its size goes into the SIZE chain (decoupled look-back on 64-bit byte offsets, common.hpp), the exclusive prefix is
//     the super-tile's place in the output.
// The units are then emitted one after the other by the code of rle3_emit8k / rle3_copy8k / rle_tile_general into the LDS
// image and leave as aligned 16-byte stores; the image is read once more for the histogram: 16 bank-spread replicas of 256
// counters in LDS, added to the PACKED tile histogram of the output (two 16-bit counts per word: no carry, a 32 KiB tile
// holds at most 32768 of a byte) with at most 128 atomics whenever the output crosses into the next 32 KiB tile and at the
// end.  rle4_freq sums a block's tile histograms into its 256 bins.  The sidecar is what shafa_hipd_rle_encode_tiles leaves
// for the one-shot Shannon-Fano encoder; without a caller's buffer it lives in the workspace.
// ================================================================================================
constexpr int R4_U = 4;                                 // units per workgroup
constexpr u32 R4_ST = (u32)R4_U * (u32)R8_TILE;         // bytes of a super-tile
constexpr int R4_HREP = 16;                             // replicas of the LDS histogram (replica = lane & 15)
constexpr u32 R4_HALO = 64;                             // bytes in front of a super-tile that are looked at for the entering run

struct R4Unit {
    u32 E[RLE_THREADS + 2];        // E masks of the lanes; [0] bits 29..31: the three bytes before the unit, [257] bits 0..2: after
    u32 wsum[4];                   // emitted bytes of every wave (mask code)
    u32 wfirst[4];                 // bytes from the start of a wave to its first run head
    u32 wlast[4];                  // last byte of every wave
    int whead[4];                  // position in the unit of every wave's last run head, -1: none
    u32 pure[4];                   // every byte of the wave is a literal
    u32 slowv;                     // the unit takes the general code whatever enters
    u32 H;                         // bytes after the unit equal to its last byte (<= 255)
};
struct R4Shared {
    R8Shared s;                    // the image of the unit being emitted / the general code's scratch
    R4Unit un[R4_U];
    u32 hist[256 * R4_HREP];       // counter of byte b in replica r: hist[b * R4_HREP + r]
    u32 Tk[2 * R4_U];              // emitted bytes of the 4 KiB tiles of units that take the general code
    u32 ticket;
    u32 halo;                      // bytes in front of the super-tile equal to the byte before it (<= R4_HALO)
    u64 R0;                        // state chain: the run that ends at the last byte before the super-tile
    u64 G;                         // size chain: bytes emitted before the super-tile
};
static_assert(sizeof(R4Shared) <= 40960, "rle4_kernel: four workgroups per CU");

// bytes p .. p + 3 of the block (p a multiple of 4), zeros past its end
__device__ __forceinline__ u32 r4_ldw(const RleBlk &blk, const u64 p)
{
    if (p + 4 <= blk.n) return gload<u32>(blk.in + p);
    u32 v = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (p + j < blk.n) v |= (u32)gload<u8>(blk.in + p + j) << (8 * j);
    return v;
}

// count the m bytes at LDS pointer `at` (any alignment) into the replicated histogram
__device__ __forceinline__ void r4_count_range(u32 *hist, const u8 *at, const u32 m)
{
    if (!m) return;                                     // (uniform)
    const u32 a = lds_addr(at), lo = a & 15u, hi = lo + m;          // valid bytes of the 16-byte pieces: [lo, hi)
    const uint4 *pc = (const uint4 *)(at - lo);
    u32 *h = hist + (threadIdx.x & (R4_HREP - 1));
    const u32 np = (hi + 15u) >> 4;
    for (u32 p = threadIdx.x; p < np; p += RLE_THREADS) {
        const uint4 v = pc[p];
        const u32 wds[4] = {v.x, v.y, v.z, v.w};
        const u32 b0 = 16u * p;
        if (b0 >= lo && b0 + 16u <= hi) {
#pragma unroll
            for (int j = 0; j < 16; ++j) atomicAdd(&h[((wds[j >> 2] >> (8 * (j & 3))) & 0xFFu) * R4_HREP], 1u);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (b0 + (u32)j >= lo && b0 + (u32)j < hi) atomicAdd(&h[((wds[j >> 2] >> (8 * (j & 3))) & 0xFFu) * R4_HREP], 1u);
        }
    }
}

// the counts so far belong to output tile `tile` of the block: add them to its packed histogram, clear the replicas
__device__ __forceinline__ void r4_flush(R4Shared &sh, const RleBlk &blk, const u32 tile)
{
    __syncthreads();
    const u32 tid = threadIdx.x;
    uint4 *row = (uint4 *)(sh.hist + tid * R4_HREP);
    u32 c = 0;
#pragma unroll
    for (u32 j = 0; j < R4_HREP / 4; ++j) {
        const u32 jj = (j + (tid >> 2)) & (R4_HREP / 4 - 1);
        const uint4 v = row[jj];
        c += v.x + v.y + v.z + v.w;
        row[jj] = make_uint4(0, 0, 0, 0);
    }
    const u32 up = (u32)__shfl_down((int)c, 1, 64);
    const u32 packed = c | (up << 16);
    if (!(tid & 1u) && packed) atomicAdd(blk.th32 + (size_t)tile * 128u + (tid >> 1), packed);
    __syncthreads();
}

// the n bytes at LDS pointer `at` are the block's output bytes [G, G + n): count them, tile by tile (n <= 32 KiB)
__device__ __forceinline__ void r4_count(R4Shared &sh, const RleBlk &blk, const u8 *at, const u32 n, const u64 G, u32 &cur_tile)
{
    const u64 tile_end = ((u64)cur_tile + 1) << 15;
    const u32 first = G + n <= tile_end ? n : (u32)(tile_end - G);
    r4_count_range(sh.hist, at, first);
    if (first < n) {                                    // (uniform) the output crosses into the next tile
        r4_flush(sh, blk, cur_tile);
        ++cur_tile;
        r4_count_range(sh.hist, at + first, n - first);
    }
}

template <bool HIST>
__global__ __launch_bounds__(RLE_THREADS, 4) void rle4_kernel(const RleBlk *__restrict__ blks, u32 *__restrict__ tickets,
                                                           u64 *__restrict__ sdesc, u64 *__restrict__ gdesc)
{
    __shared__ __attribute__((aligned(16))) R4Shared sh;
    const RleBlk blk = blks[blockIdx.y];
    if (blockIdx.x >= blk.n_st) return;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    if (tid == 0) sh.ticket = atomicAdd(tickets + blk.ticket, 1u);
    if (HIST)
        for (int i = tid; i < 256 * R4_HREP / 4; i += RLE_THREADS) ((uint4 *)sh.hist)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const u32 g = sh.ticket;                            // the super-tile
    const u64 n = blk.n, st0 = (u64)g * R4_ST;
    const u32 left = n - st0 < (u64)R4_ST ? (u32)(n - st0) : R4_ST;
    const int nun = (int)((left + (u32)R8_TILE - 1u) / (u32)R8_TILE);        // units that exist
    const bool has_next = g + 1 < blk.n_st;
    u64 *const sd = sdesc + blk.st_base, *const gd = gdesc + blk.st_base;
    u8 *const smem = (u8 *)&sh;

    // ---- everything the workgroup ever loads ----------------------------------------------------------------
    u32 w[R4_U][8], qb[R4_U], qa[R4_U], hq[R4_U];
    bool full[R4_U];                                    // the mask code is possible: the unit and four bytes more lie inside the block
#pragma unroll
    for (int u = 0; u < R4_U; ++u) {
        const u64 us = st0 + (u64)u * R8_TILE;
        full[u] = u < nun && us + R8_TILE + 4 <= n;
        qb[u] = qa[u] = hq[u] = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) w[u][i] = 0;
        if (full[u]) {
            const u64 pos = us + (u64)tid * R8_BPL;
            const uint4 v0 = gload_nt<uint4>(blk.in + pos), v1 = gload_nt<uint4>(blk.in + pos + 16);
            w[u][0] = v0.x; w[u][1] = v0.y; w[u][2] = v0.z; w[u][3] = v0.w;
            w[u][4] = v1.x; w[u][5] = v1.y; w[u][6] = v1.z; w[u][7] = v1.w;
            if (tid == RLE_THREADS - 1) qa[u] = gload<u32>(blk.in + us + R8_TILE);
            if (wv == 2) hq[u] = r4_ldw(blk, us + R8_TILE + 4u * (u32)lane);
        } else if (u < nun) {
#pragma unroll
            for (int i = 0; i < 8; ++i) w[u][i] = r4_ldw(blk, us + (u64)tid * R8_BPL + 4u * (u32)i);
        }
        if (u < nun && tid == 0 && us > 0) qb[u] = gload<u32>(blk.in + us - 4);
    }
    u32 hb = 0;                                         // wave 3, lanes 0..15: the 64 bytes in front of the super-tile
    if (wv == 3 && lane < (int)(R4_HALO / 4) && g > 0) hb = gload<u32>(blk.in + st0 - R4_HALO + 4u * (u32)lane);

    // ---- masks: what the lanes need from their neighbours first --------------------------------------------
#pragma unroll
    for (int u = 0; u < R4_U; ++u) {
        if (u >= nun) break;
        if (lane == 63) sh.un[u].wlast[wv] = w[u][7] >> 24;
        if (tid == 0) {
            sh.un[u].E[0] = (st0 + (u64)u * R8_TILE > 0) ? (zmask4(qb[u] ^ (qb[u] << 8)) >> 1) << 29 : 0u;
            sh.un[u].slowv = (blk.force_general || !full[u]) ? 1u : 0u;
        }
        if (tid == RLE_THREADS - 1)
            sh.un[u].E[RLE_THREADS + 1] = full[u] ? zmask4(qa[u] ^ ((qa[u] << 8) | (w[u][7] >> 24))) & 7u : 0u;
    }
    if (wv == 3) {                                      // the run that ends at the byte before the super-tile, inside the halo
        const u32 last = (u32)__shfl((int)hb, (int)(R4_HALO / 4) - 1, 64) >> 24;
        const u32 m = zmask4(hb ^ (last * 0x01010101u));                 // bytes equal to it
        const u32 cnt = (u32)__builtin_clz(~(m << 28));                 // ... counted down from the word's last byte
        const u64 fullm = __ballot(lane < (int)(R4_HALO / 4) && cnt >= 4u) | ~((1ull << (R4_HALO / 4)) - 1ull);
        const u64 nf = ~fullm;                           // lanes 0..15 whose word is not all equal
        const int hl = nf ? 63 - __builtin_clzll((unsigned long long)nf) : -1;
        const u32 ch = hl >= 0 ? (u32)__shfl((int)cnt, hl, 64) : 0u;
        if (lane == 0) sh.halo = g == 0 ? 0u : (hl < 0 ? R4_HALO : 4u * (u32)((int)(R4_HALO / 4) - 1 - hl) + ch);
    }
    lds_barrier();

    // ---- E / Z masks, run heads per wave, the bytes behind every unit that continue its last run ------------------
    u32 E[R4_U], Z[R4_U];
#pragma unroll
    for (int u = 0; u < R4_U; ++u) {
        E[u] = Z[u] = 0;
        if (u >= nun) break;
        const u64 us = st0 + (u64)u * R8_TILE;
        u32 pb = (u32)__shfl_up((int)(w[u][7] >> 24), 1, 64);
        if (lane == 0) pb = wv ? sh.un[u].wlast[wv - 1] : (us > 0 ? qb[u] >> 24 : 0x100u);
        u32 dif[8];                                     // every byte xor the byte before it
        dif[0] = w[u][0] ^ ((w[u][0] << 8) | (pb & 0xFFu));
#pragma unroll
        for (int i = 1; i < 8; ++i) dif[i] = w[u][i] ^ __builtin_amdgcn_alignbit(w[u][i], w[u][i - 1], 24);
        u32 e = zmask32(dif);
        Z[u] = zmask32(w[u]);
        if (pb > 0xFFu) e &= ~1u;
        if (!full[u]) {                                 // bytes past the block's end: heads, so that nothing continues into them
            const u64 p = us + (u64)tid * R8_BPL;
            const u32 nv = p >= n ? 0u : (n - p >= (u64)R8_BPL ? (u32)R8_BPL : (u32)(n - p));
            e &= nv >= 32u ? 0xFFFFFFFFu : ((1u << nv) - 1u);
        }
        E[u] = e;
        sh.un[u].E[tid + 1] = e;
        const u32 H = ~e;
        const u64 hm = __ballot(H != 0);
        const u32 transparent = (u32)__builtin_popcountll((unsigned long long)__ballot(e == 0xFFFFFFFFu));
        const int l1 = hm ? 63 - __builtin_clzll((unsigned long long)hm) : 0, l0 = hm ? __builtin_ctzll((unsigned long long)hm) : 0;
        const u32 hv = (u32)__shfl((int)H, l1, 64), h0 = (u32)__shfl((int)H, l0, 64);
        if (lane == 0) {
            if (transparent > 2) sh.un[u].slowv = 1;   // a run could reach 255 bytes inside the unit
            sh.un[u].whead[wv] = hm ? (wv * 64 + l1) * R8_BPL + (31 - __builtin_clz(hv)) : -1;
            sh.un[u].wfirst[wv] = hm ? 32u * (u32)l0 + (u32)__builtin_ctz(h0) : 0xFFFFFFFFu;
        }
        if (wv == 2) {                                  // halo: bytes after the unit equal to its last byte (<= 255)
            const u32 lastb = sh.un[u].wlast[3];
            const u64 q = us + R8_TILE + 4u * (u32)lane;
            const u32 nv = q >= n ? 0u : (n - q >= 4 ? 4u : (u32)(n - q));
            u32 cnt = 0;
            bool go = true;
#pragma unroll
            for (u32 j = 0; j < 4; ++j) {
                go &= j < nv && ((hq[u] >> (8 * j)) & 0xFFu) == lastb;
                cnt += go ? 1u : 0u;
            }
            const u64 f4 = __ballot(cnt == 4);
            const int f0 = (~f4) ? (__ffsll((unsigned long long)~f4) - 1) : 64;
            const u32 c0 = (f0 < 64) ? (u32)__shfl((int)cnt, f0, 64) : 0u;
            u32 Hh = (u32)f0 * 4 + c0;
            if (Hh > 255) Hh = 255;
            if (lane == 0) sh.un[u].H = full[u] ? Hh : 0u;
        }
    }
    lds_barrier();

    // ---- (uniform) the run that enters every unit; which units take the general code ------------------------------
    int lasthead[R4_U];
    bool enters[R4_U], slow[R4_U];
    u64 Rin[R4_U];
    const u32 halo = sh.halo;
    bool has_head = false;
    u64 trail = 0;                                      // the run that ends at the super-tile's last byte, when it starts inside
#pragma unroll
    for (int u = 0; u < R4_U; ++u) {
        lasthead[u] = -1;
        enters[u] = slow[u] = false;
        Rin[u] = 0;
        if (u >= nun) continue;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) if (sh.un[u].whead[ww] >= 0) lasthead[u] = sh.un[u].whead[ww];
        enters[u] = (sh.un[u].E[1] & 1u) != 0;
        if (lasthead[u] >= 0) { has_head = true; trail = (u64)(nun - 1 - u) * R8_TILE + (u64)(R8_TILE - lasthead[u]); }
    }
    const bool need_chain = enters[0] && halo >= R4_HALO;      // the entering run fills the halo: its length is the chain's
    auto enter_runs = [&](const u64 r0) {
        Rin[0] = enters[0] ? r0 : 0;
#pragma unroll
        for (int u = 1; u < R4_U; ++u) {
            if (u >= nun) break;
            const u64 tp = lasthead[u - 1] >= 0 ? (u64)(R8_TILE - lasthead[u - 1]) : Rin[u - 1] + R8_TILE;
            Rin[u] = enters[u] ? tp : 0;
        }
    };
    enter_runs(halo);
#pragma unroll
    for (int u = 0; u < R4_U; ++u)
        if (u < nun) slow[u] = sh.un[u].slowv != 0 || (enters[u] && Rin[u] >= 60);
    if (tid == 0 && has_next) {                         // state chain: published before anything is waited for
        if (has_head) desc_store(sd + g, DESC_PREFIX, trail);
        else if (!need_chain) desc_store(sd + g, DESC_PREFIX, Rin[0] + R4_ST);
        else desc_store(sd + g, DESC_AGG, R4_ST);
    }

    // ---- mask code: literals, triple heads, emitted bytes ------------------------------------------------------------
    u32 Lit[R4_U], T3[R4_U], tot[R4_U], incl[R4_U];
#pragma unroll
    for (int u = 0; u < R4_U; ++u) {
        Lit[u] = T3[u] = tot[u] = incl[u] = 0;
        if (u >= nun || slow[u]) continue;              // (uniform)
        const u64 B = (u64)(sh.un[u].E[tid] >> 29) | ((u64)E[u] << 3) | ((u64)(sh.un[u].E[tid + 2] & 7u) << 35);   // positions -3 .. 34
        const u64 T = B & (B >> 1) & (B >> 2);
        const u32 LC = (u32)(((T >> 1) | T | (T << 1) | (T << 2)) >> 3);    // bytes of runs of >= 4
        Lit[u] = ~Z[u] & ~LC;
        T3[u] = ~E[u] & (Z[u] | LC);
        tot[u] = (u32)__builtin_popcount(Lit[u]) + 3u * (u32)__builtin_popcount(T3[u]);
        incl[u] = wave_incl_scan_add<u32>(tot[u]);
        const bool wpure = __ballot(Lit[u] != 0xFFFFFFFFu) == 0ull;
        if (lane == 63) sh.un[u].wsum[wv] = incl[u];
        if (lane == 0) sh.un[u].pure[wv] = wpure ? 1u : 0u;
    }
    if (need_chain) {                                   // (uniform, rare) a long run enters: its exact length
        if (wv == 0) {
            const u64 R = lookback_sum(sd, (int)g, blk.err);
            if (lane == 0) {
                sh.R0 = R;
                if (!has_head && has_next) desc_store(sd + g, DESC_PREFIX, R + R4_ST);
            }
        }
        lds_barrier();
        enter_runs(sh.R0);
    }
    // run that ends at the last byte of a unit's first 4 KiB tile (the general code works on those)
    auto second_tile_run = [&](const int u) -> u64 {
        const int lh = sh.un[u].whead[1] >= 0 ? sh.un[u].whead[1] : sh.un[u].whead[0];
        return lh >= 0 ? (u64)(RLE_TILE - lh) : Rin[u] + RLE_TILE;
    };
#pragma unroll
    for (int u = 0; u < R4_U; ++u) {
        if (u >= nun || !slow[u]) continue;             // (uniform)
        for (int t = 0; t < 2; ++t) {
            const u32 k = 2u * (g * (u32)R4_U + (u32)u) + (u32)t;
            __syncthreads();
            if (k < blk.n_tiles) rle_tile_general<1>(sh.s.gen, blk, (int)k, t ? second_tile_run(u) : Rin[u], 0ull, &sh.Tk[2 * u + t]);
            else if (tid == 0) sh.Tk[2 * u + t] = 0;
        }
    }
    __syncthreads();
    u32 Tu[R4_U], Ttot = 0;
#pragma unroll
    for (int u = 0; u < R4_U; ++u) {
        Tu[u] = 0;
        if (u >= nun) continue;
        Tu[u] = slow[u] ? sh.Tk[2 * u] + sh.Tk[2 * u + 1] : sh.un[u].wsum[0] + sh.un[u].wsum[1] + sh.un[u].wsum[2] + sh.un[u].wsum[3];
        Ttot += Tu[u];
    }

    // ---- size chain -------------------------------------------------------------------------------------------------------
    if (wv == 0) {
        u64 G = 0;
        if (g == 0) {
            if (lane == 0) desc_store(gd, DESC_PREFIX, Ttot);
        } else {
            if (lane == 0) desc_store(gd + g, DESC_AGG, Ttot);
            G = lookback_sum(gd, (int)g, blk.err);
            if (lane == 0) desc_store(gd + g, DESC_PREFIX, G + Ttot);
        }
        if (lane == 0) {
            sh.G = G;
            if (!has_next) gstore<u64>(blk.out_n, G + Ttot);
        }
    }
    lds_barrier();
    u64 Gu = sh.G;
    if (Gu + Ttot > blk.out_cap) {                      // (uniform) the block does not fit: nothing of this super-tile is stored
        if (tid == 0) set_error(blk.err, SHAFA_LACK_OF_MEMORY);
        return;
    }
    u32 cur_tile = (u32)(Gu >> 15);

    // ---- the units leave, one after the other ---------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < R4_U; ++u) {
        if (u >= nun) break;
        lds_barrier();                                  // the image of the unit before is done with
        if (slow[u]) {                                  // general per-element code, 4 KiB tile by tile
            u64 Gk = Gu;
            for (int t = 0; t < 2; ++t) {
                const u32 k = 2u * (g * (u32)R4_U + (u32)u) + (u32)t;
                if (k >= blk.n_tiles) break;
                __syncthreads();
                rle_tile_general<2>(sh.s.gen, blk, (int)k, t ? second_tile_run(u) : Rin[u], Gk, nullptr);
                if (HIST) r4_count(sh, blk, (const u8 *)sh.s.gen.stage + ((u32)Gk & 3u), sh.Tk[2 * u + t], Gk, cur_tile);
                Gk += sh.Tk[2 * u + t];
            }
            Gu += Tu[u];
            continue;
        }
        const u32 Tt = Tu[u];
        if (sh.un[u].pure[0] & sh.un[u].pure[1] & sh.un[u].pure[2] & sh.un[u].pure[3]) {
            // every byte is a literal: the unit leaves as it came (rle3_copy8k)
            constexpr u32 base = (u32)offsetof(R8Fast, img) + 16u;         // 16 bytes in front: a piece may start before the unit
            *(uint4 *)(smem + base + 32u * (u32)tid) = make_uint4(w[u][0], w[u][1], w[u][2], w[u][3]);
            *(uint4 *)(smem + base + 32u * (u32)tid + 16u) = make_uint4(w[u][4], w[u][5], w[u][6], w[u][7]);
            lds_barrier();
            u8 *gout = blk.out + Gu;
            const u32 mis = (u32)((uintptr_t)gout & 15u);
            for (u32 q = (u32)tid; 16 * q < mis + (u32)R8_TILE; q += RLE_THREADS) {
                const u32 s0 = base + 16 * q - mis, sb = s0 & ~3u, sf = s0 & 3u;
                const u32 d0 = *(const u32 *)__builtin_assume_aligned(smem + sb, 4), d1 = *(const u32 *)__builtin_assume_aligned(smem + sb + 4, 4),
                          d2 = *(const u32 *)__builtin_assume_aligned(smem + sb + 8, 4), d3 = *(const u32 *)__builtin_assume_aligned(smem + sb + 12, 4),
                          d4 = *(const u32 *)__builtin_assume_aligned(smem + sb + 16, 4);
                const u32 wds[4] = {__builtin_amdgcn_alignbyte(d1, d0, sf), __builtin_amdgcn_alignbyte(d2, d1, sf),
                                    __builtin_amdgcn_alignbyte(d3, d2, sf), __builtin_amdgcn_alignbyte(d4, d3, sf)};
                u8 *ga = gout - mis + 16 * q;
                const u32 lo = 16 * q;                  // the piece is bytes [lo - mis, lo - mis + 16) of the unit
                if (lo >= mis && lo + 16 <= mis + (u32)R8_TILE) {
                    gstore_nt<uint4>(ga, make_uint4(wds[0], wds[1], wds[2], wds[3]));
                } else {
#pragma unroll
                    for (u32 b = 0; b < 16; ++b)
                        if (lo + b >= mis && lo + b < mis + (u32)R8_TILE) ga[b] = (u8)(wds[b >> 2] >> (8 * (b & 3)));
                }
            }
            if (HIST) r4_count(sh, blk, smem + base, (u32)R8_TILE, Gu, cur_tile);
            Gu += Tt;
            continue;
        }
        // the mask code (rle3_emit8k): tokens into the image, which is aligned like the output address
        const u32 H = ~E[u];
        const u64 hm = __ballot(H != 0);                // never 0: at most two lanes of a wave have no head
        u32 after;                                      // bytes from the end of this lane to the next run head
        {
            const u64 above = lane == 63 ? 0ull : hm & ~((2ull << lane) - 1);
            const int l2 = above ? __builtin_ctzll((unsigned long long)above) : 0;
            const u32 hv = (u32)__shfl((int)H, l2, 64);
            after = above ? 32u * (u32)(l2 - lane - 1) + (u32)__builtin_ctz(hv)
                          : 32u * (u32)(63 - lane) + (wv < 3 ? sh.un[u].wfirst[wv + 1] : sh.un[u].H);
        }
        u32 off = incl[u] - tot[u];
#pragma unroll
        for (int q = 0; q < 3; ++q) if (q < wv) off += sh.un[u].wsum[q];
        const u32 shift = (u32)((uintptr_t)(blk.out + Gu) & 15u);
        const u32 p0 = (u32)offsetof(R8Fast, img) + shift + off;
        {   // every byte is written somewhere, no branches (see rle3_emit8k)
            u32 p = p0;
            const u32 p_end = p0 + tot[u];
            const u32 dump = (u32)offsetof(R8Fast, dump) + 4u * (u32)tid;
#pragma unroll
            for (int q = 0; q < R8_BPL / 4; ++q) {
                const u32 ft = nibf(T3[u], q), sz = nibf(Lit[u], q) | ft | (ft << 1);
                u32 a;
                a = add_byte<0>(p, ft); smem[a < p_end ? a : dump] = (u8)w[u][q];         p = add_byte<0>(p, sz);
                a = add_byte<1>(p, ft); smem[a < p_end ? a : dump] = (u8)(w[u][q] >> 8);  p = add_byte<1>(p, sz);
                a = add_byte<2>(p, ft); smem[a < p_end ? a : dump] = (u8)(w[u][q] >> 16); p = add_byte<2>(p, sz);
                a = add_byte<3>(p, ft); smem[a < p_end ? a : dump] = (u8)(w[u][q] >> 24); p = add_byte<3>(p, sz);
            }
        }
        u32 p3 = p0;
        for (u32 t = T3[u]; t; t &= t - 1, p3 += 3u) {  // triples: the lane's heads in order
            const u32 j = (u32)__builtin_ctz(t), below = (1u << j) - 1u;
            const u32 hn = j < 31 ? H >> (j + 1) : 0u;
            u32 L = hn ? (u32)__builtin_ctz(hn) + 1u : (32u - j) + after;
            L = L > 255u ? 255u : L;
            const u32 at = p3 + (u32)__builtin_popcount(Lit[u] & below);
            smem[at] = 0;
            smem[at + 2] = (u8)L;
        }
        lds_barrier();
        {
            u8 *gbase = blk.out + Gu - shift;           // aligned 16-byte pieces, non-temporal
            const u32 npieces = (shift + Tt + 15) >> 4;
            for (u32 q = (u32)tid; q < npieces; q += RLE_THREADS) {
                const uint4 v = ((const uint4 *)sh.s.f.img)[q];
                const u32 lo = 16 * q;
                if (lo >= shift && lo + 16 <= shift + Tt) {
                    gstore_nt<uint4>(gbase + lo, v);
                } else {
                    const u32 wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (u32 b = 0; b < 16; ++b)
                        if (lo + b >= shift && lo + b < shift + Tt) gbase[lo + b] = (u8)(wds[b >> 2] >> (8 * (b & 3)));
                }
            }
        }
        if (HIST) r4_count(sh, blk, sh.s.f.img + shift, Tt, Gu, cur_tile);
        Gu += Tt;
    }
    if (HIST) r4_flush(sh, blk, cur_tile);
}

// before rle4_kernel: tickets and descriptors, the packed tile histograms, the blocks' bins and sizes
__global__ __launch_bounds__(256) void rle4_zero(const RleBlk *__restrict__ blks, u32 *__restrict__ tickets, u64 *__restrict__ sdesc,
                                                 u64 *__restrict__ gdesc)
{
    const RleBlk blk = blks[blockIdx.y];
    const u32 t = blockIdx.x * 256u + threadIdx.x, step = gridDim.x * 256u;
    if (t == 0) {
        tickets[blk.ticket] = 0;
        gstore<u64>(blk.out_n, 0);
    }
    if (blk.freq && t < 256) blk.freq[t] = 0;
    for (u32 i = t; i < blk.n_st; i += step) { sdesc[blk.st_base + i] = 0; gdesc[blk.st_base + i] = 0; }
    if (blk.th32)
        for (u64 i = t; i < blk.th_bytes / 16; i += step) ((uint4 *)blk.th32)[i] = make_uint4(0, 0, 0, 0);
}

// a block's bins = the sum of its tile histograms
__global__ __launch_bounds__(256) void rle4_freq(const RleBlk *__restrict__ blks)
{
    const RleBlk blk = blks[blockIdx.y];
    if (!blk.freq || !blk.th32) return;
    const u64 out_n = gload<u64>(blk.out_n);
    const u32 nt = (u32)((out_n + 32767) >> 15);
    const u16 *th = (const u16 *)blk.th32;
    u64 c = 0;
    for (u32 t = blockIdx.x; t < nt; t += gridDim.x) c += th[(size_t)t * 256 + threadIdx.x];
    if (c) atomicAdd((unsigned long long *)(blk.freq + threadIdx.x), (unsigned long long)c);
}

}  // namespace

static int g_rle_force_general = 0;
// "rle_encode_one_pass": 1 = rle4_kernel, 0 (default) = the two-pass kernels (rle3_*) and a histogram pass.  Measured on
// 32 x 64 MiB of Zipf symbols in geometric runs (profiles/r6_rle_one_pass.txt): one pass moves 4.10 GB where the two-pass
// sequence moves 8.87 GB (1.10 x against 2.39 x the algorithmic bytes) and takes 2.41 ms against 1.82 ms: both forms execute
// the same 8.2-8.5 G vector instructions per launch (25 per input byte: masks, classification, the byte loop, the counts),
// at four cycles each on 1024 SIMDs that is 1.55 ms, and the chained form keeps the vector ALUs 65 % busy (sixteen waves per
// CU: 39.5 KB of LDS per workgroup, ticket, look-back and a dozen barriers per super-tile) where the independent
// workgroups of the two-pass kernels keep them 84 % busy.  The family is bound by its instruction count, not by its traffic.
static int g_rle_one_pass = 0;
void rleenc_configure(int force_general) { g_rle_force_general = force_general; }
void rleenc_configure_one_pass(int on) { g_rle_one_pass = on; }

// One pass over the input: rle4_zero -> rle4_kernel -> rle4_freq (d_freq wanted).  The packed tile histograms go to the
// caller's sidecar (shafa_hipd_rle_encode_tiles) or, when only the bins are wanted, to the workspace.
static int rleenc_launch_one_pass(Batch *bt, hipStream_t st, int nblocks, const u8 *d_in, const u64 *h_in_off,
                                  const u64 *h_in_n, u8 *d_out, const u64 *h_out_off, const u64 *h_out_cap, u64 *d_out_n,
                                  u64 *d_freq, u8 *d_thist, const u64 *h_thist_off)
{
    if ((d_thist == nullptr) != (h_thist_off == nullptr)) return SHAFA_OUTSIDE_MODULE;
    if (d_thist && !d_freq) return SHAFA_OUTSIDE_MODULE;
    u64 nst = 0, ws_th = 0;
    u32 max_st = 0;
    std::vector<u64> thb(nblocks, 0);
    for (int b = 0; b < nblocks; ++b) {
        if ((h_in_off[b] & 15) || (h_out_off[b] & 15) || (d_thist && (h_thist_off[b] & 15))) return SHAFA_OUTSIDE_MODULE;
        const u64 t = ceil_div_u64(h_in_n[b], R4_ST);
        if (t > 0x7FFFFFFFull) return SHAFA_OUTSIDE_MODULE;
        nst += t;
        if (t > max_st) max_st = (u32)t;
        if (d_freq) {                                   // tiles the output can reach: 2 n + 3 bytes at most, and never past out_cap
            const u64 worst = 2 * h_in_n[b] + 3, reach = h_out_cap[b] < worst ? h_out_cap[b] : worst;
            thb[b] = h_in_n[b] ? ceil_div_u64(reach, 32768) * 512 : 0;
            if (!d_thist) ws_th += thb[b];
        }
    }
    // workspace: [RleBlk] [tickets] [state descriptors] [size descriptors] [packed tile histograms, unless the caller's]
    size_t off = 0;
    const size_t o_blk = off; off += (size_t)nblocks * sizeof(RleBlk); off = (off + 15) & ~(size_t)15;
    const size_t o_tick = off; off += (size_t)nblocks * 4; off = (off + 15) & ~(size_t)15;
    const size_t o_sd = off; off += nst * 8; off = (off + 15) & ~(size_t)15;
    const size_t o_gd = off; off += nst * 8; off = (off + 15) & ~(size_t)15;
    const size_t o_th = off; off += ws_th;
    int rc = batch_reserve(bt, st, off);
    if (rc) return rc;
    u8 *ws = (u8 *)bt->d_ws;
    RleBlk *hb = (RleBlk *)batch_stage(bt, st, (size_t)nblocks * sizeof(RleBlk));
    if (!hb) return SHAFA_LACK_OF_MEMORY;
    u32 sbase = 0;
    u64 thpos = 0, max_th = 0;
    for (int b = 0; b < nblocks; ++b) {
        RleBlk &e = hb[b];
        memset(&e, 0, sizeof(e));
        e.in = d_in + h_in_off[b];
        e.out = d_out + h_out_off[b];
        e.n = h_in_n[b];
        e.out_cap = h_out_cap[b];
        e.out_n = d_out_n + b;
        e.err = bt->d_err + b;
        e.n_tiles = (u32)ceil_div_u64(h_in_n[b], RLE_TILE);
        e.ticket = (u32)b;
        e.force_general = g_rle_force_general ? 1u : 0u;
        e.st_base = sbase;
        e.n_st = (u32)ceil_div_u64(h_in_n[b], R4_ST);
        sbase += e.n_st;
        e.freq = d_freq ? d_freq + (size_t)b * 256 : nullptr;
        e.th_bytes = thb[b];
        if (d_freq) {
            if (d_thist) e.th32 = (u32 *)(d_thist + h_thist_off[b]);
            else { e.th32 = (u32 *)(ws + o_th + thpos); thpos += thb[b]; }
        }
        if (thb[b] > max_th) max_th = thb[b];
    }
    if ((rc = batch_upload(bt, st, ws + o_blk, hb, (size_t)nblocks * sizeof(RleBlk)))) return rc;
    const RleBlk *dblk = (const RleBlk *)(ws + o_blk);
    u32 *tick = (u32 *)(ws + o_tick);
    u64 *sd = (u64 *)(ws + o_sd), *gd = (u64 *)(ws + o_gd);
    {   // enough workgroups per block to zero its tile histograms at memory speed, at least one (tickets, sizes, bins)
        u64 zx = ceil_div_u64(max_th / 16 > max_st ? max_th / 16 : max_st, 256 * 8);
        if (zx < 1) zx = 1;
        if (zx > 64) zx = 64;
        hipLaunchKernelGGL(rle4_zero, dim3((u32)zx, (u32)nblocks), dim3(256), 0, st, dblk, tick, sd, gd);
    }
    if (max_st) {
        if (d_freq) hipLaunchKernelGGL(rle4_kernel<true>, dim3(max_st, (u32)nblocks), dim3(RLE_THREADS), 0, st, dblk, tick, sd, gd);
        else hipLaunchKernelGGL(rle4_kernel<false>, dim3(max_st, (u32)nblocks), dim3(RLE_THREADS), 0, st, dblk, tick, sd, gd);
        if (d_freq) hipLaunchKernelGGL(rle4_freq, dim3(16, (u32)nblocks), dim3(256), 0, st, dblk);
    }
    HIP_TRY(hipGetLastError());
    return SHAFA_SUCCESS;
}

int rleenc_launch(Batch *bt, hipStream_t st, int nblocks, const u8 *d_in, const u64 *h_in_off,
                  const u64 *h_in_n, u8 *d_out, const u64 *h_out_off, const u64 *h_out_cap, u64 *d_out_n,
                  u64 *d_freq, u8 *d_thist, const u64 *h_thist_off)
{
    if (nblocks <= 0) return SHAFA_SUCCESS;
    if (nblocks > bt->max_blocks) return SHAFA_LACK_OF_MEMORY;
    if (g_rle_one_pass)
        return rleenc_launch_one_pass(bt, st, nblocks, d_in, h_in_off, h_in_n, d_out, h_out_off, h_out_cap, d_out_n, d_freq, d_thist,
                                      h_thist_off);
    u64 ndesc = 0;
    u32 max_tiles = 0;
    for (int b = 0; b < nblocks; ++b) {
        if ((h_in_off[b] & 15) || (h_out_off[b] & 15)) return SHAFA_OUTSIDE_MODULE;
        const u64 t = ceil_div_u64(h_in_n[b], RLE_TILE);
        ndesc += t;
        if (t > max_tiles) max_tiles = (u32)t;
    }
    // workspace: [G: output offset per tile] [RleBlk] [tsum] [R] [T] [masks]
    size_t off = 0;
    const size_t o_sum = off; off += ndesc * 8; off = (off + 15) & ~(size_t)15;
    const size_t o_blk = off; off += (size_t)nblocks * sizeof(RleBlk); off = (off + 15) & ~(size_t)15;
    const size_t o_tsum = off; off += ndesc * 4; off = (off + 15) & ~(size_t)15;
    const size_t o_R = off; off += ndesc * 4; off = (off + 15) & ~(size_t)15;
    const size_t o_T = off; off += ndesc * 4; off = (off + 15) & ~(size_t)15;
    const size_t o_M = off; off += ndesc * (RLE_TILE / 32) * sizeof(uint2);      // {E, Z} per 32 bytes: n / 4
    int rc = batch_reserve(bt, st, off);
    if (rc) return rc;
    u8 *ws = (u8 *)bt->d_ws;
    RleBlk *hb = (RleBlk *)batch_stage(bt, st, (size_t)nblocks * sizeof(RleBlk));
    if (!hb) return SHAFA_LACK_OF_MEMORY;
    u32 dbase = 0;
    for (int b = 0; b < nblocks; ++b) {
        RleBlk &e = hb[b];
        e.in = d_in + h_in_off[b];
        e.out = d_out + h_out_off[b];
        e.n = h_in_n[b];
        e.out_cap = h_out_cap[b];
        e.out_n = d_out_n + b;
        e.err = bt->d_err + b;
        e.desc_base = dbase;
        e.n_tiles = (u32)ceil_div_u64(h_in_n[b], RLE_TILE);
        e.ticket = (u32)b;
        e.force_general = g_rle_force_general ? 1u : 0u;
        e.masks = (uint2 *)(ws + o_M) + (size_t)dbase * (RLE_TILE / 32);
        dbase += e.n_tiles;
    }
    HIP_TRY(hipMemsetAsync(d_out_n, 0, (size_t)nblocks * 8, st));      // empty blocks: size 0
    if ((rc = batch_upload(bt, st, ws + o_blk, hb, (size_t)nblocks * sizeof(RleBlk)))) return rc;
    if (max_tiles) {
        const RleBlk *dblk = (const RleBlk *)(ws + o_blk);
        const dim3 grid_t((max_tiles + 1) / 2, (u32)nblocks), grid_b((u32)nblocks);     // two tiles per workgroup
        u32 *tsum = (u32 *)(ws + o_tsum), *Rr = (u32 *)(ws + o_R), *Tt = (u32 *)(ws + o_T);
        u64 *Gg = (u64 *)(ws + o_sum);
        hipLaunchKernelGGL(rle3_first, dim3((max_tiles + 3) / 4, (u32)nblocks), dim3(RLE_THREADS), 0, st, dblk, tsum, Tt);
        hipLaunchKernelGGL(rle3_carry, grid_b, dim3(SCAN_THREADS), 0, st, dblk, (const u32 *)tsum, Rr);
        hipLaunchKernelGGL(rle3_fix, dim3((max_tiles + 63) / 64, (u32)nblocks), dim3(RLE_THREADS), 0, st, dblk, (const u32 *)Rr, Tt);
        hipLaunchKernelGGL(rle3_offsets, grid_b, dim3(SCAN_THREADS), 0, st, dblk, (const u32 *)Tt, Gg);
        hipLaunchKernelGGL(rle3_emit, grid_t, dim3(RLE_THREADS), 0, st, dblk, (const u32 *)Rr, Tt, (const u64 *)Gg);
        HIP_TRY(hipGetLastError());
    }
    if (d_freq) {   // make_freq of the RLE bytes (f.c:310): sizes are on the device
        rc = hist_launch_dev(bt, st, nblocks, d_out, h_out_off, h_out_cap, d_out_n, d_freq, d_thist, h_thist_off);
        if (rc) return rc;
    }
    return SHAFA_SUCCESS;
}
