// rle_decode.hip — Module D hot path: RLE expansion (reference d.c:116-197 rle_block_decompressor).
//
// A 0 byte may be an escape, a symbol or a count, so token boundaries are found with a 3-state machine
//     S0 (token start): b == 0 -> S1, else literal -> S0      S1 (symbol) -> S2      S2 (count) -> S0
// which only looks at "is the byte zero".  A lane takes 32 bytes: their zero mask indexes a 256-entry table (8 bytes
// at a time: exit state and token-start mask for each of the three entry states), the lane's map {S0,S1,S2}->{S0,S1,S2}
// (6 bits) is composed with an ordered scan lanes -> waves.  The TILE's entry state needs nobody: two non-zero bytes in a
// row end in S0 whatever the state before them, so the map of the 32 bytes in front of the tile is almost always constant
// and then is the entry state; only a tile behind 32 zero-heavy bytes looks back over the tile maps (a map that sends
// every state to the same state ends the walk, which almost every tile's does).  With its entry state a lane has its
// token-start mask, hence its literal mask L and escape mask E as bits, and SWAR gives the per-byte output lengths
// (1 at a literal, the count byte two places on at an escape — 1 when that is 0, d.c:179-184 — else 0): their sum is
// scanned for output offsets (second look-back, 64-bit).  The tile's output is built in an LDS image
// — literals as byte writes at running positions (one add per byte), runs as word writes with byte writes at both
// ends — while wave 0 is still looking back for the tile's output offset, and leaves as aligned 16-byte stores (five
// dwords of the image funnel-shifted by the offset's misalignment).  A tile whose output exceeds the image goes in
// rounds of consecutive lanes.  A tile without a zero byte that is entered at a token start is stored straight from its
// input.
//
// What bounds it (LABNOTES.md 3.4, profiles/r4_rld_*): a tile's 10-13 us are memory round trips in a row — ticket, load,
// offset look-back, stores — at seven workgroups per CU (72 VGPRs, 12 KiB image).  The tile descriptors sit 32 bytes apart:
// agent-scope atomics are served by the memory side, and sixteen descriptors in one cache line made every in-flight tile
// of a block queue on the same three lines.  Neither fewer barriers, nor persistent workgroups with prefetch, nor a
// speculative load ahead of the ticket paid (tools/experiments/).
//
// Algorithmic HBM bytes per block: rle_n read + orig_n written.
#include "common.hpp"
#include "internal.hpp"

namespace {

constexpr int RLD_THREADS = 256;
constexpr int RLD_BPL = 32;                        // bytes per lane
constexpr int RLD_TILE = RLD_THREADS * RLD_BPL;
#define RLD_IMG_KB 12
constexpr int RLD_IMG = RLD_IMG_KB * 1024;         // bytes of the output image
constexpr u32 FN_IDENT = 0u | (1u << 2) | (2u << 4);
#define RLD_DSTRIDE 4                              // u64 words between the descriptors of consecutive tiles
#define RLD_SLEEP 1
// u32 words between the ticket counters of consecutive blocks (256 bytes).  Returning agent-scope atomics are served by the
// memory side one after the other per cache line, 13-20 ns each, whichever words of the line they hit; the tile needs its ticket
// before it can load anything, and with every block's counter in one line a launch of 8 blocks took its 46 K tickets one by one:
// 614 us where the tiles need 450 (2 / 4 / 8 blocks: - 26 .. 28 %; 32 blocks and more: the same; one block: one counter, 20 ns
// a tile = 115 us per 64 MiB, the kernel's time — the tile index from the grid position would be 77 us, but nothing orders
// the dispatch of workgroups on this hardware, and a tile that waits for a predecessor that has not started never ends).
// The chained encoders ask for tickets three tiles ahead and do not notice (A/B: tools/experiments/README.md).
#define RLD_TSTRIDE 64


// per 8-bit zero mask (bit i = byte i is 0) and entry state s: bits [10 s, 10 s + 8) = token starts, [10 s + 8, 10 s + 10) = exit
struct RldFsm {
    u32 v[256];
    constexpr RldFsm() : v()
    {
        for (u32 z = 0; z < 256; ++z) {
            u32 e = 0;
            for (u32 s0 = 0; s0 < 3; ++s0) {
                u32 st = s0, starts = 0;
                for (u32 i = 0; i < 8; ++i) {
                    if (st == 0) { starts |= 1u << i; st = ((z >> i) & 1u) ? 1u : 0u; }
                    else st = st == 1 ? 2u : 0u;
                }
                e |= (starts | (st << 8)) << (10 * s0);
            }
            v[z] = e;
        }
    }
};
__device__ const RldFsm g_rld_fsm = RldFsm();

struct RldBlk {
    const u8 *in;
    u8 *out;
    u64 n;
    u64 out_cap;
    u64 *out_n;
    int *err;
    u32 desc_base;
    u32 n_tiles;
    u32 ticket;
    u32 pad;
};

// apply a first, then b
__device__ __forceinline__ u32 fn_compose(u32 a, u32 b)
{
    const u32 r0 = (b >> (2 * (a & 3))) & 3;
    const u32 r1 = (b >> (2 * ((a >> 2) & 3))) & 3;
    const u32 r2 = (b >> (2 * ((a >> 4) & 3))) & 3;
    return r0 | (r1 << 2) | (r2 << 4);
}
__device__ __forceinline__ u32 fn_apply(u32 f, u32 s) { return (f >> (2 * s)) & 3; }
__device__ __forceinline__ bool fn_const(u32 f) { return (f & 3) == ((f >> 2) & 3) && (f & 3) == ((f >> 4) & 3); }
__device__ __forceinline__ u32 step(u32 s, u32 b) { return s == 0 ? (b == 0 ? 1u : 0u) : (s == 1 ? 2u : 0u); }

// state entering tile k (= state after tile k-1); wave 0, all lanes
__device__ __forceinline__ u32 lookback_state(const u64 *desc, int k, int *err)
{
    const int lane = lane_id();
    u32 acc = FN_IDENT;       // composition of the tiles between the current window and tile k
    int j = k - 1;
    for (;;) {
        const int idx = j - lane;
        u64 d = 0;
        SpinClock spin;
        for (;;) {
            d = (idx >= 0) ? desc_load(desc + (size_t)idx * RLD_DSTRIDE) : (DESC_PREFIX << 62);
            {   // entries behind the nearest tile that settles the state are not needed: do not wait for them
                const bool empty = (d >> 62) == DESC_EMPTY;
                const u64 sm = __ballot(!empty && ((d >> 62) == DESC_PREFIX || fn_const((u32)d & 63u))), em = __ballot(empty);
                const u64 need = sm ? ((sm & (0 - sm)) - 1) : ~0ull;
                if (!(em & need)) break;
            }
            if (spin.expired()) {
                if (lane == 0) set_error(err, SHAFA_DEVICE_ERROR);
                if ((d >> 62) == DESC_EMPTY) d = (DESC_PREFIX << 62);
                break;
            }
            __builtin_amdgcn_s_sleep(RLD_SLEEP);
        }
        const u32 val = (u32)d & 63u;
        const bool isP = (d >> 62) == DESC_PREFIX;
        const bool stop = isP || fn_const(val);
        const u64 m = __ballot(stop);
        const int pl = m ? (__ffsll((unsigned long long)m) - 1) : 64;
        if (m) {
            u32 s = (u32)__shfl((int)val, pl, 64) & 3u;       // PREFIX: the state; const map: its value
            for (int l = pl - 1; l >= 0; --l) s = fn_apply((u32)__shfl((int)val, l, 64), s);
            return fn_apply(acc, s);
        }
        u32 wfn = FN_IDENT;
        for (int l = 63; l >= 0; --l) wfn = fn_compose(wfn, (u32)__shfl((int)val, l, 64));
        acc = fn_compose(wfn, acc);
        j -= 64;
    }
}

struct RldShared {
    u8 pad0[16];                   // in front of `in`: a tile without zero bytes is stored straight from `in` (a piece may start before it)
    u8 in[RLD_TILE + 32];          // the tile, two look-ahead bytes, zeros
    u8 img[RLD_IMG + 48];          // 16 bytes in front (a piece may start before the image), slack for the last piece
    union {
        u32 fsm[256];              // token-start table: used until the lane masks exist,
        u32 dump[RLD_THREADS];     // then the words that swallow the byte writes of non-literal bytes
    };
    u32 wfn[4];
    u32 wlen[4];
    u32 wplain[4];                 // per wave: 32 valid bytes in every lane and not one zero among them
    u32 tile;
    u32 state_in;
    u32 next;
    u64 O;
};

// zero mask of 32 bytes -> token-start mask and exit state for each of the three entry states
__device__ __forceinline__ void fsm32(const u32 *fsm, u32 z, u32 (&st3)[3], u32 (&ex3)[3])
{
    const u32 e0 = fsm[z & 255u], e1 = fsm[(z >> 8) & 255u], e2 = fsm[(z >> 16) & 255u], e3 = fsm[z >> 24];
#pragma unroll
    for (int s0 = 0; s0 < 3; ++s0) {
        u32 x = (e0 >> (10 * s0)) & 1023u, starts = x & 255u;
        x = (e1 >> (10 * (x >> 8))) & 1023u; starts |= (x & 255u) << 8;
        x = (e2 >> (10 * (x >> 8))) & 1023u; starts |= (x & 255u) << 16;
        x = (e3 >> (10 * (x >> 8))) & 1023u; starts |= (x & 255u) << 24;
        st3[s0] = starts;
        ex3[s0] = x >> 8;
    }
}

// bit i of a nibble -> 0x01 in byte i
__device__ __forceinline__ u32 nib_flags(u32 mask, int i) { return __umul24((mask >> (4 * i)) & 15u, 0x00204081u) & 0x01010101u; }

// `c` copies of `sym` at image byte p (c >= 1): bytes up to the word boundary, whole words, the bytes left
__device__ __forceinline__ void rld_fill(u8 *smem, u32 p, u32 sym, u32 c)
{
    const u32 pat = sym * 0x01010101u;
    const u32 to_b = (4u - (p & 3u)) & 3u, h = to_b < c ? to_b : c;
#pragma unroll
    for (u32 q = 0; q < 3; ++q)
        if (q < h) smem[p + q] = (u8)sym;
    p += h;
    c -= h;
    const u32 e = p + (c & ~3u);
    for (; p < e; p += 4) *(u32 *)__builtin_assume_aligned(smem + p, 4) = pat;
#pragma unroll
    for (u32 q = 0; q < 3; ++q)
        if (q < (c & 3u)) smem[p + q] = (u8)sym;
}

#define RLD_WAVES 7
__global__ __launch_bounds__(RLD_THREADS) __attribute__((amdgpu_waves_per_eu(RLD_WAVES, RLD_WAVES))) void rle_decode_kernel(const RldBlk *__restrict__ blks, int nblk,
                                                                 u64 *desc_state, u64 *desc_sum, u32 *tickets)
{
    __shared__ __attribute__((aligned(16))) RldShared sh;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int b = blockIdx.x % nblk;
    const RldBlk blk = blks[b];
    if ((u32)(blockIdx.x / nblk) >= blk.n_tiles) return;
    if (tid == 0) sh.tile = atomicAdd(tickets + (size_t)blk.ticket * RLD_TSTRIDE, 1u);
    sh.fsm[tid] = g_rld_fsm.v[tid];
    if (tid < 2) *(uint4 *)(sh.in + RLD_TILE + 16 * tid) = make_uint4(0, 0, 0, 0);
    lds_barrier();
    const int k = (int)sh.tile;
    const u64 n = blk.n;
    const u64 pos = (u64)k * RLD_TILE + (u64)tid * RLD_BPL;
    u64 *dst = desc_state + (size_t)blk.desc_base * RLD_DSTRIDE, *dsum = desc_sum + (size_t)blk.desc_base * RLD_DSTRIDE;

    // ---- the lane's 32 bytes -> registers and LDS; zero mask ---------------------------------------------------
    u32 w[9];
    int nvalid = 0;
    uint4 pv0 = make_uint4(0, 0, 0, 0), pv1 = pv0;             // wave 0: the 32 bytes in front of the tile (one address for all lanes)
    if (wv == 0 && k > 0) {
        const u8 *q = blk.in + (u64)k * RLD_TILE - 32;
        pv0 = *(const uint4 *)q;
        pv1 = *(const uint4 *)(q + 16);
    }
    if (pos + RLD_BPL <= n) {
        const uint4 v0 = gload_nt<uint4>(blk.in + pos), v1 = gload_nt<uint4>(blk.in + pos + 16);
        w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w; w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
        nvalid = RLD_BPL;
    } else {
        nvalid = pos < n ? (int)(n - pos) : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = 0;
#pragma unroll
        for (int j = 0; j < RLD_BPL; ++j)
            if (j < nvalid) w[j >> 2] |= (u32)blk.in[pos + j] << (8 * (j & 3));
    }
    *(uint4 *)(sh.in + tid * RLD_BPL) = make_uint4(w[0], w[1], w[2], w[3]);
    *(uint4 *)(sh.in + tid * RLD_BPL + 16) = make_uint4(w[4], w[5], w[6], w[7]);
    if (tid == RLD_THREADS - 1) {       // count / symbol bytes of a triple that starts in the tile's last two bytes
        const u32 la = (pos + 32 < n ? (u32)blk.in[pos + 32] : 0u) | (pos + 33 < n ? (u32)blk.in[pos + 33] << 8 : 0u);
        *(u32 *)(sh.in + RLD_TILE) = la;
    }
    const u32 vm = nvalid >= 32 ? 0xFFFFFFFFu : ((1u << nvalid) - 1u);
    const u32 z = zmask32(w) & vm;

    // ---- this lane's transition map and token starts for each entry state, ordered scan over lanes and waves --------
    // A wave without a zero byte (text-like data: nearly every wave): every byte is a literal once the state is S0, which
    // it is after two bytes whatever it was — no table look-ups.  A TILE of such waves that is entered in S0 emits exactly
    // its input: no per-byte lengths, no image, it is stored straight from `in` (below).
    const bool wave_plain = __all(z == 0u && nvalid == RLD_BPL) != 0;
    u32 st3[3], ex3[3];
    if (wave_plain) {
        st3[0] = 0xFFFFFFFFu; st3[1] = 0xFFFFFFFCu; st3[2] = 0xFFFFFFFEu;
        ex3[0] = ex3[1] = ex3[2] = 0;
    } else {
        fsm32(sh.fsm, z, st3, ex3);
    }
    // The state the tile is entered in, without waiting for anybody: two non-zero bytes in a row end in S0 whatever the state
    // before them (literal literal / symbol count / count literal), so the map of the 32 bytes in front of the tile is
    // almost always constant, and then it IS the entry state.  Only a tile behind 32 bytes of zero-heavy triples looks back.
    u32 fprev = FN_IDENT;
    if (wv == 0 && k > 0) {
        const u32 pw[8] = {pv0.x, pv0.y, pv0.z, pv0.w, pv1.x, pv1.y, pv1.z, pv1.w};
        u32 pst[3], pex[3];
        fsm32(sh.fsm, zmask32(pw), pst, pex);
        fprev = pex[0] | (pex[1] << 2) | (pex[2] << 4);
    }
    // bytes past the end of the block do not move the state (vm clears them from every mask below)
    u32 f = ex3[0] | (ex3[1] << 2) | (ex3[2] << 4);
    u32 fex;
    if (__all(fn_const(f))) {                   // the usual wave: every lane's bytes settle the state whatever it was,
        fex = (u32)__shfl_up((int)f, 1, 64);    // so the map of all bytes before a lane is its neighbour's (constant) map
        if (lane == 0) fex = FN_IDENT;
        if (lane == 63) { sh.wfn[wv] = f; sh.wplain[wv] = wave_plain ? 1u : 0u; }
    } else {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 y = (u32)__shfl_up((int)f, d, 64);
            if (lane >= d) f = fn_compose(y, f);
        }
        if (lane == 63) { sh.wfn[wv] = f; sh.wplain[wv] = 0u; }
        fex = (u32)__shfl_up((int)f, 1, 64);
        if (lane == 0) fex = FN_IDENT;
    }
    lds_barrier();
    u32 wcar = FN_IDENT, ftile = FN_IDENT;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < wv) wcar = fn_compose(wcar, sh.wfn[q]);
        ftile = fn_compose(ftile, sh.wfn[q]);
    }
    const u32 fpre = fn_compose(wcar, fex);     // map of all bytes of the tile before this lane

    if (wv == 0) {
        u32 sin = 0;
        if (k > 0) {
            if (tid == 0) {
                if (fn_const(ftile)) desc_store(dst + (size_t)k * RLD_DSTRIDE, DESC_PREFIX, ftile & 3);
                else desc_store(dst + (size_t)k * RLD_DSTRIDE, DESC_AGG, ftile);
            }
            sin = fn_const(fprev) ? (fprev & 3u) : lookback_state(dst, k, blk.err);
        }
        if (tid == 0) {
            desc_store(dst + (size_t)k * RLD_DSTRIDE, DESC_PREFIX, fn_apply(ftile, sin));
            sh.state_in = sin;
        }
    }
    lds_barrier();

    // ---- literal / escape masks, per-byte output lengths ------------------------------------------------------------
    // (uniform) a tile of waves without zero bytes that is entered at a token start: its output is its input
    const bool plain = sh.state_in == 0u && (sh.wplain[0] & sh.wplain[1] & sh.wplain[2] & sh.wplain[3]) != 0u;
    u32 Lm = 0xFFFFFFFFu, E = 0u, lenw[8], len = RLD_BPL;
    if (!plain) {
        const u32 s_in = fn_apply(fpre, sh.state_in);
        const u32 S = (s_in == 0 ? st3[0] : s_in == 1 ? st3[1] : st3[2]) & vm;
        Lm = S & ~z;
        E = S & z;
        {   // a triple cut by the end of the block: error, no output
            const long long lim = (long long)n - (long long)pos - 2;           // escapes at j < lim have their count byte
            const u32 ok = lim >= 32 ? 0xFFFFFFFFu : lim <= 0 ? 0u : ((1u << (int)lim) - 1u);
            if (E & ~ok) set_error(blk.err, SHAFA_FILE_UNRECOGNIZABLE);
            E &= ok;
        }
        w[8] = *(const u32 *)(sh.in + (tid + 1) * RLD_BPL);                    // the next lane's first bytes / the look-ahead
        len = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u32 fe = nib_flags(E, i);
            const u32 cnt2 = __builtin_amdgcn_alignbit(w[i + 1], w[i], 16);    // the bytes two places on
            lenw[i] = nib_flags(Lm, i) | (cnt2 & ((fe << 8) - fe));
        }
        {   // count byte 0 behaves as 1 (d.c:179-184)
            const u32 t = ((w[8] & 0x7F7Fu) + 0x7F7Fu) | w[8];
            const u32 zn = (~t >> 7) & 0x0101u;
            const u64 z34 = (u64)z | ((u64)((zn & 1u) | ((zn >> 7) & 2u)) << 32);
            const u32 ez = E & (u32)(z34 >> 2);
            if (ez) {
#pragma unroll
                for (int i = 0; i < 8; ++i) lenw[i] |= nib_flags(ez, i);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) len = __builtin_amdgcn_sad_u8(lenw[i], 0u, len);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) lenw[i] = 0x01010101u;
    }

    // ---- output offsets: wave 0 publishes the tile's total and looks back while the other waves build the image ------
    const u32 ilen = wave_incl_scan_add<u32>(len);
    if (lane == 63) sh.wlen[wv] = ilen;
    lds_barrier();
    u32 lbase = 0, ltot = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q == wv) lbase = ltot + ilen - len;
        ltot += sh.wlen[q];
    }
    // wave 0 publishes the tile's total and asks for its predecessors' descriptors now; it reads the answer after its share of
    // the image (first round below), so the round trip costs the tile nothing when the predecessors have published
    u64 lb_first = 0;
    if (wv == 0 && k > 0) {
        if (tid == 0) desc_store(dsum + (size_t)k * RLD_DSTRIDE, DESC_AGG, ltot);
        const int idx = k - 1 - lane;
        lb_first = idx >= 0 ? desc_load(dsum + (size_t)idx * RLD_DSTRIDE) : (DESC_PREFIX << 62);
    }

    // ---- the output, in rounds of what the image holds: image byte i = output byte O + done + i ----------------------
    u8 *smem = (u8 *)&sh;
    const u32 img_off = (u32)offsetof(RldShared, img) + 16u, in_off = (u32)offsetof(RldShared, in) + (u32)tid * RLD_BPL;
    const u32 src_off = plain ? (u32)offsetof(RldShared, in) : img_off;       // where output byte O + done + i sits in LDS
    const u64 limit = blk.out_cap < (u64)SHAFA_RLE_DECODE_MAX ? blk.out_cap : (u64)SHAFA_RLE_DECODE_MAX;
    u64 O = 0;
    for (u32 done = 0;;) {
        u32 nxt = ltot;
        if (ltot - done > (u32)RLD_IMG) {               // (uniform) more than the image holds: consecutive lanes that fit
            if (tid == 0) sh.next = ltot;
            lds_barrier();
            if (len && lbase >= done && (lbase - done) + len > (u32)RLD_IMG) atomicMin(&sh.next, lbase);
            lds_barrier();
            nxt = sh.next;
            lds_barrier();
        }
        if (!plain && len && lbase >= done && lbase + len <= nxt) {
            const u32 p0 = img_off + (lbase - done);
            {   // literals: one byte write each at the running position p, no branches.  The bytes of a triple are written
                // at p as well: p is then the start of a run (filled below, later in program order) or the place of
                // the lane's next literal, which overwrites them — unless nothing follows in this lane (p has reached
                // the end of its output): those go to the lane's dump word
                u32 p = p0;
                const u32 p_end = p0 + len;
                const u32 dump = (u32)offsetof(RldShared, dump) + 4u * (u32)tid;
#pragma unroll
                for (int i = 0; i < RLD_BPL / 4; ++i) {
                    smem[p < p_end ? p : dump] = (u8)w[i];          p = add_byte<0>(p, lenw[i]);
                    smem[p < p_end ? p : dump] = (u8)(w[i] >> 8);   p = add_byte<1>(p, lenw[i]);
                    smem[p < p_end ? p : dump] = (u8)(w[i] >> 16);  p = add_byte<2>(p, lenw[i]);
                    smem[p < p_end ? p : dump] = (u8)(w[i] >> 24);  p = add_byte<3>(p, lenw[i]);
                }
            }
            u32 g = 0;                                  // bytes of the runs so far
            for (u32 e = E; e; e &= e - 1) {            // runs: the lane's escapes in order
                const u32 j = (u32)__builtin_ctz(e);
                const u32 sym = smem[in_off + j + 1], c0 = smem[in_off + j + 2], c = c0 ? c0 : 1u;
                rld_fill(smem, p0 + (u32)__builtin_popcount(Lm & ((1u << j) - 1u)) + g, sym, c);
                g += c;
            }
        }
        if (done == 0 && wv == 0) {
            u64 O0 = 0;
            if (k > 0) {
                O0 = lookback_sum<RLD_DSTRIDE, RLD_SLEEP>(dsum, k, blk.err, true, lb_first);
            }
            if (tid == 0) {
                desc_store(dsum + (size_t)k * RLD_DSTRIDE, DESC_PREFIX, O0 + ltot);
                sh.O = O0;
            }
        }
        lds_barrier();                                // the image is complete (and, the first time, O has arrived)
        if (done == 0) {
            O = sh.O;
            const u64 Oend = O + ltot;
            // The block's return code is the one a sequential decoder stops with (the oracle): the first token whose end
            // passes min(out_cap, MAX) — it is in the one tile with O <= limit < Oend — is FILE_UNRECOGNIZABLE when it passes
            // MAX, LACK_OF_MEMORY otherwise; a triple cut by the end of the block (above) comes behind every token, so
            // LACK_OF_MEMORY replaces its FILE_UNRECOGNIZABLE, whichever tile reports first.
            if (O <= limit && Oend > limit) {           // (uniform)
                if (limit == (u64)SHAFA_RLE_DECODE_MAX) {
                    if (tid == 0) set_error(blk.err, SHAFA_FILE_UNRECOGNIZABLE);
                } else if (Oend <= (u64)SHAFA_RLE_DECODE_MAX) {
                    if (tid == 0) set_error_over(blk.err, SHAFA_LACK_OF_MEMORY, SHAFA_FILE_UNRECOGNIZABLE);
                } else if (tid == 0) {                  // out_cap within a token's length of MAX and the tile passes both: the
                    const u64 tile0 = (u64)k * RLD_TILE;        //   token that passes out_cap decides; thread 0 parses the tile again
                    const u32 nv = n - tile0 < (u64)RLD_TILE ? (u32)(n - tile0) : (u32)RLD_TILE;
                    u64 l = O;
                    int code = SHAFA_LACK_OF_MEMORY;
                    for (u32 i = sh.state_in == 1u ? 2u : sh.state_in == 2u ? 1u : 0u; i < nv; ++i) {
                        u32 c = 1;
                        if (sh.in[i] == 0) {
                            if (tile0 + i + 2 >= n) break;      // cut by the end of the block: no output
                            c = sh.in[i + 2] ? sh.in[i + 2] : 1u;
                            i += 2;
                        }
                        l += c;
                        if (l > limit) { if (l > (u64)SHAFA_RLE_DECODE_MAX) code = SHAFA_FILE_UNRECOGNIZABLE; break; }
                    }
                    if (code == SHAFA_LACK_OF_MEMORY) set_error_over(blk.err, code, SHAFA_FILE_UNRECOGNIZABLE);
                    else set_error(blk.err, code);
                }
            }
            if (k == (int)blk.n_tiles - 1 && tid == 0) *blk.out_n = Oend;
            if (!ltot || O >= limit) return;
        }
        // aligned 16-byte pieces of the output; a piece's bytes sit at image offset 16 u - mis (any alignment)
        u8 *gout = blk.out + O + done;
        const u32 mis = (u32)((uintptr_t)gout & 15u), nbytes = nxt - done;
        const u64 gidx = O + done;                      // index of image byte 0 in the block's output
        for (u32 u = tid; 16 * u < mis + nbytes; u += RLD_THREADS) {
            const u32 s0 = src_off + 16 * u - mis, sb = s0 & ~3u, sf = s0 & 3u;
            const u32 d0 = *(const u32 *)__builtin_assume_aligned(smem + sb, 4), d1 = *(const u32 *)__builtin_assume_aligned(smem + sb + 4, 4),
                      d2 = *(const u32 *)__builtin_assume_aligned(smem + sb + 8, 4), d3 = *(const u32 *)__builtin_assume_aligned(smem + sb + 12, 4),
                      d4 = *(const u32 *)__builtin_assume_aligned(smem + sb + 16, 4);
            const u32 wds[4] = {__builtin_amdgcn_alignbyte(d1, d0, sf), __builtin_amdgcn_alignbyte(d2, d1, sf),
                                __builtin_amdgcn_alignbyte(d3, d2, sf), __builtin_amdgcn_alignbyte(d4, d3, sf)};
            u8 *ga = gout - mis + 16 * u;
            const u32 lo = 16 * u;                      // the piece is image bytes [lo - mis, lo - mis + 16)
            if (lo >= mis && lo + 16 <= mis + nbytes && gidx + (lo - mis) + 16 <= limit) {
                gstore_nt<uint4>(ga, make_uint4(wds[0], wds[1], wds[2], wds[3]));
            } else {
#pragma unroll
                for (u32 q = 0; q < 16; ++q)
                    if (lo + q >= mis && lo + q < mis + nbytes && gidx + (lo + q - mis) < limit)
                        ga[q] = (u8)(wds[q >> 2] >> (8 * (q & 3)));
            }
        }
        done = nxt;
        if (done >= ltot) break;
        lds_barrier();                                // the image is read out before the next round writes it
    }
}



}  // namespace

int rledec_launch(Batch *bt, hipStream_t st, int nblocks, const u8 *d_in, const u64 *h_in_off,
                  const u64 *h_in_n, u8 *d_out, const u64 *h_out_off, const u64 *h_out_cap, u64 *d_out_n)
{
    if (nblocks <= 0) return SHAFA_SUCCESS;
    if (nblocks > bt->max_blocks) return SHAFA_LACK_OF_MEMORY;
    u64 ndesc = 0;
    u32 max_tiles = 0;
    for (int b = 0; b < nblocks; ++b) {
        if ((h_in_off[b] & 15) || (h_out_off[b] & 15)) return SHAFA_OUTSIDE_MODULE;
        const u64 t = ceil_div_u64(h_in_n[b], RLD_TILE);
        ndesc += t;
        if (t > max_tiles) max_tiles = (u32)t;
    }
    size_t off = 0;
    const size_t o_state = off; off += ndesc * 8 * RLD_DSTRIDE;
    const size_t o_sum = off; off += ndesc * 8 * RLD_DSTRIDE;
    const size_t o_tick = off; off += (size_t)nblocks * 4 * RLD_TSTRIDE; off = (off + 15) & ~(size_t)15;
    const size_t o_zero_end = off;
    const size_t o_blk = off; off += (size_t)nblocks * sizeof(RldBlk);
    int rc = batch_reserve(bt, st, off);
    if (rc) return rc;
    u8 *ws = (u8 *)bt->d_ws;
    RldBlk *hb = (RldBlk *)batch_stage(bt, st, (size_t)nblocks * sizeof(RldBlk));
    if (!hb) return SHAFA_LACK_OF_MEMORY;
    u32 dbase = 0;
    for (int b = 0; b < nblocks; ++b) {
        RldBlk &e = hb[b];
        e.in = d_in + h_in_off[b];
        e.out = d_out + h_out_off[b];
        e.n = h_in_n[b];
        e.out_cap = h_out_cap[b];
        e.out_n = d_out_n + b;
        e.err = bt->d_err + b;
        e.desc_base = dbase;
        e.n_tiles = (u32)ceil_div_u64(h_in_n[b], RLD_TILE);
        e.ticket = (u32)b;
        e.pad = 0;
        dbase += e.n_tiles;
    }
    HIP_TRY(hipMemsetAsync(ws, 0, o_zero_end, st));
    HIP_TRY(hipMemsetAsync(d_out_n, 0, (size_t)nblocks * 8, st));
    if ((rc = batch_upload(bt, st, ws + o_blk, hb, (size_t)nblocks * sizeof(RldBlk)))) return rc;
    if (max_tiles) {
        hipLaunchKernelGGL(rle_decode_kernel, dim3(max_tiles * (u32)nblocks), dim3(RLD_THREADS), 0, st,
                           (const RldBlk *)(ws + o_blk), nblocks, (u64 *)(ws + o_state), (u64 *)(ws + o_sum),
                           (u32 *)(ws + o_tick));
        HIP_TRY(hipGetLastError());
    }
    return SHAFA_SUCCESS;
}
