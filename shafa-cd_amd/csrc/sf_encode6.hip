// sf_encode6.hip — Shannon-Fano bit-pack encoder (compress_to_buffer + binary_coding, reference c.c:52-237) for callers that
// hand over the TILE HISTOGRAMS of the input (the sidecar of shafa_hipd_hist256_tiles: 256 x u16 counts per 32 KiB tile,
// a by-product of Module F's make_freq, f.c:63-79): one pass over the input, n read + ceil(bits / 8) written, and no tile
// ever waits for another.
//
// Why: where tile t's bits start in the output is  sum over earlier tiles of (tile histogram . code lengths).  The chained
// encoder (sf_encode4.hip) learns that through a decoupled look-back while it runs, from a persistent grid with a
// rendezvous per tile — its ceiling as a data mover is 0.64 of HBM peak with nothing to compute.  With the histograms at
// hand the offsets exist BEFORE the launch (sfe6_dot, sfe6_scan: 1.6 % of the input's bytes), and the encoder becomes a
// one-shot grid of independent workgroups: no tickets, no descriptors, no look-back, no zeroed workspace.
//
//   sfe6_dot    tile bit totals: hist[t] . len               (reads 512 B per 32 KiB tile)
//   sfe6_scan   per block: exclusive scan -> toff[t] (u64), toff[ntiles] = the block's bits; writes out_n
//   sfe6_kernel one 256-lane workgroup per FULL tile; a block's ragged last tile is sfe4_tail_kernel's (sfe_dev.hpp)
//
// sfe6_kernel, per tile (32 KiB = four sub-tiles of 256 lanes x 32 consecutive symbols): every load of the tile is issued
// up front and waited for once (a wave that has only stores in flight never waits for them: nothing it touches later was
// loaded).  Sub-tile k is turned into octs, scanned (lane -> wave by DPP, wave -> sub-tile through LDS) and placed with
// sfe5's plain-store algebra (emit_oct: every window dword is stored exactly once, partial dwords travel as carries) into
// LDS buffer k & 1 — laid out in the OUTPUT's 16-byte alignment, because the tile's first bit B is known: buffer dword i of
// sub-tile k is the dword 4 * (S_k >> 7) + i behind the 16-byte piece that holds bit B (S_k = the sub-tile's first bit
// counted from that piece).  So the way out is ds_read_b128 -> byte swap -> one aligned non-temporal 16-byte store: no
// funnel shift, one LDS read per piece.  One barrier per sub-tile: iteration k stores sub-tile k-1 and places sub-tile k.
// The piece that straddles two sub-tiles belongs to the later one: its complete dwords are copied to the head of that
// buffer, its partial dword arrives as lane 0's carry.  A tile owns the output dwords [B >> 5, E >> 5): the B mod 32 bits
// in front of its first dword are re-encoded from the 32 symbols before the tile (the lead word: lane 0's first carry), the
// dwords of its first and last piece that it shares with its neighbours are stored one by one, the block's last tile also
// stores the final bytes.  Index algebra checked on the CPU first: tools/model/sfe6_model.py.
//
// The sidecar is trusted for placement only: a workgroup whose bits leave the range [B, E) the offsets promise stops storing
// (nothing is written outside the tile's own region) and a tile whose total differs flags its block SHAFA_OUTSIDE_MODULE.
#include "common.hpp"
#include "internal.hpp"
#include "sfe_dev.hpp"

#include <mutex>

namespace {

constexpr u32 T6_TILE = 32768;                         // symbols per tile = per histogram of the sidecar (SHAFA_TILE_BYTES)
constexpr int T6_SUB = 4;                              // sub-tiles per tile
constexpr int T6_NT = 256;                             // lanes: 32 symbols each per sub-tile
#define E6_WPS 5                                       // workgroups per CU the register allocation aims at


struct E6Static {
    uint4 stage[T6_NT / 64][128];// per wave: 2 KiB of input on its way from load order to symbol order
    u64 lut[256];                // {code (low dword), len (high dword)}; a symbol without a code: len = 1 << 16
    u32 dump[64];                // where the stores of lanes that have nothing to store go (one word per lane)
    u32 wtot[2][4];              // per wave: bits of its string (parity of the sub-tile)
    u32 tail[2][4];              // per wave: the last 32 bits of its string
    u32 ltail[T6_SUB];           // the last 32 bits of every sub-tile's string
};

// ---- tile bit totals: the sidecar's histograms times the code lengths ---------------------------------------------------
// A wave takes 16 tiles: lane l holds the counts of symbols 4 l .. 4 l + 3 of each (8 bytes, 512 B per wave-load, all
// sixteen in flight) and their lengths; totals by DPP, lane i keeps tile i's and lanes 0..15 store them side by side.
__global__ __launch_bounds__(256) void sfe6_dot(const EncBlk *__restrict__ blks, u32 *__restrict__ tbits)
{
    const EncBlk *bp = blks + blockIdx.y;
    const u32 nt = bp->n_tiles;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 t0 = blockIdx.x * 64u + (u32)wv * 16u;
    if (t0 >= nt) return;
    const u64 *lut = (const u64 *)bp->lut;
    u32 l[4], nocode = 0;                              // bit j: symbol 4 lane + j has no code
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 raw = (u32)(gload<u64>(lut + 4 * lane + j) >> 32);
        l[j] = raw & 0xFFFFu;                          // no code: counts as 0 bits
        nocode |= (raw >> 16) << j;
    }
    const u8 *th = (const u8 *)bp->thist + (u64)t0 * 512u;
    uint2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const u32 ti = t0 + (u32)i < nt ? (u32)i : 0u;          // past the block's end: tile t0 again (ignored below)
        v[i] = gload_nt_off<uint2>(th, ti * 512u + (u32)lane * 8u);
    }
    u32 mine = 0, seen = 0;                            // seen bit j: symbol 4 lane + j occurs in one of the tiles
    bool foreign = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        seen |= ((v[i].x & 0xFFFFu) ? 1u : 0u) | ((v[i].x >> 16) ? 2u : 0u) | ((v[i].y & 0xFFFFu) ? 4u : 0u) | ((v[i].y >> 16) ? 8u : 0u);
        u32 s = (v[i].x & 0xFFFFu) * l[0] + (v[i].x >> 16) * l[1] + (v[i].y & 0xFFFFu) * l[2] + (v[i].y >> 16) * l[3];
        s = dpp_scan_add(s);
        u32 tot = (u32)__builtin_amdgcn_readlane((int)s, 63);
        // the counts must be this tile's: they add up to its bytes.  Histograms of something else (an uninitialised or foreign
        // sidecar) would make totals of up to 2^28 bits a tile, offsets that wrap in sfe6_scan and stores outside the
        // block in sfe6_kernel: such a tile counts as empty (its strings then never match: nothing is stored) and the
        // block reports SHAFA_OUTSIDE_MODULE
        const u32 c = dpp_scan_add((v[i].x & 0xFFFFu) + (v[i].x >> 16) + (v[i].y & 0xFFFFu) + (v[i].y >> 16));
        const u32 have = (u32)__builtin_amdgcn_readlane((int)c, 63);
        const u64 at = (u64)(t0 + (u32)i) * T6_TILE;
        const u32 want = t0 + (u32)i < nt ? (bp->n - at < (u64)T6_TILE ? (u32)(bp->n - at) : T6_TILE) : have;
        if (have != want) { foreign = true; tot = 0; }
        if (lane == i) mine = tot;
    }
    if (foreign) set_error(bp->err, SHAFA_OUTSIDE_MODULE);
    if (lane < 16 && t0 + (u32)lane < nt) tbits[bp->desc_base + t0 + (u32)lane] = mine;
    // a data symbol without a code (c.c:156-159): reported here, ahead of sfe6_scan's size check, as the other encoders
    // meet it before they run out of room
    if (seen & nocode) set_error_over(bp->err, SHAFA_FILE_UNRECOGNIZABLE, SHAFA_LACK_OF_MEMORY);
}

// ---- per block: exclusive scan of the tile totals -> tile offsets; the block's size ---------------------------------------
__global__ __launch_bounds__(1024) void sfe6_scan(const EncBlk *__restrict__ blks, const u32 *__restrict__ tbits,
                                                  u64 *__restrict__ toff)
{
    __shared__ u32 wsum[16];
    const EncBlk *bp = blks + blockIdx.x;
    const u32 nt = bp->n_tiles, base = bp->desc_base;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    u64 carry = 0;
    for (u32 c0 = 0; c0 < nt; c0 += 1024u) {                   // 1024 tiles of <= 2^20 bits: the chunk's sum fits 32 bits
        const u32 t = c0 + (u32)tid;
        const u32 v = t < nt ? tbits[base + t] : 0u;
        const u32 incl = dpp_scan_add(v);
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        const u32 w = lane < 16 ? wsum[lane] : 0u;
        const u32 wincl = dpp_scan_add(w);
        const u32 woff = (u32)__builtin_amdgcn_readlane((int)(wincl - w), wv);
        const u32 tot = (u32)__builtin_amdgcn_readlane((int)wincl, 15);
        if (t < nt) toff[base + t] = carry + woff + incl - v;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) {
        toff[base + nt] = carry;
        const u64 bytes = (carry + 7) >> 3;
        gstore<u64>(bp->out_n, bytes);
        if (bytes > bp->out_cap) set_error(bp->err, SHAFA_LACK_OF_MEMORY);
    }
}

// ---- a sub-tile leaves LDS: its buffer is laid out like the output ---------------------------------------------------------
__device__ __forceinline__ uint4 e6_swap(const e5_u32x4 a)
{
    return make_uint4(bswap32(a.x), bswap32(a.y), bswap32(a.z), bswap32(a.w));
}
// pieces [p0, np) of the buffer at LDS byte address lds -> o + 16 p; two pieces per lane in flight
__device__ __forceinline__ void e6_store_pieces(u32 lds, u8 *o, u32 p0, u32 np, int tid)
{
    for (u32 p = p0 + (u32)tid; p < np; p += 2u * T6_NT) {
        const u32 p2 = p + T6_NT;
        const bool two = p2 < np;                      // nearly wave-uniform
        e5_u32x4 a0 = e5_lds_read128(lds + 16u * p), a1 = a0;
        if (two) a1 = e5_lds_read128(lds + 16u * p2);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1) : : "memory");
        gstore_nt_off<uint4>(o, 16u * p, e6_swap(a0));
        if (two) gstore_nt_off<uint4>(o, 16u * p2, e6_swap(a1));
    }
}

// NW: dwords an oct can touch (3: Lmax <= 8, 4: <= 12, 5: <= 16); L16: Lmax == 16.  wstride: dwords per LDS buffer.
template <int NW, bool L16>
__global__ __launch_bounds__(T6_NT, E6_WPS) void sfe6_kernel(const EncBlk *__restrict__ blks, const u64 *__restrict__ toff,
                                                               u32 wstride)
{
    __shared__ E6Static sh;
    extern __shared__ __attribute__((aligned(16))) u32 dynwin[];     // two buffers of wstride dwords
    const EncBlk *bp = blks + blockIdx.y;
    const u32 t = blockIdx.x;
    const u64 n = bp->n;
    const u32 nfull = (u32)(n >> 15);
    if (t >= nfull) return;                            // full tiles only
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u8 *in = bp->in + (u64)t * T6_TILE;

    // ---- everything this workgroup ever loads, requested at once ---------------------------------------------------
    // Fully coalesced non-temporal loads: of its 2 KiB of every sub-tile a wave loads piece (16 bytes) l and piece 64 + l
    // into lane l; the lane's own 32 consecutive symbols (pieces 2 l and 2 l + 1) come out of a wave-private LDS round
    // trip right before they are looked up.  (Loaded where they are needed — two 16-byte loads at a 32-byte lane stride,
    // half of every line per instruction — the same traffic moves 7 to 10 % slower: tools/ubench/mover.hip.)
    TileIn5 tin[T6_SUB];
#pragma unroll
    for (int k = 0; k < T6_SUB; ++k) {
        tin[k].v[0] = gload_nt_off<uint4>(in, (u32)k * 8192u + (u32)wv * 2048u + (u32)lane * 16u);
        tin[k].v[1] = gload_nt_off<uint4>(in, (u32)k * 8192u + (u32)wv * 2048u + 1024u + (u32)lane * 16u);
    }
    const u64 lut_e = gload<u64>((const u64 *)bp->lut + tid);
    u32 pv = 0;                                        // wave 0, lanes 0..31: symbol (tile start - 1 - lane)
    if (wv == 0 && t > 0 && lane < 32) pv = gload_off<u8>(in - 32, 31u - (u32)lane);
    const u64 *tp = toff + bp->desc_base + t;
    const u64 Bv = gload<u64>(tp), Ev = gload<u64>(tp + 1);
    sh.lut[tid] = lut_e;
    __syncthreads();                                   // the one wait for global memory (the barrier drains vmcnt)

    const u32 B_lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)Bv), B_hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(Bv >> 32));
    const u32 E_lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)Ev), E_hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(Ev >> 32));
    const u64 B = ((u64)B_hi << 32) | B_lo, E = ((u64)E_hi << 32) | E_lo;
    if (((E + 7) >> 3) > bp->out_cap) return;          // the block does not fit (sfe6_scan flagged it): store nothing
    if (B > E || E - B > (u64)T6_TILE * 32u) return;   // offsets no tile of codes of <= 32 bits can have
    u8 *const o_tile = bp->out + 16ull * (B >> 7);     // the 16-byte piece that holds the tile's first bit
    const u32 S0 = B_lo & 127u;                        // bit positions below count from that piece
    const u32 S_end = S0 + (u32)(E - B);               // where the offsets say the tile ends
    const u32 d0 = S0 >> 5;                            // the tile's first dword inside its head piece

    // ---- lead word: the last bits in front of the tile, right-aligned (bit i = stream bit B - 1 - i) ------------
    u32 lead = 0;
    if (wv == 0) {
        const u64 ent = sh.lut[pv & 0xFFu];
        const u32 len = lane < 32 ? ((u32)(ent >> 32) & 0xFFFFu) : 0u;
        const u32 D = dpp_scan_add(len);               // bits from this symbol's first bit to the tile's start
        u32 x = (len && D - len < 32u) ? ((u32)ent << ((D - len) & 31u)) : 0u;
        x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
        x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
        x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
        x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8
        lead = (u32)__builtin_amdgcn_readlane((int)x, 15) | (u32)__builtin_amdgcn_readlane((int)x, 31);
    }

    u32 S[T6_SUB + 1];                                 // S[k]: first bit of sub-tile k (uniform)
    S[0] = S0;
    bool good = true;                                  // the strings stay inside [B, E): the offsets match the data
#pragma unroll
    for (int k = 0; k < T6_SUB; ++k) {
        const u32 par = (u32)k & 1u;
        u32 *win = dynwin + par * wstride;
        Oct oct[4];
        u32 tot = 0, incl = 0, tail = 0, absent = 0;
        {   // pieces l and 64 + l of the wave's 2 KiB -> pieces 2 l and 2 l + 1 (same wave: LDS operations of one wave execute
            // in order; the wave barriers keep the compiler from moving the reads above the writes)
            uint4 *stg = sh.stage[wv];
            // piece j sits in slot j ^ (bit 4 of j): the 16-byte reads at a 32-byte lane stride are bank-conflict free as well
            const int wslot = lane ^ ((lane >> 4) & 1), rb = (lane >> 3) & 1;
            stg[wslot] = tin[k].v[0];
            stg[64 + wslot] = tin[k].v[1];
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            tin[k].v[0] = stg[2 * lane + rb];
            tin[k].v[1] = stg[2 * lane + (rb ^ 1)];
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
        }
        tile_octs5<NW, L16>(sh.lut, tin[k], oct, tot, incl, tail, absent);
        if (absent) set_error_over(bp->err, SHAFA_FILE_UNRECOGNIZABLE, SHAFA_LACK_OF_MEMORY);          // data symbol without a code (output undefined, in bounds)
        if (lane == 63) {
            sh.wtot[par][wv] = incl;
            sh.tail[par][wv] = tail;
            if (wv == T6_NT / 64 - 1) sh.ltail[k] = tail;
        }
        // The only barrier of the iteration.  Behind it: buffer par was last read by the stores of iteration k - 1 (sub-tile
        // k - 2), buffer par ^ 1 was filled in iteration k - 1, the hand-over words of this parity were read in iteration k - 2.
        lds_barrier();
        const uint4 wt = *(const uint4 *)sh.wtot[par];
        const u32 w0 = (u32)__builtin_amdgcn_readfirstlane((int)wt.x), w1 = (u32)__builtin_amdgcn_readfirstlane((int)wt.y);
        const u32 w2 = (u32)__builtin_amdgcn_readfirstlane((int)wt.z), w3 = (u32)__builtin_amdgcn_readfirstlane((int)wt.w);
        const u32 woff = wv == 0 ? 0u : wv == 1 ? w0 : wv == 2 ? w0 + w1 : w0 + w1 + w2;
        S[k + 1] = S[k] + w0 + w1 + w2 + w3;
        good = good && S[k + 1] <= S_end;

        auto leave = [&]() {
        if (k > 0) {                                   // sub-tile k - 1 leaves
                const u32 *pbuf = dynwin + (par ^ 1u) * wstride;
                const u32 np = (S[k] >> 7) - (S[k - 1] >> 7);
                // the piece sub-tile k starts in: its complete dwords move to the head of this buffer
                if ((u32)tid < ((S[k] >> 5) & 3u)) win[tid] = pbuf[4u * np + (u32)tid];
                if (good) {
                    const bool head = k == 1 && d0 != 0u;  // the tile's head piece is shared with the tile before
                    e6_store_pieces(lds_addr(pbuf), o_tile + 16u * (S[k - 1] >> 7), head ? 1u : 0u, np, tid);
                    if (head && (u32)tid >= d0 && tid < 4) gstore_off<u32>(o_tile, 4u * (u32)tid, bswap32(pbuf[tid]));
                }
            }
        };
        auto place_it = [&]() {
            // place sub-tile k: the lane's string is buffer bits [El - tot, El)
            const u32 El = (S[k] & 127u) + woff + incl;
            u32 e = El - tot;
            const u32 fin = __builtin_amdgcn_alignbit(tail, 0u, El);      // the lane's last partial dword
            u32 c = (u32)__builtin_amdgcn_update_dpp(0, (int)fin, 0x138, 0xf, 0xf, false);       // wave_shr:1 (lane 0 keeps 0)
            if (lane == 0) {                               // the string in front: previous wave, previous sub-tile, lead word
                const u32 prev = wv ? sh.tail[par][wv - 1] : (k ? sh.ltail[k - 1] : lead);
                c = __builtin_amdgcn_alignbit(prev, 0u, e);
            }
            {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    e += oct[q].ll;
                    c = emit_oct<NW>(win, sh.dump + lane, oct[q], e, c);
                }
            }
            if (k == T6_SUB - 1 && tid == T6_NT - 1) win[e >> 5] = c;      // the tile's final partial dword (the block's last bytes)
        };
        leave();
        place_it();
    }
    lds_barrier();
    {
        const u32 *pbuf = dynwin + ((T6_SUB - 1) & 1) * wstride;
        const u32 np = (S[T6_SUB] >> 7) - (S[T6_SUB - 1] >> 7);
        if (good && S[T6_SUB] == S_end) {
            u8 *o = o_tile + 16u * (S[T6_SUB - 1] >> 7);
            e6_store_pieces(lds_addr(pbuf), o, 0u, np, tid);
            // the complete dwords of the piece the tile ends in; the block's last tile: the final 1..3 bytes as well
            const u32 cnt = (S_end >> 5) & 3u;
            if ((u32)tid < cnt) gstore_off<u32>(o, 16u * np + 4u * (u32)tid, bswap32(pbuf[4u * np + (u32)tid]));
            const bool last = t == nfull - 1 && (n & (T6_TILE - 1)) == 0;
            const u32 nb = ((E_lo & 31u) + 7u) >> 3;
            if (last && tid >= 8 && (u32)tid - 8u < nb)
                gstore_off<u8>(o, 16u * np + 4u * cnt + ((u32)tid - 8u), (u8)(pbuf[4u * np + cnt] >> (24u - 8u * ((u32)tid - 8u))));
        } else if (tid == 0) set_error(bp->err, SHAFA_OUTSIDE_MODULE);   // the sidecar is not this block's
    }
}

}  // namespace


// Blocks whose codes are <= 16 bits, tables 256 x u64 {code, len}; blk.thist = the block's tile histograms, blk.desc_base =
// its first entry in d_tbits / d_toff (n_tiles + 1 entries per block), blk.n_tiles = ceil(n / 32768).
int sfenc6_launch(hipStream_t st, const EncBlk *dblk, int count, u32 max_tiles, u32 lmax, bool any_ragged, u32 *d_tbits, u64 *d_toff)
{
    constexpr int MAXDEV = 64;
    static int tail_attr_by_dev[MAXDEV];
    static std::mutex mu;
    if (lmax < 1 || lmax > 16 || count <= 0) return SHAFA_OUTSIDE_MODULE;
    const u32 wstride = (256u * lmax + 8u + 3u) & ~3u;               // dwords per buffer: 127 + 8192 lmax bits, the final dword
    const size_t dyn = (size_t)wstride * 2 * 4;
    hipLaunchKernelGGL(sfe6_dot, dim3((max_tiles + 63) / 64, (u32)count), dim3(256), 0, st, dblk, d_tbits);
    hipLaunchKernelGGL(sfe6_scan, dim3((u32)count), dim3(1024), 0, st, dblk, (const u32 *)d_tbits, d_toff);
    const dim3 grid(max_tiles, (u32)count);
    if (lmax <= 8) hipLaunchKernelGGL((sfe6_kernel<3, false>), grid, dim3(T6_NT), dyn, st, dblk, (const u64 *)d_toff, wstride);
    else if (lmax <= 12) hipLaunchKernelGGL((sfe6_kernel<4, false>), grid, dim3(T6_NT), dyn, st, dblk, (const u64 *)d_toff, wstride);
    else if (lmax <= 15) hipLaunchKernelGGL((sfe6_kernel<5, false>), grid, dim3(T6_NT), dyn, st, dblk, (const u64 *)d_toff, wstride);
    else hipLaunchKernelGGL((sfe6_kernel<5, true>), grid, dim3(T6_NT), dyn, st, dblk, (const u64 *)d_toff, wstride);
    if (any_ragged) {                                  // one 1024-lane workgroup per block: its remainder is < 32 KiB
        const u32 tail_stride = ((u32)E4_GUARD + 1024u * lmax + 8u + 3u) & ~3u;
        const size_t tdyn = (size_t)tail_stride * 4;
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= MAXDEV) return SHAFA_OUTSIDE_MODULE;
        if (tdyn > 65536) {                            // the attribute only ever grows (one value per device for all four forms)
            std::lock_guard<std::mutex> lk(mu);
            if (tail_attr_by_dev[dev] < 4) {
                for (const void *f : {(const void *)sfe4_tail_kernel<3, false, 1024>, (const void *)sfe4_tail_kernel<4, false, 1024>,
                                      (const void *)sfe4_tail_kernel<5, false, 1024>, (const void *)sfe4_tail_kernel<5, true, 1024>})
                    HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)((((u32)E4_GUARD + 1024u * 16u + 8u + 3u) & ~3u) * 4u)));
                tail_attr_by_dev[dev] = 4;
            }
        }
        if (lmax <= 8) hipLaunchKernelGGL((sfe4_tail_kernel<3, false, 1024>), dim3((u32)count), dim3(1024), tdyn, st, dblk, (const u64 *)nullptr, tail_stride, (const u32 *)nullptr, (const u64 *)d_toff);
        else if (lmax <= 12) hipLaunchKernelGGL((sfe4_tail_kernel<4, false, 1024>), dim3((u32)count), dim3(1024), tdyn, st, dblk, (const u64 *)nullptr, tail_stride, (const u32 *)nullptr, (const u64 *)d_toff);
        else if (lmax <= 15) hipLaunchKernelGGL((sfe4_tail_kernel<5, false, 1024>), dim3((u32)count), dim3(1024), tdyn, st, dblk, (const u64 *)nullptr, tail_stride, (const u32 *)nullptr, (const u64 *)d_toff);
        else hipLaunchKernelGGL((sfe4_tail_kernel<5, true, 1024>), dim3((u32)count), dim3(1024), tdyn, st, dblk, (const u64 *)nullptr, tail_stride, (const u32 *)nullptr, (const u64 *)d_toff);
    }
    HIP_TRY(hipGetLastError());
    return SHAFA_SUCCESS;
}
