// sf_encode.hip — Module C hot path: Shannon-Fano bit-pack encode on gfx950, host launcher + the generic kernel.
//
// Replaces compress_to_buffer + binary_coding (reference c.c:52-237): the block's code bit-strings
// concatenated MSB-first, zero padded to a byte.  One launch handles many independent blocks; blocks are
// classified by their longest code:
//   class 1, Lmax <= 16 (every Shannon-Fano table of a real 64 MiB block with a flat enough histogram):
//            sf_encode4.hip (one pass, chained scan; launches with >= SFE4_MIN_BLOCKS blocks) or
//            sf_encode3.hip (count / scan / pack; few blocks per launch, where a chain would be too long);
//   class 2, Lmax <= 32: sf_encode3.hip with 64-bit table entries;
//   class 3, Lmax <= 255 (hand-made / foreign .cod tables): sf_encode_generic below, one symbol per lane and step,
//            tiles chained by a decoupled look-back on one 64-bit {status,value} word per tile.
//
// Algorithmic HBM bytes per block: n read + ceil(sum(freq*len)/8) written.
#include "common.hpp"
#include "internal.hpp"

#include <stdlib.h>

namespace {

constexpr int ENC_THREADS = 256;
constexpr int ENC_STAGE_WORDS = 5120;   // LDS bit-stream window per round (20 KiB)

// OR an L-bit group (right-aligned in g, 1 <= L <= 64) into the LDS bit-stream at tile-local bit q.
// stage points at local word `wlo`; words outside [wlo, wlo+wcount) are skipped (other rounds).
__device__ __forceinline__ void emit_group(u32 *stage, u32 wlo, u32 wcount, u64 g, u32 L, u32 q)
{
    if (L == 0) return;
    const u64 hi = g << (64 - L);                 // left-aligned
    const u32 sh = q & 31;
    const u64 x = hi >> sh;
    const u32 w0 = (u32)(x >> 32), w1 = (u32)x;
    const u32 w2 = (u32)((((u64)(u32)hi) << 32) >> sh);
    const u32 i0 = (q >> 5) - wlo;                 // wraps for words before the window
    if (w0 && i0 < wcount) atomicOr(stage + i0, w0);
    if (w1 && i0 + 1 < wcount) atomicOr(stage + i0 + 1, w1);
    if (w2 && i0 + 2 < wcount) atomicOr(stage + i0 + 2, w2);
}

// ------------------------------------------------------------------------------------------------
// tail of the generic kernel: chain the tile total, compute the lead bits, run the rounds.
// EmitFn(stage, wlo, wcount) ORs this thread's bits of the window into the stage.
// LeadFn(need) returns the last `need` bits (1..31) of the stream that precedes the tile.
// ------------------------------------------------------------------------------------------------
struct EncShared {
    u32 stage[ENC_STAGE_WORDS + 4];   // [0] = carry/lead word, [1..] = window
    u32 wtot[16];
    u32 tile;
    u32 pad;
    u64 prefix;
};

template <typename EmitFn, typename LeadFn>
__device__ __forceinline__ void encode_tail(EncShared &sh, const EncBlk &blk, u64 *desc, int k,
                                            u32 tile_bits, EmitFn emit, LeadFn lead_bits)
{
    const int tid = threadIdx.x;
    u64 *bdesc = desc + blk.desc_base;
    const bool last = (k == (int)blk.n_tiles - 1);

    if (wave_id() == 0) {
        u64 B = 0;
        if (k > 0) {
            if (tid == 0) desc_store(bdesc + k, DESC_AGG, tile_bits);
            B = lookback_sum(bdesc, k, blk.err);
        }
        if (tid == 0) {
            desc_store(bdesc + k, DESC_PREFIX, B + tile_bits);
            sh.prefix = B;
            const u32 s = (u32)B & 31;
            sh.stage[0] = s ? lead_bits(s) : 0u;
        }
    }
    // zero the window (stage[0] is owned by thread 0 above)
    for (int i = tid; i < ENC_STAGE_WORDS; i += ENC_THREADS) sh.stage[1 + i] = 0;
    __syncthreads();

    const u64 B = sh.prefix;
    const u64 E = B + tile_bits;
    const u32 s = (u32)B & 31;
    const u64 wg_lo = B >> 5;
    const u64 wg_hi = last ? ((E + 31) >> 5) : (E >> 5);
    const u32 OW = (u32)(wg_hi - wg_lo);
    const u64 total_bytes = (E + 7) >> 3;          // meaningful for the last tile only
    const u32 need_words = OW;                     // rounds only have to cover the words we store
    u32 *out32 = (u32 *)blk.out;

    for (u32 r0 = 0; r0 < need_words; r0 += ENC_STAGE_WORDS) {
        if (r0) {   // later rounds (tile expands beyond one window): carry the last word, re-zero
            __syncthreads();
            u32 carry = 0;
            if (tid == 0) carry = sh.stage[ENC_STAGE_WORDS];
            __syncthreads();
            for (int i = tid; i < ENC_STAGE_WORDS; i += ENC_THREADS) sh.stage[1 + i] = 0;
            if (tid == 0) sh.stage[0] = carry;
            __syncthreads();
        }
        emit(sh.stage + 1, r0, (u32)ENC_STAGE_WORDS);
        __syncthreads();
        u32 jend = (OW < r0 + ENC_STAGE_WORDS) ? OW : r0 + ENC_STAGE_WORDS;
        for (u32 j = r0 + tid; j < jend; j += ENC_THREADS) {
            const u32 li = j - r0;
            const u32 w = funnel_r(sh.stage[li], sh.stage[li + 1], s);
            const u64 W = wg_lo + j;
            if (last && W == wg_hi - 1 && total_bytes < 4 * (W + 1)) {
                const u32 nb = (u32)(total_bytes - 4 * W);     // 1..3 bytes of the final word
                if (total_bytes <= blk.out_cap) {
                    for (u32 q = 0; q < nb; ++q) blk.out[4 * W + q] = (u8)(w >> (24 - 8 * q));
                } else set_error(blk.err, SHAFA_LACK_OF_MEMORY);
            } else if (4 * (W + 1) <= blk.out_cap) {
                out32[W] = bswap32(w);
            } else set_error(blk.err, SHAFA_LACK_OF_MEMORY);
        }
    }
    if (last && tid == 0) *blk.out_n = total_bytes;
}

// ------------------------------------------------------------------------------------------------
// generic kernel: any code length up to 255 bits (hand-made / foreign .cod tables).  One symbol per
// lane and step, 1024-symbol tiles; the table (len + 32 code bytes per symbol) is read from global.
// ------------------------------------------------------------------------------------------------
constexpr int GEN_SYMS = 4;                       // symbols per thread
constexpr int GEN_TILE = ENC_THREADS * GEN_SYMS;  // 1024 symbols, <= 32640 bytes of output

// bits [from, from+cnt) of symbol s's code (cnt <= 32), right-aligned
__device__ __forceinline__ u32 code_bits(const shafa_code_table *t, u32 s, u32 from, u32 cnt)
{
    u32 v = 0;
    for (u32 i = 0; i < cnt; ++i) {
        const u32 bit = from + i;
        v = (v << 1) | ((t->bits[s][bit >> 3] >> (7 - (bit & 7))) & 1u);
    }
    return v;
}

__global__ __launch_bounds__(ENC_THREADS) void sf_encode_generic(const EncBlk *__restrict__ blks, int nblk,
                                                                 u64 *desc, u32 *tickets)
{
    __shared__ __attribute__((aligned(16))) EncShared sh;
    const int tid = threadIdx.x;
    const int b = blockIdx.x % nblk;
    const EncBlk blk = blks[b];
    if ((u32)(blockIdx.x / nblk) >= blk.n_tiles) return;
    if (tid == 0) sh.tile = atomicAdd(tickets + blk.ticket, 1u);
    __syncthreads();
    const int k = (int)sh.tile;
    const u64 tile_start = (u64)k * GEN_TILE;
    const shafa_code_table *tab = (const shafa_code_table *)blk.lut;

    u32 sym[GEN_SYMS], len[GEN_SYMS];
    u32 tot = 0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < GEN_SYMS; ++j) {
        const u64 idx = tile_start + (u64)tid * GEN_SYMS + j;
        sym[j] = 0; len[j] = 0;
        if (idx < blk.n) {
            sym[j] = blk.in[idx];
            len[j] = tab->len[sym[j]];
            if (len[j] == 0) bad = true;
        }
        tot += len[j];
    }
    if (bad) set_error_over(blk.err, SHAFA_FILE_UNRECOGNIZABLE, SHAFA_LACK_OF_MEMORY);

    const u32 incl = wave_incl_scan_add<u32>(tot);
    if (lane_id() == 63) sh.wtot[wave_id()] = incl;
    __syncthreads();
    u32 run = 0, off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w == wave_id()) off = run + incl - tot;
        run += sh.wtot[w];
    }
    const u32 tile_bits = run;

    auto emit = [&](u32 *stage, u32 wlo, u32 wcount) {
        u32 q = off;
#pragma unroll
        for (int j = 0; j < GEN_SYMS; ++j) {
            for (u32 done = 0; done < len[j]; done += 32) {
                const u32 c = (len[j] - done) < 32 ? (len[j] - done) : 32;
                emit_group(stage, wlo, wcount, code_bits(tab, sym[j], done, c), c, q + done);
            }
            q += len[j];
        }
    };
    auto lead = [&](u32 need) -> u32 {
        u32 acc = 0, got = 0;
        for (u64 p = tile_start; p > 0 && got < need;) {
            --p;
            const u32 s = blk.in[p];
            const u32 l = tab->len[s];
            const u32 take = (need - got) < l ? (need - got) : l;   // the code's LAST `take` bits
            if (take) acc |= code_bits(tab, s, l - take, take) << got;
            got += take;
        }
        return acc;
    };
    encode_tail(sh, blk, desc, k, tile_bits, emit, lead);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host launcher
// ------------------------------------------------------------------------------------------------
void sfenc3_launch(hipStream_t st, const EncBlk *dblk, int count, u32 max_tiles, u32 *d_tile_bits, u64 *d_tile_off, bool lut64);
int sfenc4_launch(hipStream_t st, const EncBlk *dblk, int count, u64 *d_desc, u32 *d_tickets, u32 lmax, u32 ragged, const SfeRedo &x);
bool sfenc4_needs_redo(u32 lmax);
bool sfenc4_long_ok();
int sfenc4_launch_long(hipStream_t st, const EncBlk *dblk, int count, u64 *d_desc, u32 *d_tickets, u32 lmax, u32 ragged, const SfeRedo &x);
int sfenc6_launch(hipStream_t st, const EncBlk *dblk, int count, u32 max_tiles, u32 lmax, bool any_ragged, u32 *d_tbits, u64 *d_toff);
extern int g_sfe4_wide, g_sfe_lanes;

// A launch with at least this many class-1 blocks takes the one-pass encoder: every block is its own chain, and with this
// many chains (<= ~10 workgroups per block) a tile's look-back stays inside one 64-entry descriptor window.  Measured
// on 64 MiB Zipf blocks, GiB/s one-pass vs count/scan/pack: 16 blocks 588 / 1240, 32: 1112 / 1298, 64: 1407 / 1457,
// 128: 1900 / 1632 (sf_encode4.hip).
// launches with at least this many blocks of <= 16-bit codes take the one-pass encoder; 0 = the measured crossovers:
// 6 blocks where the 1024-lane form runs (Lmax <= 12), 80 for the 256-lane form
static int g_sfe4_min_blocks = 0;
void sfenc_configure(int sfe4_min_blocks) { g_sfe4_min_blocks = sfe4_min_blocks; }

// the code of symbol s (<= 32 bits: classes 1 and 2) right-aligned: its first four bytes, MSB first, shifted down.
// (Bit by bit this loop was most of the 1.3 ms the host needed to prepare a 128-block launch: more than the kernel takes
// on 1 GiB in 8 MiB blocks.)
static u32 code_value(const shafa_code_table &t, int s)
{
    const u32 len = t.len[s];
    if (!len) return 0;
    const u8 *b = t.bits[s];
    const u32 be = ((u32)b[0] << 24) | ((u32)b[1] << 16) | ((u32)b[2] << 8) | (u32)b[3];
    return len >= 32 ? be : be >> (32 - len);
}

int sfenc_launch(Batch *bt, hipStream_t st, int nblocks, const u8 *d_in, const u64 *h_in_off,
                 const u64 *h_in_n, const shafa_code_table *h_tables, u8 *d_out, const u64 *h_out_off,
                 const u64 *h_out_cap, u64 *d_out_n, const u8 *d_thist, const u64 *h_thist_off)
{
    if (nblocks <= 0) return SHAFA_SUCCESS;
    if (nblocks > bt->max_blocks) return SHAFA_LACK_OF_MEMORY;
    if ((d_thist == nullptr) != (h_thist_off == nullptr)) return SHAFA_OUTSIDE_MODULE;
    // with the tile histograms at hand, blocks of <= 16-bit codes take the one-shot encoder (sf_encode6.hip), whatever their number
    const bool tiles = d_thist != nullptr;

    // classify blocks: 0 = nothing to launch (empty table or empty block), 1 = Lmax <= 16, 2 = Lmax <= 32, 3 = generic
    int cls_count[4] = {0, 0, 0, 0};
    std::vector<int> cls(nblocks);
    u32 lmax1 = 0, lmax2 = 0;
    for (int b = 0; b < nblocks; ++b) {
        if ((h_in_off[b] & 15) || (h_out_off[b] & 15) || (tiles && (h_thist_off[b] & 15))) return SHAFA_OUTSIDE_MODULE;
        int lmax = 0;
        for (int s = 0; s < 256; ++s) lmax = h_tables[b].len[s] > lmax ? h_tables[b].len[s] : lmax;
        const int c = (lmax == 0 || h_in_n[b] == 0) ? 0 : (lmax <= 16 ? 1 : (lmax <= 32 ? 2 : 3));
        cls[b] = c;
        cls_count[c]++;
        if (c == 1 && (u32)lmax > lmax1) lmax1 = (u32)lmax;
        if (c == 2 && (u32)lmax > lmax2) lmax2 = (u32)lmax;
    }
    // the one-pass encoder pays from 6 blocks per launch in its 1024-lane form (every Lmax <= 16 since the windows of
    // 13..16-bit codes are sized for 12 bits per symbol with an encode-again fall-back), from 80 in the 256-lane form
    const bool wide_form = g_sfe4_wide && g_sfe_lanes == 0 && (lmax1 <= 12 || sfenc4_needs_redo(lmax1));
    const bool one_pass = tiles || cls_count[1] >= (g_sfe4_min_blocks > 0 ? g_sfe4_min_blocks : (wide_form ? 6 : 80));
    // codes of 17..32 bits: the quad form of the one-pass encoder from 6 blocks per launch, else count / scan / pack with
    // 64-bit groups
    const bool one_pass2 = cls_count[2] >= (g_sfe4_min_blocks > 0 ? g_sfe4_min_blocks : 6) && sfenc4_long_ok();
    const bool redo = (one_pass && !tiles && cls_count[1] && sfenc4_needs_redo(lmax1)) || one_pass2;
    // class 1 with tile histograms: 32 KiB tiles, and one entry more per block (the block's total behind its tile offsets)
    const u64 tile_syms[4] = {1, tiles ? 32768u : 256 * 16 * 2, 256 * 16 * 2, GEN_TILE};
    u64 total_tiles[4] = {0, 0, 0, 0};
    u32 max_tiles[4] = {0, 0, 0, 0};
    for (int b = 0; b < nblocks; ++b) {
        const int c = cls[b];
        if (!c) continue;
        const u64 t = ceil_div_u64(h_in_n[b], tile_syms[c]);
        total_tiles[c] += t + (c == 1 && tiles ? 1 : 0);
        if (t > max_tiles[c]) max_tiles[c] = (u32)t;
    }
    const u64 ndesc = total_tiles[1] + total_tiles[2] + total_tiles[3];
    if (ndesc >= 0xFFFFFFFFull) return SHAFA_LACK_OF_MEMORY;

    // device workspace: [desc u64 * ndesc][tickets u32 * nblocks] (zeroed) [tile bits u32 * ndesc]; the parameter buffer
    // (uploaded on the side stream, batch_params_*): [EncBlk * nblocks][tables]
    const size_t tab1 = one_pass ? 2048 : 1024;
    size_t off = 0;
    const size_t o_desc = off; off += ndesc * 8;
    const size_t o_tick = off; off += (size_t)nblocks * 4; off = (off + 15) & ~(size_t)15;
    const size_t o_desc2 = off; off += redo ? ndesc * 8 : 0;                 // the encode-again pass's chain and the flags
    const size_t o_tick2 = off; off += redo ? (size_t)nblocks * 4 : 0; off = (off + 15) & ~(size_t)15;
    const size_t o_redo = off; off += redo ? (size_t)nblocks * 4 : 0; off = (off + 15) & ~(size_t)15;
    const size_t o_zero_end = off;
    const size_t o_tbits = off; off += ndesc * 4; off = (off + 15) & ~(size_t)15;
    const size_t ws_bytes = off;
    const size_t o_blk = off; off += (size_t)nblocks * sizeof(EncBlk); off = (off + 15) & ~(size_t)15;
    const size_t o_tab = off;
    size_t tab_bytes = 0;
    for (int b = 0; b < nblocks; ++b)
        tab_bytes += cls[b] == 1 ? tab1 : cls[b] == 2 ? 2048 : cls[b] == 3 ? ((sizeof(shafa_code_table) + 15) & ~15ul) : 0;
    off += tab_bytes;
    int rc = batch_reserve(bt, st, ws_bytes);
    if (rc) return rc;
    u8 *ws = (u8 *)bt->d_ws;

    // host staging: EncBlk array + tables, one H2D copy on the side stream
    const size_t stage_bytes = off - o_blk;
    u8 *dpar = batch_params_begin(bt, stage_bytes);
    if (!dpar) return SHAFA_LACK_OF_MEMORY;
    ParamsScope pscope(bt, st);                        // every return below records the buffer's last reader
    u8 *hs = (u8 *)batch_stage(bt, bt->par_inline ? st : bt->copy_st, stage_bytes);
    if (!hs) return SHAFA_LACK_OF_MEMORY;
    EncBlk *hb = (EncBlk *)hs;
    u8 *htab = hs + (o_tab - o_blk);
    size_t tpos = 0;
    u32 dbase = 0;
    // blocks of one class are contiguous in the param array (class order 1,2,3), each class is one launch
    int cls_first[4] = {0, 0, 0, 0};
    int pos = 0;
    for (int c = 1; c <= 3; ++c) {
        cls_first[c] = pos;
        for (int b = 0; b < nblocks; ++b) {
            if (cls[b] != c) continue;
            EncBlk &e = hb[pos];
            const u64 n = h_in_n[b];
            e.in = d_in + h_in_off[b];
            e.out = d_out + h_out_off[b];
            e.n = n;
            e.out_cap = h_out_cap[b];
            e.out_n = d_out_n + b;
            e.err = bt->d_err + b;
            e.desc_base = dbase;
            e.n_tiles = (u32)ceil_div_u64(n, tile_syms[c]);
            e.ticket = (u32)pos;
            e.pad = 0;
            e.thist = (c == 1 && tiles) ? d_thist + h_thist_off[b] : nullptr;
            dbase += e.n_tiles + (c == 1 && tiles ? 1 : 0);
            e.lut = dpar + (o_tab - o_blk) + tpos;
            const shafa_code_table &t = h_tables[b];
            if (c == 1 && one_pass) {                 // {code, len}; a symbol without a code: len = 1 << 16
                u64 *l = (u64 *)(htab + tpos);
                for (int s = 0; s < 256; ++s) l[s] = t.len[s] ? ((u64)code_value(t, s) | ((u64)t.len[s] << 32)) : (1ull << 48);
                tpos += 2048;
            } else if (c == 1) {                      // code | len << 16; bit 31: no code
                u32 *l = (u32 *)(htab + tpos);
                for (int s = 0; s < 256; ++s) l[s] = t.len[s] ? (code_value(t, s) | ((u32)t.len[s] << 16)) : 0x80000000u;
                tpos += 1024;
            } else if (c == 2 && one_pass2) {         // {code, len}; a symbol without a code: len = 1 << 16
                u64 *l = (u64 *)(htab + tpos);
                for (int s = 0; s < 256; ++s) l[s] = t.len[s] ? ((u64)code_value(t, s) | ((u64)t.len[s] << 32)) : (1ull << 48);
                tpos += 2048;
            } else if (c == 2) {                      // code | len << 32
                u64 *l = (u64 *)(htab + tpos);
                for (int s = 0; s < 256; ++s) l[s] = (u64)code_value(t, s) | ((u64)t.len[s] << 32);
                tpos += 2048;
            } else {
                memcpy(htab + tpos, &t, sizeof(t));
                tpos += (sizeof(shafa_code_table) + 15) & ~15ul;
            }
            ++pos;
        }
    }
    const bool need_desc = cls_count[3] || (cls_count[1] && one_pass && !tiles) || one_pass2;
    if (o_zero_end && need_desc) HIP_TRY(hipMemsetAsync(ws, 0, o_zero_end, st));
    if ((rc = batch_params_commit(bt, st, hs, stage_bytes))) return rc;
    for (int b = 0; b < nblocks; ++b)
        if (cls[b] == 0) HIP_TRY(hipMemsetAsync(d_out_n + b, 0, 8, st));

    const EncBlk *dblk = (const EncBlk *)dpar;
    u64 *ddesc = (u64 *)(ws + o_desc);
    u32 *dtick = (u32 *)(ws + o_tick);
    if (cls_count[1] && tiles) {                       // tile offsets from the histograms: u64 per entry in the descriptor area,
        bool any_ragged = false;                       //   the tile totals (u32) in the tile-bits area
        for (int b = 0; b < nblocks; ++b)
            if (cls[b] == 1 && (h_in_n[b] & 32767)) any_ragged = true;
        if ((rc = sfenc6_launch(st, dblk + cls_first[1], cls_count[1], max_tiles[1], lmax1, any_ragged, (u32 *)(ws + o_tbits), ddesc))) return rc;
    } else if (cls_count[1]) {
        if (one_pass) {
            u32 ragged = 0;                            // bit 0 / 1 / 2: a block has a remainder after its full 8 / 16 / 32 KiB tiles
            for (int b = 0; b < nblocks; ++b)
                if (cls[b] == 1) ragged |= ((h_in_n[b] & 8191) ? 1u : 0u) | ((h_in_n[b] & 16383) ? 2u : 0u) | ((h_in_n[b] & 32767) ? 4u : 0u);
            const SfeRedo x = {redo ? (u64 *)(ws + o_desc2) : nullptr, redo ? (u32 *)(ws + o_tick2) : nullptr,
                               redo ? (u32 *)(ws + o_redo) + cls_first[1] : nullptr};
            if ((rc = sfenc4_launch(st, dblk + cls_first[1], cls_count[1], ddesc, dtick, lmax1, ragged, x))) return rc;
        } else sfenc3_launch(st, dblk + cls_first[1], cls_count[1], max_tiles[1], (u32 *)(ws + o_tbits), ddesc, false);
    }
    if (cls_count[2] && one_pass2) {
        u32 ragged = 0;
        for (int b = 0; b < nblocks; ++b)
            if (cls[b] == 2) ragged |= ((h_in_n[b] & 8191) ? 1u : 0u) | ((h_in_n[b] & 32767) ? 4u : 0u);
        const SfeRedo x = {(u64 *)(ws + o_desc2), (u32 *)(ws + o_tick2), (u32 *)(ws + o_redo) + cls_first[2]};
        if ((rc = sfenc4_launch_long(st, dblk + cls_first[2], cls_count[2], ddesc, dtick, lmax2, ragged, x))) return rc;
    } else if (cls_count[2]) sfenc3_launch(st, dblk + cls_first[2], cls_count[2], max_tiles[2], (u32 *)(ws + o_tbits), ddesc, true);
    if (cls_count[3])
        hipLaunchKernelGGL(sf_encode_generic, dim3(max_tiles[3] * cls_count[3]), dim3(ENC_THREADS), 0, st,
                           dblk + cls_first[3], cls_count[3], ddesc, dtick);
    HIP_TRY(hipGetLastError());
    return pscope.done();
}
