// rle_decode.hip — Module D hot path: RLE expansion (reference d.c:116-197 rle_block_decompressor).
//
// A 0 byte may be an escape, a symbol or a count, so token boundaries are found with a 3-state machine
//     S0 (token start): b == 0 -> S1, else literal -> S0      S1 (symbol) -> S2      S2 (count) -> S0
// whose per-byte transition maps {S0,S1,S2}->{S0,S1,S2} are composed with an ordered scan (6 bits per
// map): lanes -> waves -> tiles (look-back over tile maps; a map that sends every state to the same
// state ends the walk, which almost every tile's does).  Token output sizes ({0,s,c}: c, or 1 when
// c == 0, d.c:179-184; literal: 1) are scanned for output offsets (second look-back, 64-bit), and the
// output is filled in coalesced 16-byte pieces by binary-searching the tile's token table in LDS.
//
// Algorithmic HBM bytes per block: rle_n read + orig_n written.
#include "common.hpp"
#include "internal.hpp"

namespace {

constexpr int RLD_THREADS = 256;
constexpr int RLD_TILE = RLD_THREADS * 16;
constexpr u32 FN_IDENT = 0u | (1u << 2) | (2u << 4);

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
        u32 spins = 0;
        for (;;) {
            d = (idx >= 0) ? desc_load(desc + idx) : (DESC_PREFIX << 62);
            if (__all((d >> 62) != DESC_EMPTY)) break;
            if (++spins > SPIN_LIMIT) {
                if (lane == 0) set_error(err, SHAFA_DEVICE_ERROR);
                if ((d >> 62) == DESC_EMPTY) d = (DESC_PREFIX << 62);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
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
    u32 tok_off[RLD_TILE + 4];     // output offset (tile-local) of token t; [ntok] = tile total
    u8 tok_sym[RLD_TILE + 16];
    u32 wfn[4];
    u32 wcnt[4];
    u32 wlen[4];
    u32 tile;
    u32 state_in;
    u64 O;
};

__global__ __launch_bounds__(RLD_THREADS) void rle_decode_kernel(const RldBlk *__restrict__ blks, int nblk,
                                                                 u64 *desc_state, u64 *desc_sum, u32 *tickets)
{
    __shared__ __attribute__((aligned(16))) RldShared sh;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int b = blockIdx.x % nblk;
    const RldBlk blk = blks[b];
    if ((u32)(blockIdx.x / nblk) >= blk.n_tiles) return;
    if (tid == 0) sh.tile = atomicAdd(tickets + blk.ticket, 1u);
    __syncthreads();
    const int k = (int)sh.tile;
    const u64 n = blk.n;
    const u64 pos = (u64)k * RLD_TILE + (u64)tid * 16;
    u64 *dst = desc_state + blk.desc_base, *dsum = desc_sum + blk.desc_base;

    // ---- load 16 bytes (+2 of look-ahead for a triple that starts at byte 14/15) --------------------------
    u32 x[18];
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
        for (int j = 0; j < 16; ++j) x[j] = (j < nvalid) ? (u32)blk.in[pos + j] : 0u;
    }
    x[16] = (pos + 16 < n) ? (u32)blk.in[pos + 16] : 0u;
    x[17] = (pos + 17 < n) ? (u32)blk.in[pos + 17] : 0u;

    // ---- this thread's transition map, then ordered scan over lanes and waves ----------------------------
    u32 t0 = 0, t1 = 1, t2 = 2;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < nvalid) { t0 = step(t0, x[j]); t1 = step(t1, x[j]); t2 = step(t2, x[j]); }
    }
    u32 f = t0 | (t1 << 2) | (t2 << 4);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 y = (u32)__shfl_up((int)f, d, 64);
        if (lane >= d) f = fn_compose(y, f);
    }
    if (lane == 63) sh.wfn[wv] = f;
    u32 fex = (u32)__shfl_up((int)f, 1, 64);
    if (lane == 0) fex = FN_IDENT;
    __syncthreads();
    u32 wcar = FN_IDENT, ftile = FN_IDENT;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wv) wcar = fn_compose(wcar, sh.wfn[w]);
        ftile = fn_compose(ftile, sh.wfn[w]);
    }
    const u32 fpre = fn_compose(wcar, fex);     // map of all bytes of the tile before this thread

    if (wv == 0) {
        u32 sin = 0;
        if (k > 0) {
            if (tid == 0) {
                if (fn_const(ftile)) desc_store(dst + k, DESC_PREFIX, ftile & 3);
                else desc_store(dst + k, DESC_AGG, ftile);
            }
            sin = lookback_state(dst, k, blk.err);
        }
        if (tid == 0) {
            desc_store(dst + k, DESC_PREFIX, fn_apply(ftile, sin));
            sh.state_in = sin;
        }
    }
    __syncthreads();

    // ---- tokens of this thread ---------------------------------------------------------------------------------
    u32 s = fn_apply(fpre, sh.state_in);
    u32 tlen[16];
    u32 cnt = 0, len = 0;
    bool trunc = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        u32 l = 0;
        if (j < nvalid) {
            if (s == 0) {
                if (x[j] == 0) {
                    if (pos + j + 2 >= n) trunc = true;          // triple cut by the block end
                    else l = x[j + 2] ? x[j + 2] : 1u;           // count 0 behaves as one literal (d.c:179-184)
                } else l = 1;
                cnt += 1;
            }
            s = step(s, x[j]);
        }
        tlen[j] = l;
        len += l;
    }
    // (a token start with l == 0 only happens for a truncated triple)
    if (trunc) set_error(blk.err, SHAFA_FILE_UNRECOGNIZABLE);

    // ---- scan token counts and output lengths ----------------------------------------------------------------------
    const u32 icnt = wave_incl_scan_add<u32>(cnt), ilen = wave_incl_scan_add<u32>(len);
    if (lane == 63) { sh.wcnt[wv] = icnt; sh.wlen[wv] = ilen; }
    __syncthreads();
    u32 cbase = 0, lbase = 0, ctot = 0, ltot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w == wv) { cbase = ctot + icnt - cnt; lbase = ltot + ilen - len; }
        ctot += sh.wcnt[w];
        ltot += sh.wlen[w];
    }
    // token table
    {
        u32 s2 = fn_apply(fpre, sh.state_in), ti = cbase, lo = lbase;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < nvalid) {
                if (s2 == 0) {
                    sh.tok_off[ti] = lo;
                    sh.tok_sym[ti] = (u8)(x[j] == 0 ? x[j + 1] : x[j]);
                    ++ti;
                    lo += tlen[j];
                }
                s2 = step(s2, x[j]);
            }
        }
    }
    if (tid == 0) sh.tok_off[ctot] = ltot;

    if (wv == 0) {
        u64 O = 0;
        if (k > 0) {
            if (tid == 0) desc_store(dsum + k, DESC_AGG, ltot);
            O = lookback_sum(dsum, k, blk.err);
        }
        if (tid == 0) {
            desc_store(dsum + k, DESC_PREFIX, O + ltot);
            sh.O = O;
        }
    }
    __syncthreads();

    const u64 O = sh.O, Oend = O + ltot;
    const u64 limit = blk.out_cap < (u64)SHAFA_RLE_DECODE_MAX ? blk.out_cap : (u64)SHAFA_RLE_DECODE_MAX;
    if (Oend > (u64)SHAFA_RLE_DECODE_MAX) { if (tid == 0) set_error(blk.err, SHAFA_FILE_UNRECOGNIZABLE); }
    else if (Oend > blk.out_cap) { if (tid == 0) set_error(blk.err, SHAFA_LACK_OF_MEMORY); }
    if (k == (int)blk.n_tiles - 1 && tid == 0) *blk.out_n = Oend;

    // ---- fill: 16-byte pieces aligned in the output, token found by binary search --------------------------------
    if (ltot && O < limit) {
        const u64 wend = Oend < limit ? Oend : limit;
        const u64 c0 = O >> 4, c1 = (wend + 15) >> 4;
        const int ntok = (int)ctot;
        for (u64 c = c0 + tid; c < c1; c += RLD_THREADS) {
            const u64 g0 = c << 4;
            const u64 lo_g = g0 < O ? O : g0, hi_g = (g0 + 16 < wend) ? g0 + 16 : wend;
            const u32 lstart = (u32)(lo_g - O);
            // last token with tok_off <= lstart
            int a = 0, z = ntok - 1;
            while (a < z) {
                const int mid = (a + z + 1) >> 1;
                if (sh.tok_off[mid] <= lstart) a = mid; else z = mid - 1;
            }
            u32 wds[4] = {0, 0, 0, 0};
            int t = a;
            u32 nxt = sh.tok_off[t + 1];
            u32 sym = sh.tok_sym[t];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const u64 gp = g0 + q;
                if (gp >= lo_g && gp < hi_g) {
                    const u32 lp = (u32)(gp - O);
                    while (lp >= nxt) { ++t; nxt = sh.tok_off[t + 1]; sym = sh.tok_sym[t]; }
                    wds[q >> 2] |= sym << (8 * (q & 3));
                }
            }
            if (lo_g == g0 && hi_g == g0 + 16) {
                *(uint4 *)(blk.out + g0) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (g0 + q >= lo_g && g0 + q < hi_g) blk.out[g0 + q] = (u8)(wds[q >> 2] >> (8 * (q & 3)));
            }
        }
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
    const size_t o_state = off; off += ndesc * 8;
    const size_t o_sum = off; off += ndesc * 8;
    const size_t o_tick = off; off += (size_t)nblocks * 4; off = (off + 15) & ~(size_t)15;
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
    HIP_TRY(hipMemcpyAsync(ws + o_blk, hb, (size_t)nblocks * sizeof(RldBlk), hipMemcpyHostToDevice, st));
    if (max_tiles) {
        hipLaunchKernelGGL(rle_decode_kernel, dim3(max_tiles * (u32)nblocks), dim3(RLD_THREADS), 0, st,
                           (const RldBlk *)(ws + o_blk), nblocks, (u64 *)(ws + o_state), (u64 *)(ws + o_sum),
                           (u32 *)(ws + o_tick));
        HIP_TRY(hipGetLastError());
    }
    return SHAFA_SUCCESS;
}
