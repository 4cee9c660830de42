// sf_encode2.hip — persistent, software-pipelined Shannon-Fano bit-pack encoder (codes <= 16 bits).
//
// Same algorithm and output as sf_encode_fast<4> (sf_encode.hip); what changes is the schedule.
// Profiling showed the one-tile-per-workgroup version latency-bound (65 % of wave cycles waiting on
// the chain  block record -> ticket -> input load -> look-back  at 3 workgroups per CU).  Here a
// workgroup is persistent on "its" block and runs a three-stage pipeline over the tiles it pulls:
//
//   iteration i:   [input of tile i was requested one iteration ago, its ticket two iterations ago]
//     1. look up + group tile i, scan the bit totals                                     (barrier A)
//     2. waves 1-3: store tile i-1 (its bit-stream is still staged in LDS, prefix known),
//                   then request the input of tile i+1
//        wave 0   : publish tile i's aggregate, look back for its prefix (the poll round trip hides
//                   behind the stores), request input i+1 and ticket i+2                 (barrier B)
//     3. lead bits, zero the LDS window (barrier C), merge tile i's groups into it
//
// so neither the ticket, nor the HBM load, nor the look-back poll, nor the store acknowledgements sit
// on a workgroup's critical path (memory operations retire in issue order per wave, hence the
// stores-before-loads order in step 2).  Other differences from the first version: the block's LUT
// stays in LDS across tiles; a lane's groups reach the LDS bit-stream through a 128-bit register
// accumulator as whole 64-bit words (plain stores) plus one atomic OR per item edge, instead of three
// atomic ORs per group; wave scans are DPP adds; absent-symbol detection is one OR per symbol.
#include "common.hpp"
#include "internal.hpp"

#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int E2_THREADS = 256;
constexpr int E2_ITEMS = 2;                            // 16-byte items per lane and tile
constexpr int E2_TILE = E2_THREADS * 16 * E2_ITEMS;    // 8 KiB of symbols
constexpr int E2_SW64 = 1280;                          // LDS bit-stream window: 1280 x 64 bit = 10 KiB
constexpr u32 E2_SW32 = 2 * E2_SW64;

struct Enc2Shared {
    u64 stage[E2_SW64 + 2];      // [0] = {-, lead/carry word}; [1..] = window
    u32 lut[256];
    u32 wtot[4 * E2_ITEMS];
    u32 prev[2][8];              // the 32 input bytes before the tile (for the lead bits), 2 tiles
    u32 next_tile;
    u32 pad;
    u64 prefix;
};

// wave-wide inclusive add scan with DPP row shifts / broadcasts (gfx9 family)
__device__ __forceinline__ u32 wave_scan_dpp(u32 v)
{
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return v;
}

// local 32-bit word j of the staged stream (j = -1 is the lead/carry word); u64 words are stored
// little-endian in LDS, so the high half (first 32 stream bits) is the odd dword
__device__ __forceinline__ u32 stage_word(const u32 *st32, int j) { return st32[2 + (j ^ 1)]; }

// what is needed to store a tile whose bit-stream sits in the LDS window
struct Pending {
    u64 B;          // global bit offset of the first staged local word (tile offset, or later window)
    u32 bits;       // bits from B to the tile's end
    u32 valid;      // 0 = nothing staged
    u32 last;       // last tile of the block
};

// store the local words [0, SW32) of the staged window: funnel-shift by (B mod 32), big-endian, coalesced.
// A lane handles the two output words that start in staged u64 word i (needs the low half of word i-1).
__device__ __forceinline__ void store_window(const u64 *stage, const EncBlk &blk, const Pending &pd, int t0,
                                             int nthreads, u32 dbg = 0)
{
    const u64 E = pd.B + pd.bits;
    const u32 s = (u32)pd.B & 31;
    const u64 wg_lo = pd.B >> 5;
    const u64 wg_hi = pd.last ? ((E + 31) >> 5) : (E >> 5);
    const u32 OW = (u32)(wg_hi - wg_lo);
    const u64 total_bytes = (E + 7) >> 3;
    // words [0, jfull) are stored whole; the block's final word may be partial (1..3 bytes)
    u32 jend = OW < E2_SW32 ? OW : E2_SW32;
    const bool tail = pd.last && OW <= E2_SW32 && OW > 0 && total_bytes < 4 * wg_hi;
    u32 jfull = tail ? jend - 1 : jend;
    const u64 cap_words = blk.out_cap >> 2;
    if (wg_lo + jfull > cap_words) {                   // does not fit the caller's buffer
        if (t0 == 0) set_error(blk.err, SHAFA_LACK_OF_MEMORY);
        jfull = cap_words > wg_lo ? (u32)(cap_words - wg_lo) : 0u;
    }
    u32 *out32 = (u32 *)blk.out + wg_lo;
    for (u32 i = (u32)t0; 2 * i < jfull; i += (u32)nthreads) {
        const u64 cw = stage[1 + i], pw = stage[i];        // staged words i and i-1 ([0] holds the lead word)
        const u32 hi = (u32)(cw >> 32), lo = (u32)cw;
        const u32 w0 = bswap32(funnel_r((u32)pw, hi, s)), w1 = bswap32(funnel_r(hi, lo, s));
        if (dbg & 8) { if (w0 == 0x12345678u) gstore<u32>(out32 + 2 * i, w1); continue; }
        gstore<u32>(out32 + 2 * i, w0);
        if (2 * i + 1 < jfull) gstore<u32>(out32 + 2 * i + 1, w1);
    }
    if (tail && t0 == 0) {
        const u32 j = jend - 1;
        const u64 cw = stage[1 + (j >> 1)], pw = stage[j >> 1];
        const u32 a = (j & 1) ? (u32)(cw >> 32) : (u32)pw, c = (j & 1) ? (u32)cw : (u32)(cw >> 32);
        const u32 w = funnel_r(a, c, s);
        const u64 W = wg_lo + j;
        const u32 nb = (u32)(total_bytes - 4 * W);
        if (total_bytes <= blk.out_cap) {
            for (u32 q = 0; q < nb; ++q) gstore<u8>(blk.out + 4 * W + q, (u8)(w >> (24 - 8 * q)));
        } else set_error(blk.err, SHAFA_LACK_OF_MEMORY);
    }
}

// PROF: thread 0 of every workgroup accumulates s_memtime deltas per pipeline phase (diagnostic
// build only: never on in a timed or shipped run; the stamps go to a buffer nothing else reads)
#define STAMP(i)                                                         \
    do {                                                                 \
        if (PROF && tid == 0) {                                          \
            const u64 t_ = __builtin_amdgcn_s_memtime();                 \
            acc_[i] += t_ - t0_;                                         \
            t0_ = t_;                                                    \
        }                                                                \
    } while (0)

template <bool PROF>
__global__ __launch_bounds__(E2_THREADS, 4) void sf_encode_persistent(const EncBlk *__restrict__ blks, int nblk,
                                                                      u64 *desc, u32 *tickets, u64 *prof, u32 dbg)
{
    u64 acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64 t0_ = PROF ? __builtin_amdgcn_s_memtime() : 0;
    __shared__ __attribute__((aligned(16))) Enc2Shared sh;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    u32 *st32 = (u32 *)sh.stage;

    for (int b = blockIdx.x % nblk; b < nblk; b += gridDim.x) {
        const EncBlk blk = blks[b];
        if (blk.n_tiles == 0) continue;
        u64 *bdesc = desc + blk.desc_base;
        u32 *ctr = tickets + blk.ticket;
        __syncthreads();                               // previous block: LUT and stage no longer in use
        sh.lut[tid] = gload<u32>((const u32 *)blk.lut + tid);
        if (tid == 0) sh.next_tile = atomicAdd(ctr, 1u);
        __syncthreads();
        u32 k = sh.next_tile;

        uint4 cur[E2_ITEMS];
        u32 pv = 0;                                    // lanes 0..7: the 32 input bytes before the tile
        auto issue_loads = [&](u32 tile) {
            if (tid < 8 && tile > 0) pv = gload<u32>(blk.in + (u64)tile * E2_TILE - 32 + 4 * tid);
            if (((u64)tile + 1) * E2_TILE <= blk.n) {          // whole tile inside the block: plain loads
#pragma unroll
                for (int it = 0; it < E2_ITEMS; ++it)
                    cur[it] = gload<uint4>(blk.in + (u64)tile * E2_TILE + (u64)it * (E2_THREADS * 16) + (u64)tid * 16);
            } else {                                            // the block's ragged last tile
#pragma unroll
                for (int it = 0; it < E2_ITEMS; ++it) {
                    const u64 idx = (u64)tile * E2_TILE + (u64)it * (E2_THREADS * 16) + (u64)tid * 16;
                    u32 w[4] = {0, 0, 0, 0};
                    if (idx < blk.n) {
                        const int nv = blk.n - idx >= 16 ? 16 : (int)(blk.n - idx);
#pragma unroll
                        for (int q = 0; q < 16; ++q)
                            if (q < nv) w[q >> 2] |= (u32)gload<u8>(blk.in + idx + q) << (8 * (q & 3));
                    }
                    cur[it] = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        };
        // lead bits of a tile whose global bit offset is B: the last (B mod 32) bits of the stream
        // before it, re-encoded from the 32 input bytes in sh.prev[slot]
        auto lead_word = [&](u64 B, int slot) -> u32 {
            const u32 need = (u32)B & 31;
            if (!need) return 0u;
            u64 acc = 0;
            u32 got = 0;
            const u8 *pb = (const u8 *)sh.prev[slot];
            for (int p = 31; p >= 0 && got < need; --p) {       // >= 1 bit per symbol: 31 symbols suffice
                const u32 x = sh.lut[pb[p]];
                acc |= (u64)(x & 0xFFFFu) << got;
                got += (x >> 16) & 31u;
            }
            return (u32)acc & ((1u << need) - 1u);
        };
        if (k < blk.n_tiles) issue_loads(k);
        u32 tk = 0;                                    // thread 0: ticket requested for the tile after next
        if (tid == 0) tk = atomicAdd(ctr, 1u);
        Pending pd = {0, 0, 0, 0};                     // the tile staged in LDS, prefix not yet resolved
        u32 pd_k = 0;
        int slot = 0;
        STAMP(0);                                      // block setup

        while (k < blk.n_tiles) {
            const u64 tile_start = (u64)k * E2_TILE;
            if (tid < 8) sh.prev[slot][tid] = pv;
            // wave 0: the staged tile's predecessors, requested now, looked at after the look-ups
            u64 dsc = 0;
            if (wv == 0 && pd.valid && pd_k > 0) {
                const int idx = (int)pd_k - 1 - lane;
                dsc = idx >= 0 ? desc_load(bdesc + idx) : (DESC_PREFIX << 62);
            }
            if (PROF && tid == 0) acc_[7] += 1;

            // ---- 1. look up + group (4 symbols -> one group of <= 64 bits), scan ------------------------
            u64 grp[E2_ITEMS][4];
            u32 glen[E2_ITEMS];          // four lengths (<= 64) packed, 8 bits each
            u32 itot[E2_ITEMS];
            u32 flags = 0;
#pragma unroll
            for (int it = 0; it < E2_ITEMS; ++it) {
                const u64 idx = tile_start + (u64)it * (E2_THREADS * 16) + (u64)tid * 16;
                const u32 wds[4] = {cur[it].x, cur[it].y, cur[it].z, cur[it].w};
                const bool partial = idx + 16 > blk.n;          // only in a block's last tile
                u32 tot = 0, packed = 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32 e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const u32 sym = (wds[g] >> (8 * j)) & 0xFFu;
                        u32 x = sh.lut[sym];
                        if (partial && idx + 4 * g + j >= blk.n) x = 0;
                        flags |= x;
                        e[j] = x;
                    }
                    const u32 l0 = (e[0] >> 16) & 31u, l1 = (e[1] >> 16) & 31u;
                    const u32 l2 = (e[2] >> 16) & 31u, l3 = (e[3] >> 16) & 31u;
                    const u32 a = ((e[0] & 0xFFFFu) << l1) | (e[1] & 0xFFFFu);
                    const u32 c = ((e[2] & 0xFFFFu) << l3) | (e[3] & 0xFFFFu);
                    const u32 L = l0 + l1 + l2 + l3;
                    grp[it][g] = ((u64)a << (l2 + l3)) | c;
                    packed |= L << (8 * g);
                    tot += L;
                }
                glen[it] = packed;
                itot[it] = tot;
            }
            if (flags & 0x80000000u) set_error(blk.err, SHAFA_FILE_UNRECOGNIZABLE);

            u32 incl[E2_ITEMS];
#pragma unroll
            for (int it = 0; it < E2_ITEMS; ++it) incl[it] = wave_scan_dpp(itot[it]);
            if (lane == 63) {
#pragma unroll
                for (int it = 0; it < E2_ITEMS; ++it) sh.wtot[it * 4 + wv] = incl[it];
            }
            if (tid == 0) sh.next_tile = tk;            // the ticket requested one iteration ago
            STAMP(1);                                  // input wait + look-ups + wave scan
            __syncthreads();                                                                   // A
            STAMP(2);
            const u32 knext = sh.next_tile;
            u32 ioff[E2_ITEMS];
            u32 run = 0;
#pragma unroll
            for (int it = 0; it < E2_ITEMS; ++it) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (w == wv) ioff[it] = run + incl[it] - itot[it];
                    run += sh.wtot[it * 4 + w];
                }
            }
            const u32 tile_bits = run;
            const bool last = (k == blk.n_tiles - 1);
            const bool big = tile_bits + 64 > E2_SW32 * 32;      // expands past one LDS window (rare)

            // ---- 2. publish this tile's aggregate; resolve the staged tile's prefix (its predecessors
            //         published long ago, the descriptor load was issued before the look-ups) ------------
            if (wv == 0) {
                if (tid == 0) {
                    if (k > 0) desc_store(bdesc + k, DESC_AGG, tile_bits);
                    else desc_store(bdesc + k, DESC_PREFIX, tile_bits);
                }
                if (pd.valid) {
                    u64 Bp = 0;
                    if (pd_k > 0) Bp = lookback_sum(bdesc, (int)pd_k, blk.err, true, dsc);
                    if (tid == 0) {
                        if (pd_k > 0) desc_store(bdesc + pd_k, DESC_PREFIX, Bp + pd.bits);
                        sh.prefix = Bp;
                        st32[0] = lead_word(Bp, slot ^ 1);
                        if (pd.last) gstore<u64>(blk.out_n, (Bp + pd.bits + 7) >> 3);
                    }
                }
            }
            STAMP(3);                                  // wave 0: publish + resolve
            __syncthreads();                                                                   // B'
            // ---- 3. store the staged tile, then request the next input (stores retire first) ------------
            if (pd.valid) {
                pd.B = sh.prefix;
                store_window(sh.stage, blk, pd, tid, E2_THREADS, dbg);
                pd.valid = 0;
            }
            if (knext < blk.n_tiles && !((dbg & 16) && knext > 8)) issue_loads(knext);
            if (tid == 0) tk = atomicAdd(ctr, 1u);
            STAMP(4);                                  // store issue + load issue
            __syncthreads();                                                                   // B
            for (int i = tid; i < E2_SW64; i += E2_THREADS) sh.stage[1 + i] = 0;
            if (big) {                                  // synchronous path: this tile's prefix is needed now
                if (wv == 0) {
                    u64 B0 = 0;
                    if (k > 0) B0 = lookback_sum(bdesc, (int)k, blk.err);
                    if (tid == 0) {
                        if (k > 0) desc_store(bdesc + k, DESC_PREFIX, B0 + tile_bits);
                        sh.prefix = B0;
                        st32[0] = lead_word(B0, slot);
                        if (last) gstore<u64>(blk.out_n, (B0 + tile_bits + 7) >> 3);
                    }
                }
            }
            __syncthreads();                                                                   // C
            STAMP(5);                                  // zero + barriers

            // ---- 4. merge the groups into the window: per item a 128-bit accumulator --------------------
            pd.B = big ? sh.prefix : 0;
            pd.bits = tile_bits; pd.valid = 1; pd.last = last ? 1u : 0u;
            pd_k = k;
            for (u32 r0 = 0;; r0 += E2_SW32) {
                const u32 w64lo = r0 >> 1;
#pragma unroll
                for (int it = 0; it < E2_ITEMS; ++it) {
                    u32 q = ioff[it];
                    u32 w64 = q >> 6, fill = q & 63;
                    u64 hi = 0, lo = 0;
                    bool first = true;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const u32 L = (glen[it] >> (8 * g)) & 0xFFu;
                        if (L) {
                            const u64 G = grp[it][g];
                            const u32 sft = 128 - fill - L;                 // 1..127
                            const u64 up = G << ((sft - 64) & 63), dn = G >> ((64 - sft) & 63);
                            hi |= sft >= 64 ? up : dn;
                            lo |= sft >= 64 ? 0ull : (G << (sft & 63));
                            fill += L;
                        }
                        if (fill >= 64) {
                            const u32 i = w64 - w64lo;
                            if (i < (u32)E2_SW64) {
                                if (first) atomicOr((unsigned long long *)&sh.stage[1 + i], (unsigned long long)hi);
                                else sh.stage[1 + i] = hi;
                            }
                            hi = lo; lo = 0; fill -= 64; ++w64; first = false;
                        }
                    }
                    if (fill) {
                        const u32 i = w64 - w64lo;
                        if (i < (u32)E2_SW64) atomicOr((unsigned long long *)&sh.stage[1 + i], (unsigned long long)hi);
                    }
                }
                if (!big) break;                        // the common case: stored in the next iteration
                // big tile: store window after window right away
                __syncthreads();
                store_window(sh.stage, blk, pd, tid, E2_THREADS);
                const u64 E = pd.B + pd.bits;
                const u32 OWr = (u32)((pd.last ? ((E + 31) >> 5) : (E >> 5)) - (pd.B >> 5));
                __syncthreads();
                if (OWr <= E2_SW32) { pd.valid = 0; break; }
                u32 carry = 0;
                if (tid == 0) carry = stage_word(st32, (int)E2_SW32 - 1);
                __syncthreads();
                for (int i = tid; i < E2_SW64; i += E2_THREADS) sh.stage[1 + i] = 0;
                if (tid == 0) st32[0] = carry;
                __syncthreads();
                pd.B += (u64)E2_SW32 * 32;              // the rest of the tile, same bit phase
                pd.bits -= E2_SW32 * 32;
            }
            k = knext;
            slot ^= 1;
            STAMP(6);                                  // merge
        }
        // drain: the last tile of this workgroup is staged and unresolved
        __syncthreads();
        if (pd.valid) {
            if (wv == 0) {
                u64 Bp = 0;
                if (pd_k > 0) Bp = lookback_sum(bdesc, (int)pd_k, blk.err);
                if (tid == 0) {
                    if (pd_k > 0) desc_store(bdesc + pd_k, DESC_PREFIX, Bp + pd.bits);
                    sh.prefix = Bp;
                    st32[0] = lead_word(Bp, slot ^ 1);
                    if (pd.last) gstore<u64>(blk.out_n, (Bp + pd.bits + 7) >> 3);
                }
            }
            __syncthreads();
            pd.B = sh.prefix;
            store_window(sh.stage, blk, pd, tid, E2_THREADS);
        }
    }
    if (PROF && tid == 0) {
        for (int i = 0; i < 8; ++i) prof[(size_t)blockIdx.x * 8 + i] = acc_[i];
    }
}

}  // namespace

// launched from sfenc_launch (sf_encode.hip) for the Lmax <= 16 class
void sfenc2_launch(hipStream_t st, const EncBlk *dblk, int count, u32 total_tiles, u64 *ddesc, u32 *dtick)
{
    static int wgs = 0;
    if (!wgs) {
        hipDeviceProp_t prop;
        int dev = 0;
        (void)hipGetDevice(&dev);
        wgs = (hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256) * 4;
    }
    const u32 grid = total_tiles < (u32)wgs ? total_tiles : (u32)wgs;
    const char *dbg_env = getenv("SHAFA_ENC_DBG");
    const u32 dbg = dbg_env ? (u32)atoi(dbg_env) : 0u;
    if (getenv("SHAFA_ENC_PROF")) {          // diagnostic: per-phase cycle shares, printed to stderr
        u64 *dprof = nullptr;
        std::vector<u64> h((size_t)grid * 8);
        if (hipMalloc((void **)&dprof, h.size() * 8) != hipSuccess) return;
        hipLaunchKernelGGL(sf_encode_persistent<true>, dim3(grid), dim3(E2_THREADS), 0, st, dblk, count, ddesc, dtick, dprof, dbg);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), dprof, h.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(dprof);
        double sum[8] = {0};
        for (u32 g = 0; g < grid; ++g) for (int i = 0; i < 8; ++i) sum[i] += (double)h[(size_t)g * 8 + i];
        const char *nm[7] = {"setup", "input+lookup+scan", "barrierA", "publish+resolve", "store+load issue", "zero+barriers", "merge"};
        fprintf(stderr, "[enc prof] tiles/wg %.1f ; cycles per tile:", sum[7] / grid);
        for (int i = 0; i < 7; ++i) fprintf(stderr, " %s=%.0f", nm[i], sum[i] / (sum[7] > 0 ? sum[7] : 1));
        fprintf(stderr, "\n");
        return;
    }
    hipLaunchKernelGGL(sf_encode_persistent<false>, dim3(grid ? grid : 1), dim3(E2_THREADS), 0, st, dblk, count, ddesc,
                       dtick, (u64 *)nullptr, dbg);
}
