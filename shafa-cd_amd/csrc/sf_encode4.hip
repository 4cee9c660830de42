// sf_encode4.hip — Shannon-Fano bit-pack encoder for codes <= 32 bits in ONE pass over the input, for callers WITHOUT
// tile histograms (compress_to_buffer + binary_coding, reference c.c:52-237): n bytes read + ceil(bits / 8) bytes written,
// which is the kernel's algorithmic HBM traffic.  (With Module F's tile histograms the tile offsets
// exist before the launch and the encoder is a one-shot grid: sf_encode6.hip.)
//
// Why one pass: the three-kernel form (sf_encode3.hip: count, scan, pack) reads the input twice, 2.67 n of traffic on
// Zipf data, and sits at the HBM ceiling of that traffic.  The prefix problem (where does tile t start in the output?) is
// solved here with a chained scan whose waiting never reaches the critical path (sfe5_kernel):
//
//   * workgroups are persistent and stick to one block at a time (block = blockIdx % nconc, then + nconc): the
//     block's look-up table is loaded once, and with many blocks per launch every block is a separate chain with
//     only a handful of tiles in flight, so a look-back is one 64-entry window;
//   * tiles of a block are handed out by a per-block ticket (atomicAdd), requested three iterations ahead: a tile's
//     predecessors were always taken by workgroups that are running, so the chain cannot deadlock whatever part
//     of the grid is resident (other kernels on other streams included);
//   * a three-stage software pipeline, one LDS window per stage: iteration i looks up, groups and scans tile i, publishes
//     its bit total (aggregate) and places its bit strings in a window at TILE-LOCAL offsets (plain stores with carries,
//     sfe_dev.hpp: emit_oct); the descriptor window of the tile is requested in iteration i+1 (its predecessors hold
//     earlier tickets and have had a whole tile time to publish) and consumed in iteration i+2 (a whole tile time to
//     arrive), when the tile's window is funnel-shifted by (B mod 32) on its way out: byte-swapped, 16-byte aligned
//     non-temporal stores.  Wave 0 runs the chain and consumes every chain value BEFORE it issues new memory operations
//     (s_waitcnt vmcnt counts in order); one barrier per tile (the hand-over words exist twice, by iteration parity).
//   * A tile owns the output dwords [B >> 5, E >> 5); the B mod 32 leading bits of its first dword are re-encoded from the
//     up to 31 symbols before the tile into the "lead word" in front of the window, so every output dword is written
//     exactly once: no global atomics, no pre-zeroed output.  A block's ragged remainder is sfe4_tail_kernel's.
//   * 1024 lanes and 32 KiB tiles where three windows fit a CU's LDS (codes <= 12 bits; 13..16-bit and, as quads, 17..32-bit
//     codes with windows sized for 12 bits per symbol and a flag-and-encode-again pass for tiles that do not fit).
#include "common.hpp"
#include "internal.hpp"
#include "sfe_dev.hpp"

#include <mutex>

int g_sfe4_wide = 1;                                   // 0: always the 256-lane form (A/B and tests)
int g_sfe_window_bits = 0;                             // test knob: bits per symbol of the wide form's windows (0: min(Lmax, 12))
int g_sfe_lanes = 0;                                   // 0: widest form that fits; 256 / 512: that workgroup width (A/B and tests)

namespace {

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
        // tickets are four deep so that no atomic's round trip is ever waited for: `cur` is processed, `nxt` is being
        // loaded, `nn` is known, and thread 0 holds the request issued one iteration ago (req)
        u32 cur = sh.tick, nxt = cur + 1, nn = cur + 2;
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

            // ---- wave 0, before it issues anything new: everything it consumes here was requested at least half an
            //      iteration ago (it takes no part in the window stores below), so these waits are short ---------------------
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
                        if (r && p_T <= cap_bits) lead_bits(sh.lut, pwin, pv_w, r, lane);     // (a tile that did not fit its window was not placed)
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
                if (absent) set_error_over(bp->err, SHAFA_FILE_UNRECOGNIZABLE, SHAFA_LACK_OF_MEMORY);   // data symbol without a code (output undefined, in bounds)
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
            // (a tile that did not fit its window was never placed and its block is encoded again: nothing to store, and
            // reading p_T bits of a window that holds cap_bits would run past the workgroup's LDS)
            if (wv >= E5_STORE_W0 && have_p && p_T <= cap_bits)
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

// the quad form's launcher (codes of 17..32 bits): as e4_launch_nt, windows of lmax_win bits per symbol
template <int NT>
int e5q_launch_nt(hipStream_t st, const EncBlk *dblk, int count, u64 *d_desc, u32 *d_tickets, u32 lmax_win, u32 lmax, bool any_ragged,
                  u32 *d_redo, int redo_only)
{
    constexpr int MAXDEV = 64;
    static int wgs_by_dev_lmax[MAXDEV][33], cus_by_dev[MAXDEV], tail_attr_by_dev[MAXDEV], main_attr_by_dev[MAXDEV];
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
        // the attribute only ever grows: a later launch with smaller windows must not lower the limit under an earlier,
        // larger window size whose occupancy is cached
        if (dyn > 65536 && (int)dyn > main_attr_by_dev[dev]) {
            HIP_TRY(hipFuncSetAttribute((const void *)sfe5_kernel<5, false, NT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            main_attr_by_dev[dev] = (int)dyn;
        }
        if (!wgs_by_lmax[lmax_win]) {
            int occ = 0;
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, dev));
            cus_by_dev[dev] = prop.multiProcessorCount;
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
    HIP_TRY(hipGetLastError());
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
    static int wgs_by_dev_lmax[MAXDEV][17], cus_by_dev[MAXDEV], tail_attr_by_dev[MAXDEV], main_attr_by_dev[MAXDEV];
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
        // more than the default 64 KiB of dynamic LDS per workgroup: the attribute only ever grows (a launch with smaller
        // windows — sf_encode_window_bits, a launch of shorter codes — must not lower it under a cached larger size)
        if (dyn > 65536 && (int)dyn > main_attr_by_dev[dev]) {
            HIP_TRY(hipFuncSetAttribute((const void *)sfe5_kernel<NW, L16, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            main_attr_by_dev[dev] = (int)dyn;
        }
        if (!wgs_by_lmax[lmax_win]) {
            int occ = 0;
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, dev));
            cus_by_dev[dev] = prop.multiProcessorCount;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)sfe5_kernel<NW, L16, NT>, NT, dyn));
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
    hipLaunchKernelGGL((sfe5_kernel<NW, L16, NT>), dim3((u32)(nconc * per)), dim3(NT), dyn, st, dblk, count, nconc, d_desc, d_tickets,
                       win_stride, cap_bits, d_redo, redo_only);
    if (any_ragged)
        hipLaunchKernelGGL((sfe4_tail_kernel<NW, L16, NT>), dim3((u32)count), dim3(NT), (size_t)tail_stride * 4, st, dblk,
                           (const u64 *)d_desc, tail_stride, (const u32 *)(redo_only ? d_redo : nullptr), (const u64 *)nullptr);
    HIP_TRY(hipGetLastError());
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
    if (g_sfe_lanes == 512)
        return e4_launch_nt<NW, L16, 512>(st, dblk, count, d_desc, d_tickets, lmax, lmax, ragged16, nullptr, 0);
    if (g_sfe_lanes == 256 || !g_sfe4_wide)
        return e4_launch_nt<NW, L16, 256>(st, dblk, count, d_desc, d_tickets, lmax, lmax, ragged8, nullptr, 0);
    const u32 wbits = e5_window_bits(lmax);
    if (wbits == lmax) return e4_launch_nt<NW, L16, 1024>(st, dblk, count, d_desc, d_tickets, lmax, lmax, ragged32, nullptr, 0);
    if (!x.redo)                                       // no second chain at hand: the 256-lane form holds the worst case
        return e4_launch_nt<NW, L16, 256>(st, dblk, count, d_desc, d_tickets, lmax, lmax, ragged8, nullptr, 0);
    int rc = e4_launch_nt<NW, L16, 1024>(st, dblk, count, d_desc, d_tickets, wbits, lmax, ragged32, x.redo, 0);
    if (rc) return rc;
    return e4_launch_nt<NW, L16, 256>(st, dblk, count, x.desc2, x.tickets2, lmax, lmax, ragged8, x.redo, 1);
}

}  // namespace

// launched from sfenc_launch (sf_encode.hip) for blocks whose codes are <= 16 bits when the launch holds enough blocks
// to keep every chain short; desc (one u64 per tile) and tickets (one u32 per block) are zeroed by the caller;
// tables are 256 x u64 {code, len}
// does a launch with this longest code run the wide form with windows smaller than its worst case (then it needs `x`)?
bool sfenc4_needs_redo(u32 lmax)
{
    return g_sfe4_wide && g_sfe_lanes == 0 && e5_window_bits(lmax) < lmax;
}

// can launches of codes of 17..32 bits take the one-pass encoder (quad form)?  It always runs with windows smaller than
// its worst case, so it needs the second chain.
bool sfenc4_long_ok() { return g_sfe4_wide && g_sfe_lanes == 0; }

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
