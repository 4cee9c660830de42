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
//     block's look-up table is loaded once, and with >= 32 blocks per launch every block is a separate chain with
//     only a handful of tiles in flight, so a look-back is one 64-entry window;
//   * tiles of a block are handed out by a per-block ticket (atomicAdd), requested two iterations ahead: a tile's
//     predecessors were always taken by workgroups that are running, so the chain cannot deadlock whatever part
//     of the grid is resident (other kernels on other streams included);
//   * iteration i looks up, groups and scans tile i, publishes its bit total (aggregate) and ORs its bit strings
//     into an LDS window at tile-local offsets; the tile's prefix is resolved in iteration i+1 from a descriptor
//     window that was requested at the top of that iteration, i.e. >= one tile time after every predecessor of
//     the chain published its aggregate, and the window (double buffered) is stored then.
//
// Per tile (8 KiB of symbols, 256 lanes x two 16-byte items): every lane turns its 16 symbols into two "octs"
// (8 symbols, <= 128 bits, right-aligned in four dwords) with a tree of shift-or steps on {code, len} pairs read
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

namespace {

constexpr int E4_THREADS = 256;
constexpr int E4_TILE = 8192;                          // symbols per tile
constexpr int E4_GUARD = 8;                            // dwords in front of the window: an oct writes up to 4 dwords before its last one
constexpr u32 E4_NONE = 0xFFFFFFFFu;

struct E4Static {
    u64 lut[256];                // {code (low dword), len (high dword)}; a symbol without a code: len = 1 << 16
    u32 wtot[8];                 // [item][wave] bit totals
    u64 prefix;                  // bits before the pending tile
    u32 tick;                    // next ticket of the block (broadcast)
    u32 pad;
};

struct Oct {                     // 8 symbols: right-aligned 128-bit string r3:r2:r1:r0, ll bits (bit 16+: a symbol had no code)
    u32 r0, r1, r2, r3, ll;
};

// four symbols of one input dword -> right-aligned group g of L bits (L <= 64)
template <bool SAFE>
__device__ __forceinline__ void quad(const u64 *lut, u32 w, u32 drop4, u64 &g, u32 &L)
{
    u64 e0 = lut[w & 0xFFu], e1 = lut[(w >> 8) & 0xFFu], e2 = lut[(w >> 16) & 0xFFu], e3 = lut[w >> 24];
    if (SAFE) {                                        // ragged last tile: symbols past the block encode as nothing
        if (drop4 & 1u) e0 = 0;
        if (drop4 & 2u) e1 = 0;
        if (drop4 & 4u) e2 = 0;
        if (drop4 & 8u) e3 = 0;
    }
    const u32 c0 = (u32)e0, c1 = (u32)e1, c2 = (u32)e2, c3 = (u32)e3;
    const u32 l0 = (u32)(e0 >> 32), l1 = (u32)(e1 >> 32), l2 = (u32)(e2 >> 32), l3 = (u32)(e3 >> 32);
    const u32 a = (c0 << (l1 & 31u)) | c1;             // <= 32 bits
    const u32 b = (c2 << (l3 & 31u)) | c3;
    const u32 lb = l2 + l3;
    g = ((u64)a << (lb & 63u)) | b;
    L = l0 + l1 + lb;
}

// SAFE: lengths may be 0 (dropped symbols); L16: a group may be exactly 64 bits (four 16-bit codes)
template <bool SAFE, bool L16, bool HAVE_R3>
__device__ __forceinline__ Oct make_oct(const u64 *lut, u32 w0, u32 w1, u32 drop8)
{
    u64 g0, g1;
    u32 L0, L1;
    quad<SAFE>(lut, w0, drop8 & 15u, g0, L0);
    quad<SAFE>(lut, w1, drop8 >> 4, g1, L1);
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
    uint4 v[2];
    u32 pv;                      // lanes 0..31 of wave 0: the symbol (tile start - 1 - lane)
};

__device__ __forceinline__ void load_tile(const EncBlk &blk, u32 tile, int tid, TileIn &t)
{
    const u64 base = (u64)tile * E4_TILE;
    if (base + E4_TILE <= blk.n) {
#pragma unroll
        for (int it = 0; it < 2; ++it) t.v[it] = gload_nt<uint4>(blk.in + base + (u64)it * 4096 + (u64)tid * 16);
    } else {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const u64 idx = base + (u64)it * 4096 + (u64)tid * 16;
            u32 w[4] = {0, 0, 0, 0};
            if (idx + 16 <= blk.n) {
                const uint4 x = gload<uint4>(blk.in + idx);
                w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w;
            } else if (idx < blk.n) {
                const int nv = (int)(blk.n - idx);
                for (int q = 0; q < nv; ++q) w[q >> 2] |= (u32)gload<u8>(blk.in + idx + q) << (8 * (q & 3));
            }
            t.v[it] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    t.pv = 0;
    if (tid < 32 && tile > 0) t.pv = gload<u8>(blk.in + base - 1 - (u64)tid);
}

// NW: dwords an oct can touch (3: Lmax <= 8, 4: <= 12, 5: <= 16); L16: Lmax == 16
//
// One iteration of a workgroup = tile `cur` is looked up, grouped, scanned and ORed into its LDS window at TILE-LOCAL bit
// offsets (the output position is not needed for that), while the tile of the previous iteration (`pending`), whose
// window is the other buffer, gets its prefix from the descriptors, is funnel-shifted by (B mod 32) on the way out of
// LDS and stored.  Two barriers per tile; the look-back runs in wave 0 beside the other waves' look-ups.
template <int NW, bool L16>
__global__ __launch_bounds__(E4_THREADS, 5) void sfe4_kernel(const EncBlk *__restrict__ blks, int nblk, int nconc,
                                                             u64 *__restrict__ desc, u32 *__restrict__ tickets, u32 win_stride)
{
    __shared__ E4Static sh;
    extern __shared__ __attribute__((aligned(16))) u32 dynwin[];     // two buffers of [E4_GUARD][window dwords]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool R3 = NW >= 5;

    for (int b = (int)(blockIdx.x % (u32)nconc); b < nblk; b += nconc) {
        const EncBlk blk = blks[b];
        u64 *bdesc = desc + blk.desc_base;
        __syncthreads();                               // the previous block's table and windows are no longer in use
        sh.lut[tid] = gload<u64>((const u64 *)blk.lut + tid);
        if (tid == 0) sh.tick = atomicAdd(tickets + blk.ticket, 2u);
        __syncthreads();
        u32 cur = sh.tick, nxt = cur + 1;
        if (cur >= blk.n_tiles) continue;
        TileIn cin, nin;
        load_tile(blk, cur, tid, cin);

        bool have_pend = false;
        u32 p_tile = 0, p_T = 0, p_pv = 0;
        u32 dirty[2] = {win_stride - E4_GUARD, win_stride - E4_GUARD};   // dwords of each buffer that may be non-zero
        u32 it_no = 0;

        for (;; ++it_no) {
            const bool cur_ok = cur < blk.n_tiles;
            if (!cur_ok && !have_pend) break;
            const bool nxt_ok = cur_ok && nxt < blk.n_tiles;
            const u32 buf = it_no & 1u;
            u32 *win = dynwin + buf * win_stride + E4_GUARD;               // this tile's window
            u32 *pwin = dynwin + (buf ^ 1u) * win_stride + E4_GUARD;       // the pending tile's window

            // ---- requests that have a whole iteration to come back ------------------------------------------
            u32 nn = E4_NONE;
            if (nxt_ok && tid == 0) nn = atomicAdd(tickets + blk.ticket, 1u);          // ticket after next
            u64 first = 0;
            if (have_pend && wv == 0 && p_tile > 0) {                                  // descriptor window of the pending tile
                const int idx = (int)p_tile - 1 - lane;
                first = idx >= 0 ? desc_load(bdesc + idx) : (DESC_PREFIX << 62);
            }
            if (nxt_ok) load_tile(blk, nxt, tid, nin);

            // ---- this tile: zero its window, look up, group, scan ------------------------------------------------
            Oct c_oct[2][2];
            u32 incl[2] = {0, 0}, itot[2] = {0, 0};
            if (cur_ok) {
                for (u32 i = (u32)tid; i < ((dirty[buf] + 3u) >> 2) + 1u; i += E4_THREADS)
                    ((uint4 *)win)[(int)i - 1] = make_uint4(0, 0, 0, 0);              // from dword -4: the lead word is win[-1]
                const u64 base = (u64)cur * E4_TILE;
                u32 absent = 0;
                if (base + E4_TILE <= blk.n) {
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        c_oct[it][0] = make_oct<false, L16, R3>(sh.lut, cin.v[it].x, cin.v[it].y, 0u);
                        c_oct[it][1] = make_oct<false, L16, R3>(sh.lut, cin.v[it].z, cin.v[it].w, 0u);
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const u64 idx = base + (u64)it * 4096 + (u64)tid * 16;
                        const u32 keep = idx >= blk.n ? 0u : (blk.n - idx >= 16 ? 16u : (u32)(blk.n - idx));
                        const u32 drop = (0xFFFFu << keep) & 0xFFFFu;
                        c_oct[it][0] = make_oct<true, L16, R3>(sh.lut, cin.v[it].x, cin.v[it].y, drop & 0xFFu);
                        c_oct[it][1] = make_oct<true, L16, R3>(sh.lut, cin.v[it].z, cin.v[it].w, drop >> 8);
                    }
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    absent |= (c_oct[it][0].ll | c_oct[it][1].ll) >> 16;
                    c_oct[it][0].ll &= 0xFFFFu;
                    c_oct[it][1].ll &= 0xFFFFu;
                    itot[it] = c_oct[it][0].ll + c_oct[it][1].ll;
                    incl[it] = dpp_scan_add(itot[it]);
                }
                if (absent) set_error(blk.err, SHAFA_FILE_UNRECOGNIZABLE);   // data symbol without a code (output undefined, in bounds)
                if (lane == 63) {
                    sh.wtot[wv] = incl[0];
                    sh.wtot[4 + wv] = incl[1];
                }
            }

            // ---- wave 0: the pending tile's prefix (its descriptor window was requested at the top) and lead bits ----
            if (have_pend && wv == 0) {
                u64 B = 0;
                if (p_tile > 0) {
                    B = lookback_sum(bdesc, (int)p_tile, blk.err, true, first);
                    if (lane == 0) {
                        desc_store(bdesc + p_tile, DESC_PREFIX, B + p_T);
                        if (p_tile == blk.n_tiles - 1) gstore<u64>(blk.out_n, (B + p_T + 7) >> 3);
                    }
                    const u32 r = (u32)B & 31u;
                    if (r) {                           // the last r bits before the tile, right-aligned in the lead word pwin[-1]
                        const u64 ent = lane < 32 ? sh.lut[p_pv & 0xFFu] : 0ull;
                        const u32 code = (u32)ent, len = (u32)(ent >> 32) & 0xFFFFu;
                        const u32 D = dpp_scan_add(len);               // bits from this symbol's first bit to the tile start
                        if (len && D - len < r)
                            __hip_atomic_fetch_or(pwin - 1, code << ((D - len) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if (lane == 0) sh.prefix = B;
            }
            if (tid == 0) sh.tick = nn;
            __syncthreads();                                                           // A
            nn = sh.tick;
            u32 c_T = 0;
            if (cur_ok) {
                u32 run = 0, ioff[2] = {0, 0};
#pragma unroll
                for (int it = 0; it < 2; ++it) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        if (w == wv) ioff[it] = run + incl[it] - itot[it];
                        run += sh.wtot[it * 4 + w];
                    }
                }
                c_T = run;
                if (tid == 0) {
                    if (cur == 0) {
                        desc_store(bdesc, DESC_PREFIX, c_T);
                        if (blk.n_tiles == 1) gstore<u64>(blk.out_n, ((u64)c_T + 7) >> 3);
                    } else desc_store(bdesc + cur, DESC_AGG, c_T);
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const u32 e0 = ioff[it] + c_oct[it][0].ll;
                    place<NW>(win, c_oct[it][0], e0);
                    place<NW>(win, c_oct[it][1], e0 + c_oct[it][1].ll);
                }
                dirty[buf] = (c_T >> 5) + 2u;
            }

            // ---- the pending tile leaves LDS: out dword j = bits [32 j - r, 32 j - r + 32) of its window ----------
            if (have_pend) {
                const u64 B = sh.prefix;
                const u32 r = (u32)B & 31u;
                const bool last = p_tile == blk.n_tiles - 1;
                const u64 E = B + p_T;
                const u64 total_bytes = (E + 7) >> 3;                                  // the block's size when this is its last tile
                const u64 gd0 = B >> 5;
                const u32 count = (u32)((last ? (total_bytes >> 2) : (E >> 5)) - gd0);  // owned dwords
                const u64 end_bytes = last ? total_bytes : 4 * (E >> 5);
                if (end_bytes > blk.out_cap) {
                    if (tid == 0) set_error(blk.err, SHAFA_LACK_OF_MEMORY);
                } else {
                    u8 *o = blk.out + 4 * gd0;                                         // owned dword 0
                    u32 h = (u32)(0 - gd0) & 3u;                                       // dwords up to the first 16-byte boundary
                    if (h > count) h = count;
                    const u32 nq = (count - h) >> 2;                                   // aligned 16-byte pieces
                    for (u32 q = (u32)tid; q < nq; q += E4_THREADS) {
                        const u32 j = h + 4 * q;
                        const u32 w0 = pwin[(int)j - 1], w1 = pwin[j], w2 = pwin[j + 1], w3 = pwin[j + 2], w4 = pwin[j + 3];
                        gstore_nt<uint4>(o + 4 * (u64)j, make_uint4(bswap32(__builtin_amdgcn_alignbit(w0, w1, r)),
                                                                   bswap32(__builtin_amdgcn_alignbit(w1, w2, r)),
                                                                   bswap32(__builtin_amdgcn_alignbit(w2, w3, r)),
                                                                   bswap32(__builtin_amdgcn_alignbit(w3, w4, r))));
                    }
                    const u32 t0 = h + 4 * nq;                                         // tail dwords [t0, count)
                    if ((u32)tid < 8) {
                        const u32 j = (u32)tid < 4 ? (u32)tid : t0 + (u32)tid - 4;
                        const bool ok = (u32)tid < 4 ? j < h : j < count;
                        if (ok) gstore<u32>(o + 4 * (u64)j, bswap32(__builtin_amdgcn_alignbit(pwin[(int)j - 1], pwin[j], r)));
                    } else if (last && (u32)tid < 11) {                                // the block's final 1..3 bytes
                        const u32 q = (u32)tid - 8;
                        const u32 w = __builtin_amdgcn_alignbit(pwin[(int)count - 1], pwin[count], r);
                        if (q < (u32)(total_bytes & 3)) gstore<u8>(o + 4 * (u64)count + q, (u8)(w >> (24 - 8 * q)));
                    }
                }
            }
            __syncthreads();                                                           // B

            // ---- rotate ------------------------------------------------------------------------------------------
            have_pend = cur_ok;
            p_tile = cur;
            p_T = c_T;
            p_pv = cin.pv;
            cur = nxt;
            cin = nin;
            nxt = nn;
        }
    }
}

template <int NW, bool L16>
int e4_launch_t(hipStream_t st, const EncBlk *dblk, int count, u64 *d_desc, u32 *d_tickets, u32 lmax)
{
    static int wgs_per_cu = 0, cus = 0;
    const u32 win_stride = ((u32)E4_GUARD + (u32)(((size_t)E4_TILE * lmax) >> 5) + 8u + 3u) & ~3u;     // dwords per buffer
    const size_t dyn = (size_t)win_stride * 2 * 4;
    if (!wgs_per_cu) {
        int dev = 0, occ = 0;
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        cus = prop.multiProcessorCount;
        const size_t dyn_max = (((size_t)E4_GUARD + (((size_t)E4_TILE * 16) >> 5) + 8 + 3) & ~(size_t)3) * 2 * 4;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)sfe4_kernel<NW, L16>, E4_THREADS, dyn_max));
        wgs_per_cu = occ < 1 ? 1 : (occ > 6 ? 6 : occ);
    }
    // the grid is a multiple of the number of concurrently served blocks, so a workgroup stays with one block
    int target = cus * wgs_per_cu;
    int nconc = count < target ? count : target;
    int per = target / nconc;
    if (per < 1) per = 1;
    hipLaunchKernelGGL((sfe4_kernel<NW, L16>), dim3((u32)(nconc * per)), dim3(E4_THREADS), dyn, st, dblk, count, nconc, d_desc, d_tickets,
                       win_stride);
    return SHAFA_SUCCESS;
}

}  // namespace

// launched from sfenc_launch (sf_encode.hip) for blocks whose codes are <= 16 bits when the launch holds enough blocks
// to keep every chain short; desc (one u64 per tile) and tickets (one u32 per block) are zeroed by the caller;
// tables are 256 x u64 {code, len}
int sfenc4_launch(hipStream_t st, const EncBlk *dblk, int count, u64 *d_desc, u32 *d_tickets, u32 lmax)
{
    if (lmax <= 8) return e4_launch_t<3, false>(st, dblk, count, d_desc, d_tickets, lmax);
    if (lmax <= 12) return e4_launch_t<4, false>(st, dblk, count, d_desc, d_tickets, lmax);
    if (lmax <= 15) return e4_launch_t<5, false>(st, dblk, count, d_desc, d_tickets, lmax);
    return e4_launch_t<5, true>(st, dblk, count, d_desc, d_tickets, lmax);
}
