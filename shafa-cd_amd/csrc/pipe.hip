// pipe.hip — layer 3 of the C-ABI: a bounded, in-order block pipeline.
//
// The reference starts one host thread per block and lets the reader run ahead without bound
// (multithread.c:126-194; c.c:383-411 keeps every block's input and output resident).  Here a fixed
// number of slots, each with its own HIP stream, pinned host buffers and device buffers, gives
//     fread(block b+2)  ‖  H2D(block b+1)  ‖  kernels(block b)  ‖  D2H(block b-1)  ‖  fwrite(block b-2)
// with at most n_slots blocks in flight.  The caller fills shafa_pipe_in(slot), submits, and retires
// slots in submission order, which is the reference's ordered-write chain (multithread.c:75-86).
//
// Transfers (round 4): the blocks' H2D copies of all slots of a device go through ONE stream, chained to the slot's kernel
// stream by an event; the payload D2H copies have a stream per slot.  With the H2D in its slot's own stream, two slots'
// blocks crossed the link at the same time at half its rate each, both slots' kernels and D2H then started late together and
// the host-to-device direction idled meanwhile (tools/dbg/pipe_trace.sh): 37-39 GiB/s encode.  In submission order every
// block has the whole link: 49 GiB/s.  The other direction is the opposite: one D2H stream for all slots stops at 38-39
// GiB/s (decode), two at 39.5, one per slot reaches 41-43 (tools/dbg/pipe_rate_ab.sh).
#include "common.hpp"
#include "internal.hpp"

#include <stdio.h>
#include <stdlib.h>

int shafa_set_hip_error(hipError_t e, const char *what);
int api_lazy_init();      // api.hip: shafa_hip_init(0) unless a device was selected already
int api_pipe_device(int slot, int n_slots);   // api.hip: device of slot `slot` of a pipe of n_slots (shafa_hip_init_devices), else the layer-1 device

#define PIPE_GROUP_MAX SHAFA_PIPE_GROUP_MAX

namespace {

struct Slot {
    int device;                // the slot's stream, buffers and kernels live here
    hipStream_t st;
    Batch *batch;
    u8 *h_in, *h_out;          // pinned
    size_t h_in_cap, h_out_cap;
    u8 *d_in, *d_out, *d_mid;  // device: input, result, SF-decoded bytes of the fused decode
    size_t d_in_cap, d_out_cap, d_mid_cap;
    // SHAFA_OP_FTC: tile histograms of the RLE bytes / of the input, the .shaf payload on the device and on the host
    u8 *d_th_rle, *d_th_in, *d_out2, *h_out2;
    size_t d_th_rle_cap, d_th_in_cap, d_out2_cap, h_out2_cap;
    int stage;                 // SHAFA_OP_FTC: 0 = idle, 1 = F submitted / retired (the slot stays reserved), 2 = C submitted
    int ftc_flags;
    size_t rle_n;              // SHAFA_OP_FTC: size of the RLE bytes (known after the first wait)
    u64 *d_small;              // [0..255] hist of the result / input, [256..511] hist of the input (-c f), [512] size, [513] mid size
    u64 *h_small;              // pinned mirror
    int op;
    bool busy, want_in_hist;
    size_t in_n, out_cap, n_symbols;
    size_t copied;             // bytes of the result fetched speculatively at submit (pipe_prefetch)
    int rc;                    // error found while submitting
    hipStream_t st_h2d, st_d2h;   // the device's H2D stream (shared by its slots, owned by the pipe); the slot's D2H stream
    hipEvent_t ev_in, ev_k, ev_out;   // block on the device / kernels done / payload on the host
    bool out_queued;           // a payload copy was queued on st_d2h (ev_out recorded)
    // a group of blocks in the slot (shafa_pipe_submit_group): block i's result at d_out / h_out + i * g_stride
    int g_n;                   // 0: the slot holds a single block
    bool g_wide;               // RLE decodes of the group with the full 85 x capacities (second attempt)
    size_t g_stride, g_mid_stride;
    u64 *g_in_off, *g_in_n, *g_out_off, *g_cap, *g_nsym, *g_mid_off;   // PIPE_GROUP_MAX entries each (allocated with the first group)
    shafa_pipe_block *g_blk;   // the group as it was submitted (for the second attempt)
    shafa_code_table *g_tab;
    u64 *d_gsmall, *h_gsmall;  // per block of a group: [i * 256] hist of the result / input, [(G + i) * 256] hist of the input, [2 G * 256 + i] size
};

}  // namespace

struct shafa_pipe {
    int n_slots;
    Slot *slots;
    size_t last_out[8];        // per op: result size of the block retired last (what the next submit fetches ahead)
    int n_xf;                  // H2D streams, one per device the slots live on
    struct { int device; hipStream_t h2d; } xf[64];
};

namespace {

// A result whose size only the device knows (RLE / SF encode, RLE decodes) is fetched SPECULATIVELY when the op is
// submitted: as many bytes as the pipe's previous block of that op produced, plus 3 % — consecutive blocks of a file differ
// little — by one asynchronous copy behind the op's kernels, so that it overlaps the next block's H2D on the other
// direction of the link; shafa_pipe_wait fetches what is missing (the first block of a pipe: everything).  (Fetching the
// whole payload from shafa_pipe_wait, after the host has seen the size, put every D2H behind a host round trip: 28 GiB/s
// where the link does two directions.  A copy KERNEL that reads the size on the device — exact, no host — was measured
// too: its 1-2 ms on a hardware queue hold back the barrier packets of the other slots' copies, 28-34 GiB/s.)
// the first nbytes of the slot's result, behind the kernels queued so far, on the device's D2H stream
int pipe_payload(Slot &s, size_t nbytes, bool second = false)
{
    if (!nbytes) return SHAFA_SUCCESS;
    HIP_TRY(hipEventRecord(s.ev_k, s.st));
    HIP_TRY(hipStreamWaitEvent(s.st_d2h, s.ev_k, 0));
    HIP_TRY(hipMemcpyAsync(second ? s.h_out2 : s.h_out, second ? s.d_out2 : s.d_out, nbytes, hipMemcpyDeviceToHost, s.st_d2h));
    HIP_TRY(hipEventRecord(s.ev_out, s.st_d2h));
    s.out_queued = true;
    return SHAFA_SUCCESS;
}

int pipe_prefetch(Slot &s, size_t pred, size_t cap, bool second = false)
{
    s.copied = 0;
    if (!pred) return SHAFA_SUCCESS;
    size_t nbytes = pred + pred / 32 + 4096;
    if (nbytes > cap) nbytes = cap;
    if (nbytes > (second ? s.h_out2_cap : s.h_out_cap)) nbytes = second ? s.h_out2_cap : s.h_out_cap;
    const int rc = pipe_payload(s, nbytes, second);
    if (rc) return rc;
    s.copied = nbytes;
    return SHAFA_SUCCESS;
}

int grow_pinned(u8 **p, size_t *cap, size_t need)
{
    if (need <= *cap) return SHAFA_SUCCESS;
    if (*p) { HIP_TRY(hipHostFree(*p)); *p = nullptr; *cap = 0; }
    const size_t want = (need + 4095) & ~(size_t)4095;
    HIP_TRY(hipHostMalloc((void **)p, want, hipHostMallocPortable));       // pinned for every device of the pipe
    *cap = want;
    return SHAFA_SUCCESS;
}

int grow_dev(u8 **p, size_t *cap, size_t need)
{
    if (need <= *cap) return SHAFA_SUCCESS;
    if (*p) { HIP_TRY(hipFree(*p)); *p = nullptr; *cap = 0; }
    const size_t want = (need + 64 + 255) & ~(size_t)255;
    HIP_TRY(hipMalloc((void **)p, want));
    *cap = want;
    return SHAFA_SUCCESS;
}

int slot_submit(Slot &s, const shafa_code_table *table, const size_t pred)
{
    int rc;
    DeviceGuard dg(s.device);                      // the caller's current device is restored on return
    if ((rc = batch_enter(s.batch, s.st))) return rc;
    const u64 off0[1] = {0}, in_n[1] = {s.in_n};
    if ((rc = grow_dev(&s.d_in, &s.d_in_cap, s.in_n))) return rc;
    if (s.in_n) {                                  // the block: the device's H2D stream, in submission order
        HIP_TRY(hipMemcpyAsync(s.d_in, s.h_in, s.in_n, hipMemcpyHostToDevice, s.st_h2d));
        HIP_TRY(hipEventRecord(s.ev_in, s.st_h2d));
        HIP_TRY(hipStreamWaitEvent(s.st, s.ev_in, 0));
    }
    u64 *d_size = s.d_small + 512;
    switch (s.op) {
    case SHAFA_OP_HIST:
        if ((rc = hist_launch(s.batch, s.st, 1, s.d_in, off0, in_n, s.d_small))) return rc;
        HIP_TRY(hipMemcpyAsync(s.h_small, s.d_small, 256 * sizeof(u64), hipMemcpyDeviceToHost, s.st));
        break;
    case SHAFA_OP_RLE_ENCODE: {
        const size_t cap = (2 * s.in_n + 3 + 15) & ~(size_t)15;                 // f.c:244 worst case
        if ((rc = grow_dev(&s.d_out, &s.d_out_cap, cap))) return rc;
        if ((rc = grow_pinned(&s.h_out, &s.h_out_cap, cap))) return rc;
        if (s.want_in_hist && (rc = hist_launch(s.batch, s.st, 1, s.d_in, off0, in_n, s.d_small + 256))) return rc;
        const u64 ocap[1] = {cap};
        if ((rc = rleenc_launch(s.batch, s.st, 1, s.d_in, off0, in_n, s.d_out, off0, ocap, d_size, s.d_small))) return rc;
        HIP_TRY(hipMemcpyAsync(s.h_small, s.d_small, 514 * sizeof(u64), hipMemcpyDeviceToHost, s.st));
        if ((rc = pipe_prefetch(s, pred, cap))) return rc;
        break;
    }
    case SHAFA_OP_FTC: {                                                         // stage one: Module F, the block stays
        const u64 th0[1] = {0};
        if (s.ftc_flags & SHAFA_PIPE_FTC_RLE) {
            const size_t cap = (2 * s.in_n + 3 + 15) & ~(size_t)15;             // f.c:244 worst case
            if ((rc = grow_dev(&s.d_out, &s.d_out_cap, cap))) return rc;
            if ((rc = grow_pinned(&s.h_out, &s.h_out_cap, cap))) return rc;
            if ((rc = grow_dev(&s.d_th_rle, &s.d_th_rle_cap, shafa_hip_tile_hist_bytes(cap)))) return rc;
            const u64 ocap[1] = {cap};
            if ((rc = rleenc_launch(s.batch, s.st, 1, s.d_in, off0, in_n, s.d_out, off0, ocap, d_size, s.d_small, s.d_th_rle, th0))) return rc;
        }
        if (s.ftc_flags & (SHAFA_PIPE_FTC_PLAIN | SHAFA_PIPE_INPUT_HIST)) {
            if ((rc = grow_dev(&s.d_th_in, &s.d_th_in_cap, shafa_hip_tile_hist_bytes(s.in_n)))) return rc;
            u64 *d_f = (s.ftc_flags & SHAFA_PIPE_FTC_RLE) ? s.d_small + 256 : s.d_small;   // as HIST / RLE_ENCODE put them
            if ((rc = hist_launch_dev(s.batch, s.st, 1, s.d_in, off0, in_n, nullptr, d_f, s.d_th_in, th0))) return rc;
        }
        HIP_TRY(hipMemcpyAsync(s.h_small, s.d_small, 514 * sizeof(u64), hipMemcpyDeviceToHost, s.st));
        if ((s.ftc_flags & SHAFA_PIPE_FTC_RLE) && (rc = pipe_prefetch(s, pred, s.d_out_cap))) return rc;
        break;
    }
    case SHAFA_OP_SF_ENCODE: {
        if (!table) return SHAFA_OUTSIDE_MODULE;
        if ((rc = grow_dev(&s.d_out, &s.d_out_cap, s.out_cap + 16))) return rc;
        if ((rc = grow_pinned(&s.h_out, &s.h_out_cap, s.out_cap))) return rc;
        const u64 ocap[1] = {s.out_cap};
        if ((rc = sfenc_launch(s.batch, s.st, 1, s.d_in, off0, in_n, table, s.d_out, off0, ocap, d_size))) return rc;
        HIP_TRY(hipMemcpyAsync(s.h_small + 512, d_size, sizeof(u64), hipMemcpyDeviceToHost, s.st));
        if ((rc = pipe_prefetch(s, pred, s.out_cap))) return rc;
        break;
    }
    case SHAFA_OP_SF_DECODE: {
        if (!table) return SHAFA_OUTSIDE_MODULE;
        if ((rc = grow_dev(&s.d_out, &s.d_out_cap, s.n_symbols))) return rc;
        if ((rc = grow_pinned(&s.h_out, &s.h_out_cap, s.n_symbols))) return rc;
        const u64 ns[1] = {s.n_symbols};
        if ((rc = sfdec_launch(s.batch, s.st, 1, s.d_in, off0, in_n, table, ns, s.d_out, off0))) return rc;
        if ((rc = pipe_payload(s, s.n_symbols))) return rc;
        break;
    }
    case SHAFA_OP_RLE_DECODE:
    case SHAFA_OP_SF_RLE_DECODE: {
        const u8 *rle_in = s.d_in;
        u64 rle_n[1] = {s.in_n};
        if (s.op == SHAFA_OP_SF_RLE_DECODE) {                                   // d.c:565-586, fused on the device
            if (!table) return SHAFA_OUTSIDE_MODULE;
            if ((rc = grow_dev(&s.d_mid, &s.d_mid_cap, s.n_symbols))) return rc;
            const u64 ns[1] = {s.n_symbols};
            if ((rc = sfdec_launch(s.batch, s.st, 1, s.d_in, off0, in_n, table, ns, s.d_mid, off0))) return rc;
            rle_in = s.d_mid;
            rle_n[0] = s.n_symbols;
        }
        const size_t cap = SHAFA_RLE_DECODE_MAX;                                 // d.c:129-169
        if ((rc = grow_dev(&s.d_out, &s.d_out_cap, cap))) return rc;
        if ((rc = grow_pinned(&s.h_out, &s.h_out_cap, cap))) return rc;
        const u64 ocap[1] = {cap};
        if ((rc = rledec_launch(s.batch, s.st, 1, rle_in, off0, rle_n, s.d_out, off0, ocap, d_size))) return rc;
        HIP_TRY(hipMemcpyAsync(s.h_small + 512, d_size, sizeof(u64), hipMemcpyDeviceToHost, s.st));
        if ((rc = pipe_prefetch(s, pred, cap))) return rc;
        break;
    }
    default:
        return SHAFA_OUTSIDE_MODULE;
    }
    return SHAFA_SUCCESS;
}

// ---- groups: several consecutive blocks of a file in one slot, one launch per kernel for all of them --------------------
// (one launch per block costs the submitting thread ~120 us whatever the block's size: at the reference's default 64 KiB
// blocks that is all a file's time.  Results lie at a uniform stride so that they come back as one 2-D copy.)
constexpr size_t G_SIZE_AT = (size_t)2 * PIPE_GROUP_MAX * 256;      // u64 index of block 0's size in d_gsmall

int group_alloc(Slot &s)
{
    if (s.g_in_off) return SHAFA_SUCCESS;
    u64 *a = (u64 *)calloc((size_t)6 * PIPE_GROUP_MAX, sizeof(u64));
    shafa_code_table *t = (shafa_code_table *)calloc(PIPE_GROUP_MAX, sizeof(shafa_code_table));
    s.g_blk = (shafa_pipe_block *)calloc(PIPE_GROUP_MAX, sizeof(shafa_pipe_block));
    if (!a || !t || !s.g_blk) { free(a); free(t); free(s.g_blk); s.g_blk = nullptr; return SHAFA_LACK_OF_MEMORY; }
    const size_t nsmall = G_SIZE_AT + 2 * PIPE_GROUP_MAX;
    HIP_TRY(hipMalloc((void **)&s.d_gsmall, nsmall * sizeof(u64)));
    HIP_TRY(hipHostMalloc((void **)&s.h_gsmall, nsmall * sizeof(u64), hipHostMallocPortable));
    s.g_in_off = a; s.g_in_n = a + PIPE_GROUP_MAX; s.g_out_off = a + 2 * PIPE_GROUP_MAX; s.g_cap = a + 3 * PIPE_GROUP_MAX;
    s.g_nsym = a + 4 * PIPE_GROUP_MAX; s.g_mid_off = a + 5 * PIPE_GROUP_MAX;
    s.g_tab = t;
    return SHAFA_SUCCESS;
}

// results at a uniform stride: rows of `width` bytes, one per block
int group_fetch(Slot &s, size_t width, hipStream_t st)
{
    if (!width || !s.g_n) return SHAFA_SUCCESS;
    if (width > s.g_stride) width = s.g_stride;
    if (s.g_n == 1) HIP_TRY(hipMemcpyAsync(s.h_out, s.d_out, width, hipMemcpyDeviceToHost, st));
    else HIP_TRY(hipMemcpy2DAsync(s.h_out, s.g_stride, s.d_out, s.g_stride, width, (size_t)s.g_n, hipMemcpyDeviceToHost, st));
    return SHAFA_SUCCESS;
}

int slot_submit_group(Slot &s, int n, const shafa_pipe_block *blk)
{
    int rc;
    DeviceGuard dg(s.device);
    if ((rc = group_alloc(s))) return rc;
    if ((rc = batch_enter(s.batch, s.st))) return rc;
    if (blk != s.g_blk) memcpy(s.g_blk, blk, (size_t)n * sizeof(*blk));
    size_t total_in = 0, max_cap = 0, max_sym = 0;
    for (int i = 0; i < n; ++i) {
        if (blk[i].in_off & 15) return SHAFA_OUTSIDE_MODULE;
        size_t in_end = 0;
        if (__builtin_add_overflow(blk[i].in_off, blk[i].in_n, &in_end) || in_end > s.h_in_cap) return SHAFA_OUTSIDE_MODULE;
        s.g_in_off[i] = blk[i].in_off;
        s.g_in_n[i] = blk[i].in_n;
        if (in_end > total_in) total_in = in_end;
        // n_symbols comes from a file's text.  A code has at least one bit, so a stream of in_n bytes holds at most 8 in_n
        // symbols: rows are sized for that at most (a block that announces more ends as "stream too short", the kernels
        // write only symbols they counted), and no product below can wrap
        const size_t sym_most = blk[i].in_n <= (SIZE_MAX >> 4) ? 8 * blk[i].in_n + 1 : SIZE_MAX >> 1;      // + 1: still "too short"
        const size_t nsym = blk[i].n_symbols < sym_most ? blk[i].n_symbols : sym_most;
        s.g_nsym[i] = nsym;
        size_t cap = 0;
        switch (s.op) {
        case SHAFA_OP_RLE_ENCODE:                                                                      // f.c:244 worst case
            if (blk[i].in_n > (SIZE_MAX >> 2)) return SHAFA_LACK_OF_MEMORY;
            cap = 2 * blk[i].in_n + 3;
            break;
        case SHAFA_OP_SF_ENCODE: cap = blk[i].out_cap; break;
        case SHAFA_OP_SF_DECODE: cap = nsym; break;
        case SHAFA_OP_RLE_DECODE:                                                                      // a triple of 3 bytes expands to 255 at most
        case SHAFA_OP_SF_RLE_DECODE: {
            const size_t rle_n = s.op == SHAFA_OP_RLE_DECODE ? blk[i].in_n : nsym;
            cap = rle_n <= (size_t)SHAFA_RLE_DECODE_MAX / 85 ? 85 * rle_n + 256 : (size_t)SHAFA_RLE_DECODE_MAX;
            if (cap > (size_t)SHAFA_RLE_DECODE_MAX) cap = SHAFA_RLE_DECODE_MAX;                     // d.c:129-169
            // ... but real blocks expand by a few per cent: the group is sized for 8 x + 4 KiB first (s.g_wide false) and
            // decoded again with the full capacities when a block does not fit (shafa_pipe_wait_group)
            if (!s.g_wide && cap > 8 * rle_n + 4096) cap = 8 * rle_n + 4096;
            break;
        }
        default: break;
        }
        s.g_cap[i] = cap;
        if (cap > max_cap) max_cap = cap;
        if (nsym > max_sym) max_sym = nsym;
        if ((s.op == SHAFA_OP_SF_ENCODE || s.op == SHAFA_OP_SF_DECODE || s.op == SHAFA_OP_SF_RLE_DECODE)) {
            if (!blk[i].table) return SHAFA_OUTSIDE_MODULE;
            if (blk[i].table != &s.g_tab[i]) s.g_tab[i] = *blk[i].table;
            s.g_blk[i].table = &s.g_tab[i];                                   // the caller's table need not outlive the call
        }
    }
    if (total_in > s.h_in_cap) return SHAFA_OUTSIDE_MODULE;
    if (max_cap > (SIZE_MAX >> 8) || max_sym > (SIZE_MAX >> 8)) return SHAFA_LACK_OF_MEMORY;
    s.g_stride = (max_cap + 16 + 15) & ~(size_t)15;
    s.g_mid_stride = (max_sym + 16 + 15) & ~(size_t)15;
    size_t total_out = 0, total_mid = 0;
    if (__builtin_mul_overflow((size_t)n, s.g_stride, &total_out) || __builtin_mul_overflow((size_t)n, s.g_mid_stride, &total_mid))
        return SHAFA_LACK_OF_MEMORY;
    for (int i = 0; i < n; ++i) { s.g_out_off[i] = (u64)i * s.g_stride; s.g_mid_off[i] = (u64)i * s.g_mid_stride; }
    if ((rc = grow_dev(&s.d_in, &s.d_in_cap, total_in))) return rc;
    if (total_in && !s.g_wide) {                   // (the second attempt finds the input where the first left it)
        HIP_TRY(hipMemcpyAsync(s.d_in, s.h_in, total_in, hipMemcpyHostToDevice, s.st_h2d));
        HIP_TRY(hipEventRecord(s.ev_in, s.st_h2d));
        HIP_TRY(hipStreamWaitEvent(s.st, s.ev_in, 0));
    }
    u64 *d_size = s.d_gsmall + G_SIZE_AT;
    if (s.op != SHAFA_OP_HIST) {
        if ((rc = grow_dev(&s.d_out, &s.d_out_cap, total_out))) return rc;
        if ((rc = grow_pinned(&s.h_out, &s.h_out_cap, total_out))) return rc;
    }
    switch (s.op) {
    case SHAFA_OP_HIST:
        if ((rc = hist_launch(s.batch, s.st, n, s.d_in, s.g_in_off, s.g_in_n, s.d_gsmall))) return rc;
        HIP_TRY(hipMemcpyAsync(s.h_gsmall, s.d_gsmall, (size_t)n * 256 * sizeof(u64), hipMemcpyDeviceToHost, s.st));
        break;
    case SHAFA_OP_RLE_ENCODE:
        if (s.want_in_hist) {
            if ((rc = hist_launch(s.batch, s.st, n, s.d_in, s.g_in_off, s.g_in_n, s.d_gsmall + (size_t)PIPE_GROUP_MAX * 256))) return rc;
            HIP_TRY(hipMemcpyAsync(s.h_gsmall + (size_t)PIPE_GROUP_MAX * 256, s.d_gsmall + (size_t)PIPE_GROUP_MAX * 256,
                                   (size_t)n * 256 * sizeof(u64), hipMemcpyDeviceToHost, s.st));
        }
        if ((rc = rleenc_launch(s.batch, s.st, n, s.d_in, s.g_in_off, s.g_in_n, s.d_out, s.g_out_off, s.g_cap, d_size, s.d_gsmall))) return rc;
        HIP_TRY(hipMemcpyAsync(s.h_gsmall, s.d_gsmall, (size_t)n * 256 * sizeof(u64), hipMemcpyDeviceToHost, s.st));
        HIP_TRY(hipMemcpyAsync(s.h_gsmall + G_SIZE_AT, d_size, (size_t)n * sizeof(u64), hipMemcpyDeviceToHost, s.st));
        break;
    case SHAFA_OP_SF_ENCODE:
        if ((rc = sfenc_launch(s.batch, s.st, n, s.d_in, s.g_in_off, s.g_in_n, s.g_tab, s.d_out, s.g_out_off, s.g_cap, d_size))) return rc;
        HIP_TRY(hipMemcpyAsync(s.h_gsmall + G_SIZE_AT, d_size, (size_t)n * sizeof(u64), hipMemcpyDeviceToHost, s.st));
        break;
    case SHAFA_OP_SF_DECODE:
        if ((rc = sfdec_launch(s.batch, s.st, n, s.d_in, s.g_in_off, s.g_in_n, s.g_tab, s.g_nsym, s.d_out, s.g_out_off))) return rc;
        HIP_TRY(hipEventRecord(s.ev_k, s.st));                                   // sizes known: the rows leave behind the kernels
        HIP_TRY(hipStreamWaitEvent(s.st_d2h, s.ev_k, 0));
        if ((rc = group_fetch(s, max_sym, s.st_d2h))) return rc;
        HIP_TRY(hipEventRecord(s.ev_out, s.st_d2h));
        s.out_queued = true;
        break;
    case SHAFA_OP_RLE_DECODE:
    case SHAFA_OP_SF_RLE_DECODE: {
        const u8 *rle_in = s.d_in;
        const u64 *rle_off = s.g_in_off, *rle_n = s.g_in_n;
        if (s.op == SHAFA_OP_SF_RLE_DECODE) {                                   // d.c:565-586, fused on the device
            if ((rc = grow_dev(&s.d_mid, &s.d_mid_cap, total_mid))) return rc;
            if ((rc = sfdec_launch(s.batch, s.st, n, s.d_in, s.g_in_off, s.g_in_n, s.g_tab, s.g_nsym, s.d_mid, s.g_mid_off))) return rc;
            rle_in = s.d_mid;
            rle_off = s.g_mid_off;
            rle_n = s.g_nsym;
        }
        if ((rc = rledec_launch(s.batch, s.st, n, rle_in, rle_off, rle_n, s.d_out, s.g_out_off, s.g_cap, d_size))) return rc;
        HIP_TRY(hipMemcpyAsync(s.h_gsmall + G_SIZE_AT, d_size, (size_t)n * sizeof(u64), hipMemcpyDeviceToHost, s.st));
        break;
    }
    default:
        return SHAFA_OUTSIDE_MODULE;
    }
    return SHAFA_SUCCESS;
}

}  // namespace

extern "C" {

int shafa_pipe_create(int n_slots, shafa_pipe **out)
{
    if (!out || n_slots <= 0 || n_slots > 64) return SHAFA_OUTSIDE_MODULE;
    int rc = api_lazy_init();
    if (rc) return rc;
    shafa_pipe *p = (shafa_pipe *)calloc(1, sizeof(shafa_pipe));
    if (!p) return SHAFA_LACK_OF_MEMORY;
    p->slots = (Slot *)calloc((size_t)n_slots, sizeof(Slot));
    if (!p->slots) { free(p); return SHAFA_LACK_OF_MEMORY; }
    p->n_slots = n_slots;
    for (int i = 0; i < n_slots; ++i) {
        Slot &s = p->slots[i];
        shafa_hipd_batch *bh = nullptr;
        s.device = api_pipe_device(i, n_slots);
        DeviceGuard dg(s.device);
        hipError_t e = hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking);
        int x = 0;
        while (x < p->n_xf && p->xf[x].device != s.device) ++x;
        if (e == hipSuccess && x == p->n_xf) {          // the device's first slot: its H2D stream
            p->xf[x].device = s.device;
            e = hipStreamCreateWithFlags(&p->xf[x].h2d, hipStreamNonBlocking);
            if (e == hipSuccess) p->n_xf = x + 1;
        }
        if (e == hipSuccess) {
            s.st_h2d = p->xf[x].h2d;
            e = hipStreamCreateWithFlags(&s.st_d2h, hipStreamNonBlocking);
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_k, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc((void **)&s.d_small, 514 * sizeof(u64));
        if (e == hipSuccess) e = hipHostMalloc((void **)&s.h_small, 514 * sizeof(u64), hipHostMallocPortable);
        if (e != hipSuccess) { shafa_pipe_destroy(p); return shafa_set_hip_error(e, "shafa_pipe_create"); }
        if ((rc = shafa_hipd_batch_create(PIPE_GROUP_MAX, (size_t)1 << 27, &bh))) { shafa_pipe_destroy(p); return rc; }
        s.batch = (Batch *)bh;
        s.batch->par_dma = true;                        // see batch_params_commit
    }
    *out = p;
    return SHAFA_SUCCESS;
}

void shafa_pipe_destroy(shafa_pipe *p)
{
    if (!p) return;
    for (int i = 0; i < p->n_slots; ++i) {
        Slot &s = p->slots[i];
        DeviceGuard dg(s.device);
        if (s.st_h2d) (void)hipStreamSynchronize(s.st_h2d);
        if (s.st) (void)hipStreamSynchronize(s.st);
        if (s.st_d2h) (void)hipStreamSynchronize(s.st_d2h);
        if (s.ev_in) (void)hipEventDestroy(s.ev_in);
        if (s.ev_k) (void)hipEventDestroy(s.ev_k);
        if (s.ev_out) (void)hipEventDestroy(s.ev_out);
        if (s.batch) shafa_hipd_batch_destroy((shafa_hipd_batch *)s.batch);
        if (s.h_in) hipHostFree(s.h_in);
        if (s.h_out) hipHostFree(s.h_out);
        if (s.d_in) hipFree(s.d_in);
        if (s.d_out) hipFree(s.d_out);
        if (s.d_mid) hipFree(s.d_mid);
        if (s.d_th_rle) hipFree(s.d_th_rle);
        if (s.d_th_in) hipFree(s.d_th_in);
        if (s.d_out2) hipFree(s.d_out2);
        if (s.h_out2) hipHostFree(s.h_out2);
        if (s.d_small) hipFree(s.d_small);
        if (s.h_small) hipHostFree(s.h_small);
        if (s.d_gsmall) hipFree(s.d_gsmall);
        if (s.h_gsmall) hipHostFree(s.h_gsmall);
        free(s.g_in_off);
        free(s.g_tab);
        free(s.g_blk);
        if (s.st) hipStreamDestroy(s.st);
        if (s.st_d2h) hipStreamDestroy(s.st_d2h);
    }
    for (int x = 0; x < p->n_xf; ++x) {
        DeviceGuard dg(p->xf[x].device);
        if (p->xf[x].h2d) hipStreamDestroy(p->xf[x].h2d);
    }
    free(p->slots);
    free(p);
}

int shafa_pipe_slots(const shafa_pipe *p) { return p ? p->n_slots : 0; }
int shafa_pipe_slot_device(const shafa_pipe *p, int slot) { return (p && slot >= 0 && slot < p->n_slots) ? p->slots[slot].device : -1; }

uint8_t *shafa_pipe_in(shafa_pipe *p, int slot, size_t bytes)
{
    if (!p || slot < 0 || slot >= p->n_slots || p->slots[slot].busy || p->slots[slot].stage) return nullptr;
    Slot &s = p->slots[slot];
    if (grow_pinned(&s.h_in, &s.h_in_cap, bytes ? bytes : 1)) return nullptr;
    return s.h_in;
}

int shafa_pipe_submit(shafa_pipe *p, int slot, int op, size_t in_n, const shafa_code_table *table,
                      size_t n_symbols, size_t out_cap, int flags)
{
    if (!p || slot < 0 || slot >= p->n_slots) return SHAFA_OUTSIDE_MODULE;
    Slot &s = p->slots[slot];
    if (s.busy || s.stage || in_n > s.h_in_cap) return SHAFA_OUTSIDE_MODULE;
    if (op == SHAFA_OP_FTC && !(flags & (SHAFA_PIPE_FTC_RLE | SHAFA_PIPE_FTC_PLAIN))) return SHAFA_OUTSIDE_MODULE;
    s.op = op;
    s.in_n = in_n;
    s.n_symbols = n_symbols;
    s.out_cap = out_cap;
    s.want_in_hist = (flags & SHAFA_PIPE_INPUT_HIST) != 0;
    s.ftc_flags = flags;
    s.stage = op == SHAFA_OP_FTC ? 1 : 0;
    s.busy = true;
    s.copied = 0;
    s.out_queued = false;
    s.g_n = 0;
    s.rc = slot_submit(s, table, (op >= 0 && op < 8) ? p->last_out[op] : 0);            // errors are reported by shafa_pipe_wait, in block order
    return SHAFA_SUCCESS;
}

int shafa_pipe_wait(shafa_pipe *p, int slot, shafa_pipe_result *res)
{
    if (!p || slot < 0 || slot >= p->n_slots || !res) return SHAFA_OUTSIDE_MODULE;
    Slot &s = p->slots[slot];
    if (!s.busy) return SHAFA_OUTSIDE_MODULE;
    s.busy = false;
    memset(res, 0, sizeof(*res));
    DeviceGuard dg(s.device);
    int rc = shafa_hipd_finish((shafa_hipd_batch *)s.batch, s.st, 1, nullptr);   // synchronises the slot's stream
    if (s.out_queued) {                              // and the payload copy behind it on the device's D2H stream
        const hipError_t e = hipEventSynchronize(s.ev_out);
        if (e != hipSuccess && !s.rc && !rc) rc = shafa_set_hip_error(e, "shafa_pipe_wait");
    }
    if (s.rc || rc) { s.stage = 0; return s.rc ? s.rc : rc; }
    res->out = s.h_out;
    if (s.op == SHAFA_OP_FTC && s.stage == 2) {          // Module C's payload; the slot is idle again
        s.stage = 0;
        res->out = s.h_out2;
        const size_t sz = (size_t)s.h_small[513];
        if (sz > s.out_cap || sz > s.h_out2_cap) return SHAFA_LACK_OF_MEMORY;
        if (sz > s.copied) {
            HIP_TRY(hipMemcpyAsync(s.h_out2 + s.copied, s.d_out2 + s.copied, sz - s.copied, hipMemcpyDeviceToHost, s.st));
            HIP_TRY(hipStreamSynchronize(s.st));
        }
        p->last_out[SHAFA_OP_SF_ENCODE] = sz;
        res->out_n = sz;
        return SHAFA_SUCCESS;
    }
    switch (s.op) {
    case SHAFA_OP_FTC:                                   // Module F's results; the slot stays reserved for shafa_pipe_ftc_encode
        memcpy(res->freq, s.h_small, 256 * sizeof(u64));
        if ((s.ftc_flags & SHAFA_PIPE_FTC_RLE) && (s.ftc_flags & (SHAFA_PIPE_FTC_PLAIN | SHAFA_PIPE_INPUT_HIST)))
            memcpy(res->freq_in, s.h_small + 256, 256 * sizeof(u64));       // (only a stage one that counted the input has it)
        s.rle_n = (s.ftc_flags & SHAFA_PIPE_FTC_RLE) ? (size_t)s.h_small[512] : 0;
        if (!(s.ftc_flags & SHAFA_PIPE_FTC_RLE)) { res->out_n = 0; return SHAFA_SUCCESS; }
        break;
    case SHAFA_OP_HIST:
        memcpy(res->freq, s.h_small, 256 * sizeof(u64));
        return SHAFA_SUCCESS;
    case SHAFA_OP_SF_DECODE:
        res->out_n = s.n_symbols;
        return SHAFA_SUCCESS;
    case SHAFA_OP_RLE_ENCODE:
        memcpy(res->freq, s.h_small, 256 * sizeof(u64));
        if (s.want_in_hist) memcpy(res->freq_in, s.h_small + 256, 256 * sizeof(u64));
        break;
    case SHAFA_OP_SF_ENCODE:
        if (s.h_small[512] > s.out_cap) return SHAFA_LACK_OF_MEMORY;
        break;
    case SHAFA_OP_SF_RLE_DECODE:
        res->mid_n = s.n_symbols;
        break;
    default:
        break;
    }
    // the size is known now: fetch what the speculative copy at submit did not bring (usually nothing)
    const size_t sz = (size_t)s.h_small[512];
    const int tail_rc = [&]() -> int {
        if (sz > s.h_out_cap) return SHAFA_LACK_OF_MEMORY;
        if (sz > s.copied) {
            HIP_TRY(hipMemcpyAsync(s.h_out + s.copied, s.d_out + s.copied, sz - s.copied, hipMemcpyDeviceToHost, s.st));
            HIP_TRY(hipStreamSynchronize(s.st));
        }
        return SHAFA_SUCCESS;
    }();
    if (tail_rc) { s.stage = 0; return tail_rc; }          // (a failed FTC stage one does not keep the slot reserved)
    if (s.op >= 0 && s.op < 8) p->last_out[s.op] = sz;
    res->out_n = sz;
    return SHAFA_SUCCESS;
}

int shafa_pipe_ftc_encode(shafa_pipe *p, int slot, int use_rle, const shafa_code_table *table, size_t out_cap)
{
    if (!p || slot < 0 || slot >= p->n_slots || !table) return SHAFA_OUTSIDE_MODULE;
    Slot &s = p->slots[slot];
    if (s.busy || s.stage != 1 || s.op != SHAFA_OP_FTC) return SHAFA_OUTSIDE_MODULE;
    if (use_rle ? !(s.ftc_flags & SHAFA_PIPE_FTC_RLE) : !(s.ftc_flags & (SHAFA_PIPE_FTC_PLAIN | SHAFA_PIPE_INPUT_HIST)))
        return SHAFA_OUTSIDE_MODULE;                       // stage one did not leave that form's tile histograms
    s.stage = 2;
    s.busy = true;
    s.copied = 0;
    s.out_queued = false;
    s.out_cap = out_cap;
    s.rc = [&]() -> int {                                  // errors are reported by shafa_pipe_wait, in block order
        int rc;
        DeviceGuard dg(s.device);
        if ((rc = batch_enter(s.batch, s.st))) return rc;
        if ((rc = grow_dev(&s.d_out2, &s.d_out2_cap, out_cap + 16))) return rc;
        if ((rc = grow_pinned(&s.h_out2, &s.h_out2_cap, out_cap ? out_cap : 16))) return rc;
        const u64 off0[1] = {0}, n[1] = {use_rle ? (u64)s.rle_n : (u64)s.in_n}, ocap[1] = {out_cap};
        if ((rc = sfenc_launch(s.batch, s.st, 1, use_rle ? s.d_out : s.d_in, off0, n, table, s.d_out2, off0, ocap, s.d_small + 513,
                               use_rle ? s.d_th_rle : s.d_th_in, off0))) return rc;
        HIP_TRY(hipMemcpyAsync(s.h_small + 513, s.d_small + 513, sizeof(u64), hipMemcpyDeviceToHost, s.st));
        return pipe_prefetch(s, p->last_out[SHAFA_OP_SF_ENCODE], out_cap, true);
    }();
    return SHAFA_SUCCESS;
}

int shafa_pipe_submit_group(shafa_pipe *p, int slot, int op, int nblocks, const shafa_pipe_block *blocks, int flags)
{
    if (!p || slot < 0 || slot >= p->n_slots || !blocks || nblocks <= 0 || nblocks > PIPE_GROUP_MAX) return SHAFA_OUTSIDE_MODULE;
    Slot &s = p->slots[slot];
    if (s.busy || s.stage || op == SHAFA_OP_FTC) return SHAFA_OUTSIDE_MODULE;
    s.op = op;
    s.want_in_hist = (flags & SHAFA_PIPE_INPUT_HIST) != 0;
    s.busy = true;
    s.copied = 0;
    s.out_queued = false;
    s.g_n = nblocks;
    s.g_wide = false;
    s.rc = slot_submit_group(s, nblocks, blocks);        // errors are reported by shafa_pipe_wait_group
    return SHAFA_SUCCESS;
}

int shafa_pipe_wait_group(shafa_pipe *p, int slot, int nblocks, shafa_pipe_result *res, int *block_rc)
{
    if (!p || slot < 0 || slot >= p->n_slots || !res || !block_rc) return SHAFA_OUTSIDE_MODULE;
    Slot &s = p->slots[slot];
    if (!s.busy || s.g_n != nblocks) return SHAFA_OUTSIDE_MODULE;
    s.busy = false;
    memset(res, 0, (size_t)nblocks * sizeof(*res));
    for (int i = 0; i < nblocks; ++i) block_rc[i] = SHAFA_SUCCESS;
    DeviceGuard dg(s.device);
    bool call_failed = false;
    int rc = batch_finish(s.batch, s.st, nblocks, block_rc, &call_failed);   // synchronises the slot's stream
    if (s.out_queued) {
        const hipError_t e = hipEventSynchronize(s.ev_out);
        if (e != hipSuccess && !s.rc) s.rc = shafa_set_hip_error(e, "shafa_pipe_wait_group");
    }
    if (!s.rc && !s.g_wide && (s.op == SHAFA_OP_RLE_DECODE || s.op == SHAFA_OP_SF_RLE_DECODE)) {
        bool again = false;                              // a block that expands more than eightfold: the whole group once more,
        for (int i = 0; i < nblocks; ++i) again = again || block_rc[i] == SHAFA_LACK_OF_MEMORY;   //   sized for the worst case
        if (again) {
            s.g_wide = true;
            s.rc = slot_submit_group(s, nblocks, s.g_blk);
            for (int i = 0; i < nblocks; ++i) block_rc[i] = SHAFA_SUCCESS;
            rc = batch_finish(s.batch, s.st, nblocks, block_rc, &call_failed);
        }
    }
    if (s.rc) return s.rc;                               // the submission itself failed: no block has a result
    if (call_failed) return rc;                          // so did the wait (stream or copy error): sizes and rows are stale
    const u64 *h_size = s.h_gsmall + G_SIZE_AT;
    size_t width = 0;
    for (int i = 0; i < nblocks; ++i) {
        shafa_pipe_result &r = res[i];
        r.out = s.h_out + (size_t)i * s.g_stride;
        switch (s.op) {
        case SHAFA_OP_HIST:
            memcpy(r.freq, s.h_gsmall + (size_t)i * 256, 256 * sizeof(u64));
            break;
        case SHAFA_OP_SF_DECODE:
            r.out_n = (size_t)s.g_nsym[i];
            break;
        case SHAFA_OP_RLE_ENCODE:
            memcpy(r.freq, s.h_gsmall + (size_t)i * 256, 256 * sizeof(u64));
            if (s.want_in_hist) memcpy(r.freq_in, s.h_gsmall + ((size_t)PIPE_GROUP_MAX + i) * 256, 256 * sizeof(u64));
            r.out_n = (size_t)h_size[i];
            break;
        case SHAFA_OP_SF_ENCODE:
            r.out_n = (size_t)h_size[i];
            if (!block_rc[i] && r.out_n > s.g_cap[i]) block_rc[i] = SHAFA_LACK_OF_MEMORY;
            break;
        case SHAFA_OP_SF_RLE_DECODE:
            r.mid_n = (size_t)s.g_nsym[i];
            r.out_n = (size_t)h_size[i];
            break;
        default:
            r.out_n = (size_t)h_size[i];
            break;
        }
        if (block_rc[i]) r.out_n = 0;
        if (s.op != SHAFA_OP_HIST && s.op != SHAFA_OP_SF_DECODE && r.out_n > width) width = r.out_n;
    }
    if (width) {                                         // the sizes are known now: the rows, as wide as the largest
        int frc = group_fetch(s, width, s.st);
        if (frc) return frc;
        HIP_TRY(hipStreamSynchronize(s.st));
    }
    return SHAFA_SUCCESS;
}

}  // extern "C"
