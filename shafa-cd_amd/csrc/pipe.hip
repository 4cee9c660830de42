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

namespace {

struct Slot {
    int device;                // the slot's stream, buffers and kernels live here
    hipStream_t st;
    Batch *batch;
    u8 *h_in, *h_out;          // pinned
    size_t h_in_cap, h_out_cap;
    u8 *d_in, *d_out, *d_mid;  // device: input, result, SF-decoded bytes of the fused decode
    size_t d_in_cap, d_out_cap, d_mid_cap;
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
int pipe_payload(Slot &s, size_t nbytes)
{
    if (!nbytes) return SHAFA_SUCCESS;
    HIP_TRY(hipEventRecord(s.ev_k, s.st));
    HIP_TRY(hipStreamWaitEvent(s.st_d2h, s.ev_k, 0));
    HIP_TRY(hipMemcpyAsync(s.h_out, s.d_out, nbytes, hipMemcpyDeviceToHost, s.st_d2h));
    HIP_TRY(hipEventRecord(s.ev_out, s.st_d2h));
    s.out_queued = true;
    return SHAFA_SUCCESS;
}

int pipe_prefetch(Slot &s, size_t pred, size_t cap)
{
    s.copied = 0;
    if (!pred) return SHAFA_SUCCESS;
    size_t nbytes = pred + pred / 32 + 4096;
    if (nbytes > cap) nbytes = cap;
    if (nbytes > s.h_out_cap) nbytes = s.h_out_cap;
    const int rc = pipe_payload(s, nbytes);
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
        if ((rc = shafa_hipd_batch_create(1, (size_t)1 << 27, &bh))) { shafa_pipe_destroy(p); return rc; }
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
        if (s.d_small) hipFree(s.d_small);
        if (s.h_small) hipHostFree(s.h_small);
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
    if (!p || slot < 0 || slot >= p->n_slots || p->slots[slot].busy) return nullptr;
    Slot &s = p->slots[slot];
    if (grow_pinned(&s.h_in, &s.h_in_cap, bytes ? bytes : 1)) return nullptr;
    return s.h_in;
}

int shafa_pipe_submit(shafa_pipe *p, int slot, int op, size_t in_n, const shafa_code_table *table,
                      size_t n_symbols, size_t out_cap, int flags)
{
    if (!p || slot < 0 || slot >= p->n_slots) return SHAFA_OUTSIDE_MODULE;
    Slot &s = p->slots[slot];
    if (s.busy || in_n > s.h_in_cap) return SHAFA_OUTSIDE_MODULE;
    s.op = op;
    s.in_n = in_n;
    s.n_symbols = n_symbols;
    s.out_cap = out_cap;
    s.want_in_hist = (flags & SHAFA_PIPE_INPUT_HIST) != 0;
    s.busy = true;
    s.copied = 0;
    s.out_queued = false;
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
    if (s.rc) return s.rc;
    if (rc) return rc;
    res->out = s.h_out;
    switch (s.op) {
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
    if (sz > s.h_out_cap) return SHAFA_LACK_OF_MEMORY;
    if (sz > s.copied) {
        HIP_TRY(hipMemcpyAsync(s.h_out + s.copied, s.d_out + s.copied, sz - s.copied, hipMemcpyDeviceToHost, s.st));
        HIP_TRY(hipStreamSynchronize(s.st));
    }
    if (s.op >= 0 && s.op < 8) p->last_out[s.op] = sz;
    res->out_n = sz;
    return SHAFA_SUCCESS;
}

}  // extern "C"
