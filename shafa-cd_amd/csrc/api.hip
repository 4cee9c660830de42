// api.hip — the C-ABI of libshafa_hip.so (include/shafa_hip.h): lifecycle, batch contexts and the
// host-buffer one-block wrappers that stand where the reference calls block_compression / make_freq
// (f.c:248,310,325), compress_to_buffer (c.c:411) and the decompressors (d.c:342,735).
#include "common.hpp"
#include "internal.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <mutex>

// ------------------------------------------------------------------------------------------------
// error text (per calling thread: the reference's drivers call the per-block functions from one thread per block)
// ------------------------------------------------------------------------------------------------
static thread_local char g_last_error[512] = "";

int shafa_set_hip_error(hipError_t e, const char *what)
{
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
    return SHAFA_DEVICE_ERROR;
}

// ------------------------------------------------------------------------------------------------
// batch context
// ------------------------------------------------------------------------------------------------
int batch_enter(Batch *b, hipStream_t st)
{
    if (!b) return SHAFA_OUTSIDE_MODULE;
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != b->device) {
        snprintf(g_last_error, sizeof(g_last_error), "batch of device %d used while device %d is current", b->device, dev);
        return SHAFA_DEVICE_ERROR;
    }
    if (b->has_last && b->last_st != st) {
        const hipError_t e = hipStreamSynchronize(b->last_st);
        if (e == hipErrorInvalidHandle || e == hipErrorInvalidResourceHandle || e == hipErrorContextIsDestroyed)
            (void)hipGetLastError();       // the previous stream was destroyed by its owner (legal once its work is done): nothing to wait for
        else if (e != hipSuccess)          // a kernel or copy on it failed: the workspace and the error words cannot be trusted
            return shafa_set_hip_error(e, "hipStreamSynchronize(previous stream of the batch)");
    }
    b->last_st = st;
    b->has_last = true;
    return SHAFA_SUCCESS;
}

int batch_reserve(Batch *b, hipStream_t st, size_t bytes)
{
    if (bytes <= b->ws_bytes) return SHAFA_SUCCESS;
    if (b->d_ws) {
        HIP_TRY(hipStreamSynchronize(st));          // kernels of earlier launches on this batch's stream
        HIP_TRY(hipFree(b->d_ws));
        b->d_ws = nullptr;
        b->ws_bytes = 0;
    }
    const size_t want = bytes + bytes / 8 + 4096;
    HIP_TRY(hipMalloc(&b->d_ws, want));
    b->ws_bytes = want;
    return SHAFA_SUCCESS;
}

static void seg_release(StageSeg &s, bool wait)
{
    if (s.ev) {
        if (wait && s.sealed) (void)hipEventSynchronize(s.ev);
        (void)hipEventDestroy(s.ev);
        s.ev = nullptr;
    }
}

void *batch_stage(Batch *b, hipStream_t st, size_t bytes)
{
    bytes = (bytes + 63) & ~(size_t)63;
    if (!b->segs) b->segs = new std::vector<StageSeg>();
    std::vector<StageSeg> &segs = *b->segs;
    // seal the previous region: its copies were enqueued by the caller after it was handed out
    if (!segs.empty() && !segs.back().sealed) {
        if (hipEventRecord(segs.back().ev, segs.back().st) != hipSuccess) return nullptr;
        segs.back().sealed = true;
    }
    if (bytes > b->stage_bytes) {          // grow: wait for the copies that still read the old arena (their events only)
        for (StageSeg &s : segs) seg_release(s, true);
        segs.clear();
        if (b->h_stage) (void)hipHostFree(b->h_stage);
        b->h_stage = nullptr;
        b->stage_bytes = 0;
        const size_t want = 4 * bytes + (1 << 20);
        if (hipHostMalloc((void **)&b->h_stage, want, hipHostMallocDefault) != hipSuccess) return nullptr;
        b->stage_bytes = want;
        b->stage_used = 0;
    }
    size_t off = b->stage_used;
    if (off + bytes > b->stage_bytes) off = 0;                 // wrap
    // regions that overlap the new one must have been consumed: wait for exactly their events
    for (size_t i = 0; i < segs.size();) {
        StageSeg &s = segs[i];
        if (s.off < off + bytes && off < s.off + s.bytes) {
            seg_release(s, true);
            segs.erase(segs.begin() + (long)i);
        } else ++i;
    }
    StageSeg seg = {off, bytes, st, nullptr, false};
    if (hipEventCreateWithFlags(&seg.ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    segs.push_back(seg);
    b->stage_used = off + bytes;
    return b->h_stage + off;
}

u8 *batch_params_begin(Batch *b, size_t bytes)
{
    if (!b->copy_st) {
        if (hipStreamCreateWithFlags(&b->copy_st, hipStreamNonBlocking) != hipSuccess) return nullptr;
        for (int i = 0; i < 2; ++i)
            if (hipEventCreateWithFlags(&b->par_ready[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&b->par_free[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    const int i = b->par_turn;
    b->par_turn ^= 1;
    b->par_cur = i;
    b->par_inline = !b->par_dma && bytes <= PARAMS_INLINE_BYTES;   // the caller stages on the stream that will read the staging arena
    if (bytes > b->par_bytes[i]) {                     // grow: the kernels that read the old buffer must be through
        if (b->par_used[i] && hipEventSynchronize(b->par_free[i]) != hipSuccess) return nullptr;
        if (b->d_par[i]) (void)hipFree(b->d_par[i]);
        b->d_par[i] = nullptr;
        b->par_bytes[i] = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipMalloc(&b->d_par[i], want) != hipSuccess) return nullptr;
        b->par_bytes[i] = want;
    }
    return (u8 *)b->d_par[i];
}

// The parameter upload as a kernel that reads the pinned staging arena (host memory the device can address): above 16 KB
// hipMemcpyAsync hands a host-to-device copy to the SDMA engine, and a launch of 8..30 blocks then waited ~0.6 ms for its
// 17..60 KB of parameters (tools/dbg/small_launch_encode.sh: 8 x 64 MiB encoded in 790 us, 7 x 64 MiB in 190 us).
// Except in a pipe (layer 3): there the link is busy with the blocks themselves and a kernel's reads of host memory queue
// behind 64 MiB transfers (decode 41 -> 34 GiB/s, tools/dbg/pipe_rate_ab.sh), while the engine's latency hides behind the
// block's own transfer as long as the copy is NOT in the slot's stream (there it starts after the block has arrived: 33
// GiB/s as well): a pipe slot's batch (par_dma) keeps hipMemcpyAsync on the side stream.
__global__ __launch_bounds__(256) void param_copy_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// The small per-launch records of the other launchers (hist, rle_encode, rle_decode), in the launch's stream: the same
// kernel once the runtime would take the SDMA engine for them (more than 16 KB: hundreds of blocks), hipMemcpyAsync below
// that and in a pipe.  `src` is in the batch's pinned staging arena (64-byte units), `dst` 16-byte aligned with room for
// the last piece.
int batch_upload(Batch *b, hipStream_t st, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return SHAFA_SUCCESS;
    if (b->par_dma || bytes <= 16384 || (bytes & 15) || ((uintptr_t)dst & 15)) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        return SHAFA_SUCCESS;
    }
    const size_t n16 = bytes / 16, wgs = (n16 + 255) / 256;
    hipLaunchKernelGGL(param_copy_kernel, dim3((unsigned)(wgs < 512 ? wgs : 512)), dim3(256), 0, st, (uint4 *)dst, (const uint4 *)src, n16);
    HIP_TRY(hipGetLastError());
    return SHAFA_SUCCESS;
}

int batch_params_commit(Batch *b, hipStream_t st, const void *hs, size_t bytes)
{
    const int i = b->par_cur;
    const size_t n16 = (bytes + 15) / 16;              // the arena is handed out in 64-byte units, the device buffer has slack
    const size_t wgs = (n16 + 255) / 256;
    if (b->par_inline) {
        // a few blocks' worth (layer 3 submits one block per launch): in the launch's own stream, in order — a pipe's slots
        // already outnumber the hardware queues, and a side stream's kernel would wait behind another slot's kernels
        if (n16) {
            hipLaunchKernelGGL(param_copy_kernel, dim3((unsigned)wgs), dim3(256), 0, st, (uint4 *)b->d_par[i], (const uint4 *)hs, n16);
            HIP_TRY(hipGetLastError());
        }
        return SHAFA_SUCCESS;
    }
    if (b->par_used[i]) HIP_TRY(hipStreamWaitEvent(b->copy_st, b->par_free[i], 0));
    if (b->par_dma) {
        if (bytes) HIP_TRY(hipMemcpyAsync(b->d_par[i], hs, bytes, hipMemcpyHostToDevice, b->copy_st));
    } else if (n16) {
        hipLaunchKernelGGL(param_copy_kernel, dim3((unsigned)(wgs < 512 ? wgs : 512)), dim3(256), 0, b->copy_st,
                           (uint4 *)b->d_par[i], (const uint4 *)hs, n16);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(b->par_ready[i], b->copy_st));
    HIP_TRY(hipStreamWaitEvent(st, b->par_ready[i], 0));
    return SHAFA_SUCCESS;
}

int batch_params_done(Batch *b, hipStream_t st)
{
    HIP_TRY(hipEventRecord(b->par_free[b->par_cur], st));
    b->par_used[b->par_cur] = true;
    return SHAFA_SUCCESS;
}

void batch_stage_retire(Batch *b, hipStream_t st)
{
    if (!b->segs) return;
    std::vector<StageSeg> &segs = *b->segs;
    for (size_t i = 0; i < segs.size();) {
        if (segs[i].st == st) {
            seg_release(segs[i], false);
            segs.erase(segs.begin() + (long)i);
        } else ++i;
    }
    if (segs.empty()) b->stage_used = 0;
}

extern "C" {

int shafa_hip_abi_version(void) { return SHAFA_HIP_ABI_VERSION; }

int shafa_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *shafa_hip_last_error(void) { return g_last_error; }

int shafa_hip_set_option(const char *name, long value)
{
    if (name && !strcmp(name, "sf_encode_one_pass_min_blocks")) {
        sfenc_configure(value < 0 ? 0 : (value > (1 << 30) ? (1 << 30) : (int)value));      // 0: measured defaults
        return SHAFA_SUCCESS;
    }
    if (name && !strcmp(name, "sf_encode_window_bits")) {
        extern int g_sfe_window_bits;
        if (value < 0 || value > 16) return SHAFA_OUTSIDE_MODULE;
        g_sfe_window_bits = (int)value;
        return SHAFA_SUCCESS;
    }
    if (name && !strcmp(name, "sf_encode_lanes")) {
        extern int g_sfe_lanes;
        if (value != 0 && value != 256 && value != 512) return SHAFA_OUTSIDE_MODULE;
        g_sfe_lanes = (int)value;
        return SHAFA_SUCCESS;
    }
    if (name && !strcmp(name, "sf_decode_path")) {
        if (value < 0 || value > 2) return SHAFA_OUTSIDE_MODULE;
        sfdec_configure_path((int)value);
        return SHAFA_SUCCESS;
    }
    if (name && !strcmp(name, "rle_encode_general")) {
        rleenc_configure(value != 0);
        return SHAFA_SUCCESS;
    }
    if (name && !strcmp(name, "rle_encode_one_pass")) {
        rleenc_configure_one_pass(value != 0);
        return SHAFA_SUCCESS;
    }
    if (name && !strcmp(name, "sf_decode_speculate")) {
        sfdec_configure(value <= 0 ? 0 : value >= 2 ? 2 : 1);
        return SHAFA_SUCCESS;
    }
    return SHAFA_OUTSIDE_MODULE;
}

int shafa_hipd_batch_create(int max_blocks, size_t max_block_bytes, shafa_hipd_batch **out)
{
    if (!out || max_blocks <= 0) return SHAFA_OUTSIDE_MODULE;
    Batch *b = (Batch *)calloc(1, sizeof(Batch));
    if (!b) return SHAFA_LACK_OF_MEMORY;
    b->max_blocks = max_blocks;
    b->max_block_bytes = max_block_bytes;
    if (hipGetDevice(&b->device) != hipSuccess) { free(b); return SHAFA_DEVICE_ERROR; }
    b->h_hosterr = (int *)calloc((size_t)max_blocks, sizeof(int));
    if (!b->h_hosterr) { free(b); return SHAFA_LACK_OF_MEMORY; }
    hipError_t e = hipMalloc((void **)&b->d_err, (size_t)max_blocks * sizeof(int));
    if (e == hipSuccess) e = hipMemset(b->d_err, 0, (size_t)max_blocks * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&b->d_par_hist, (size_t)max_blocks * 48);
    if (e == hipSuccess) e = hipHostMalloc((void **)&b->h_err, (size_t)max_blocks * sizeof(int), hipHostMallocDefault);
    if (e != hipSuccess) {
        if (b->d_err) hipFree(b->d_err);
        free(b);
        return shafa_set_hip_error(e, "shafa_hipd_batch_create");
    }
    *out = (shafa_hipd_batch *)b;
    return SHAFA_SUCCESS;
}

void shafa_hipd_batch_destroy(shafa_hipd_batch *hb)
{
    Batch *b = (Batch *)hb;
    if (!b) return;
    DeviceGuard dg(b->device);                     // the batch's device, whatever the caller's current device is
    if (b->has_last) (void)hipStreamSynchronize(b->last_st);
    hipDeviceSynchronize();
    if (b->copy_st) {
        (void)hipStreamDestroy(b->copy_st);
        for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(b->par_ready[i]); (void)hipEventDestroy(b->par_free[i]); }
    }
    for (int i = 0; i < 2; ++i)
        if (b->d_par[i]) hipFree(b->d_par[i]);
    if (b->d_ws) hipFree(b->d_ws);
    if (b->segs) {
        for (StageSeg &sg : *b->segs) seg_release(sg, false);
        delete b->segs;
    }
    if (b->h_stage) hipHostFree(b->h_stage);
    if (b->d_err) hipFree(b->d_err);
    if (b->d_par_hist) hipFree(b->d_par_hist);
    if (b->h_err) hipHostFree(b->h_err);
    free(b->h_hosterr);
    free(b);
}

int shafa_hipd_hist256(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                       const uint64_t *h_in_off, const uint64_t *h_in_n, uint64_t *d_freq)
{
    if (int rc = batch_enter((Batch *)b, (hipStream_t)stream)) return rc;
    return hist_launch((Batch *)b, (hipStream_t)stream, nblocks, d_in, h_in_off, h_in_n, d_freq);
}

int shafa_hipd_sf_build_codes(shafa_hipd_batch *b, void *stream, int nblocks, const uint64_t *d_freq, shafa_code_table *d_tables)
{
    if (!d_freq || !d_tables) return SHAFA_OUTSIDE_MODULE;
    if (int rc = batch_enter((Batch *)b, (hipStream_t)stream)) return rc;
    return sftab_launch((Batch *)b, (hipStream_t)stream, nblocks, d_freq, d_tables);
}

size_t shafa_hip_tile_hist_bytes(size_t n) { return ((n + SHAFA_TILE_BYTES - 1) / SHAFA_TILE_BYTES) * 512; }

int shafa_hipd_hist256_tiles(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                             const uint64_t *h_in_off, const uint64_t *h_in_n, uint64_t *d_freq,
                             uint8_t *d_tile_hist, const uint64_t *h_tile_hist_off)
{
    if (!d_tile_hist || !h_tile_hist_off) return SHAFA_OUTSIDE_MODULE;
    if (int rc = batch_enter((Batch *)b, (hipStream_t)stream)) return rc;
    return hist_launch_dev((Batch *)b, (hipStream_t)stream, nblocks, d_in, h_in_off, h_in_n, nullptr, d_freq, d_tile_hist,
                           h_tile_hist_off);
}

int shafa_hipd_rle_encode_tiles(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                                const uint64_t *h_in_off, const uint64_t *h_in_n, uint8_t *d_out,
                                const uint64_t *h_out_off, const uint64_t *h_out_cap, uint64_t *d_out_n,
                                uint64_t *d_freq, uint8_t *d_tile_hist, const uint64_t *h_tile_hist_off)
{
    if (!d_freq || !d_tile_hist || !h_tile_hist_off) return SHAFA_OUTSIDE_MODULE;
    if (int rc = batch_enter((Batch *)b, (hipStream_t)stream)) return rc;
    return rleenc_launch((Batch *)b, (hipStream_t)stream, nblocks, d_in, h_in_off, h_in_n, d_out, h_out_off,
                         h_out_cap, d_out_n, d_freq, d_tile_hist, h_tile_hist_off);
}

int shafa_hipd_sf_encode_tiles(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                               const uint64_t *h_in_off, const uint64_t *h_in_n, const shafa_code_table *h_tables,
                               const uint8_t *d_tile_hist, const uint64_t *h_tile_hist_off, uint8_t *d_out,
                               const uint64_t *h_out_off, const uint64_t *h_out_cap, uint64_t *d_out_n)
{
    if (!d_tile_hist || !h_tile_hist_off) return SHAFA_OUTSIDE_MODULE;
    if (int rc = batch_enter((Batch *)b, (hipStream_t)stream)) return rc;
    return sfenc_launch((Batch *)b, (hipStream_t)stream, nblocks, d_in, h_in_off, h_in_n, h_tables, d_out,
                        h_out_off, h_out_cap, d_out_n, d_tile_hist, h_tile_hist_off);
}

int shafa_hipd_rle_encode(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                          const uint64_t *h_in_off, const uint64_t *h_in_n, uint8_t *d_out,
                          const uint64_t *h_out_off, const uint64_t *h_out_cap,
                          uint64_t *d_out_n, uint64_t *d_freq)
{
    if (int rc = batch_enter((Batch *)b, (hipStream_t)stream)) return rc;
    return rleenc_launch((Batch *)b, (hipStream_t)stream, nblocks, d_in, h_in_off, h_in_n, d_out, h_out_off,
                         h_out_cap, d_out_n, d_freq);
}

int shafa_hipd_sf_encode(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                         const uint64_t *h_in_off, const uint64_t *h_in_n,
                         const shafa_code_table *h_tables, uint8_t *d_out, const uint64_t *h_out_off,
                         const uint64_t *h_out_cap, uint64_t *d_out_n)
{
    if (int rc = batch_enter((Batch *)b, (hipStream_t)stream)) return rc;
    return sfenc_launch((Batch *)b, (hipStream_t)stream, nblocks, d_in, h_in_off, h_in_n, h_tables, d_out,
                        h_out_off, h_out_cap, d_out_n);
}

int shafa_hipd_sf_decode(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                         const uint64_t *h_in_off, const uint64_t *h_in_n,
                         const shafa_code_table *h_tables, const uint64_t *h_n_symbols,
                         uint8_t *d_out, const uint64_t *h_out_off)
{
    if (int rc = batch_enter((Batch *)b, (hipStream_t)stream)) return rc;
    return sfdec_launch((Batch *)b, (hipStream_t)stream, nblocks, d_in, h_in_off, h_in_n, h_tables,
                        h_n_symbols, d_out, h_out_off);
}

int shafa_hipd_rle_decode(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                          const uint64_t *h_in_off, const uint64_t *h_in_n, uint8_t *d_out,
                          const uint64_t *h_out_off, const uint64_t *h_out_cap, uint64_t *d_out_n)
{
    if (int rc = batch_enter((Batch *)b, (hipStream_t)stream)) return rc;
    return rledec_launch((Batch *)b, (hipStream_t)stream, nblocks, d_in, h_in_off, h_in_n, d_out, h_out_off,
                         h_out_cap, d_out_n);
}

}  // extern "C"

// the call itself (not a block) failed: no block has a result — every block reports the call's code
static int finish_failed(Batch *b, int nblocks, int *h_block_err, int rc)
{
    for (int i = 0; b && i < nblocks && i < b->max_blocks; ++i) {
        if (h_block_err) h_block_err[i] = rc;
        b->h_hosterr[i] = 0;
    }
    return rc;
}

int batch_finish(Batch *b, hipStream_t st, int nblocks, int *h_block_err, bool *call_failed)
{
    if (call_failed) *call_failed = true;
    if (int rc = batch_enter(b, st)) return finish_failed(b, nblocks, h_block_err, rc);
    if (nblocks > b->max_blocks) nblocks = b->max_blocks;
    if (nblocks < 0) nblocks = 0;
    hipError_t e = hipSuccess;
    if (nblocks) {
        e = hipMemcpyAsync(b->h_err, b->d_err, (size_t)nblocks * sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemsetAsync(b->d_err, 0, (size_t)nblocks * sizeof(int), st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return finish_failed(b, nblocks, h_block_err, shafa_set_hip_error(e, "shafa_hipd_finish"));
    if (call_failed) *call_failed = false;
    batch_stage_retire(b, st);
    if (b->copy_st) batch_stage_retire(b, b->copy_st);  // `st` waited for every parameter copy: they are through as well
    int first = SHAFA_SUCCESS;
    for (int i = 0; i < nblocks; ++i) {
        const int e = b->h_hosterr[i] ? b->h_hosterr[i] : b->h_err[i];
        b->h_hosterr[i] = 0;
        if (h_block_err) h_block_err[i] = e;
        if (!first && e) first = e;
    }
    if (first == SHAFA_DEVICE_ERROR)
        snprintf(g_last_error, sizeof(g_last_error), "kernel reported a lost predecessor tile (spin bound hit)");
    return first;
}

extern "C" {

int shafa_hipd_finish(shafa_hipd_batch *hb, void *stream, int nblocks, int *h_block_err)
{
    return batch_finish((Batch *)hb, (hipStream_t)stream, nblocks, h_block_err, nullptr);
}

int shafa_hipd_gen_bytes(void *stream, uint64_t seed, uint64_t first_index, const uint8_t *d_map65536,
                         uint8_t *d_out, size_t n)
{
    return gen_launch((hipStream_t)stream, seed, first_index, d_map65536, d_out, n);
}

// ------------------------------------------------------------------------------------------------
// layer 1: host buffers, one block per call, synchronous
// ------------------------------------------------------------------------------------------------
static struct {
    bool ready;
    int device;
    hipStream_t stream;
    Batch *batch;
    u8 *d_a; size_t a_bytes;     // input staging on the device
    u8 *d_b; size_t b_bytes;     // output staging on the device
    u64 *d_small;                // 256 u64 histogram + 1 u64 size
} g;

static int g_devs[64];
static int g_ndevs = 0;                     // 0: layer 3 uses the layer-1 device only

// Layer 1 keeps ONE stream, batch context and pair of staging buffers per process.  The reference calls the functions
// layer 1 replaces from one thread per block at the same time (multithread.c:70-87 with c.c:411, d.c:735), so every
// layer-1 entry point takes this lock for its whole duration: calls from several threads are safe and run one after the
// other on the GPU (a block's kernels fill the device anyway; overlap of copies and kernels is what layer 3 is for).
static std::recursive_mutex g_l1_mu;
#define L1_ENTER()                                   \
    std::lock_guard<std::recursive_mutex> l1_lock(g_l1_mu); \
    int rc = lazy_init();                            \
    if (rc) return rc;                               \
    DeviceGuard l1_dev(g.device)

static int ensure_dev(u8 **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return SHAFA_SUCCESS;
    if (*p) { HIP_TRY(hipFree(*p)); *p = nullptr; *cap = 0; }
    const size_t want = ((bytes + (bytes >> 3) + 4096) + 255) & ~(size_t)255;
    HIP_TRY(hipMalloc((void **)p, want));
    *cap = want;
    return SHAFA_SUCCESS;
}

int shafa_hip_init(int device)
{
    std::lock_guard<std::recursive_mutex> l1_lock(g_l1_mu);
    if (g.ready && g.device == device) return SHAFA_SUCCESS;
    if (g.ready) shafa_hip_shutdown();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        snprintf(g_last_error, sizeof(g_last_error), "no HIP device visible");
        return SHAFA_DEVICE_ERROR;
    }
    if (device < 0 || device >= n) return SHAFA_OUTSIDE_MODULE;
    DeviceGuard dg(device);                      // the caller's current device is left as it was
    if (const char *e = getenv("SHAFA_SF_ENCODE_ONE_PASS_MIN_BLOCKS")) shafa_hip_set_option("sf_encode_one_pass_min_blocks", atol(e));
    if (const char *e = getenv("SHAFA_SF_DECODE_SPECULATE")) shafa_hip_set_option("sf_decode_speculate", atol(e));
    HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    shafa_hipd_batch *bh = nullptr;
    int rc = shafa_hipd_batch_create(1, (size_t)1 << 27, &bh);
    if (rc) return rc;
    g.batch = (Batch *)bh;
    HIP_TRY(hipMalloc((void **)&g.d_small, 257 * sizeof(u64)));
    g.device = device;
    g.ready = true;
    return SHAFA_SUCCESS;
}

void shafa_hip_shutdown(void)
{
    std::lock_guard<std::recursive_mutex> l1_lock(g_l1_mu);
    if (!g.ready) return;
    DeviceGuard dg(g.device);
    hipDeviceSynchronize();
    shafa_hipd_batch_destroy((shafa_hipd_batch *)g.batch);
    if (g.d_a) hipFree(g.d_a);
    if (g.d_b) hipFree(g.d_b);
    if (g.d_small) hipFree(g.d_small);
    hipStreamDestroy(g.stream);
    memset(&g, 0, sizeof(g));
    g_ndevs = 0;
}


int shafa_hip_init_devices(const int *devices, int n_devices)
{
    std::lock_guard<std::recursive_mutex> l1_lock(g_l1_mu);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        snprintf(g_last_error, sizeof(g_last_error), "no HIP device visible");
        return SHAFA_DEVICE_ERROR;
    }
    if (n_devices < 0 || n_devices > 64 || (n_devices && !devices)) return SHAFA_OUTSIDE_MODULE;
    int list[64], m = 0;
    if (n_devices == 0) { for (; m < n && m < 64; ++m) list[m] = m; }
    else for (; m < n_devices; ++m) { if (devices[m] < 0 || devices[m] >= n) return SHAFA_OUTSIDE_MODULE; list[m] = devices[m]; }
    int rc = shafa_hip_init(list[0]);
    if (rc) return rc;
    memcpy(g_devs, list, sizeof(int) * (size_t)m);
    g_ndevs = m;
    return SHAFA_SUCCESS;
}

int shafa_hip_devices(void) { return g_ndevs ? g_ndevs : 1; }

static int lazy_init(void)
{
    std::lock_guard<std::recursive_mutex> l1_lock(g_l1_mu);
    return g.ready ? SHAFA_SUCCESS : shafa_hip_init(0);
}

static int upload(const uint8_t *in, size_t n)
{
    int rc = ensure_dev(&g.d_a, &g.a_bytes, n + 64);
    if (rc) return rc;
    if (n) HIP_TRY(hipMemcpyAsync(g.d_a, in, n, hipMemcpyHostToDevice, g.stream));
    return SHAFA_SUCCESS;
}

int shafa_hip_hist256(const uint8_t *in, size_t n, uint64_t freq[256])
{
    L1_ENTER();
    if ((rc = upload(in, n))) return rc;
    const u64 off[1] = {0}, len[1] = {n};
    if ((rc = hist_launch(g.batch, g.stream, 1, g.d_a, off, len, g.d_small))) return rc;
    HIP_TRY(hipMemcpyAsync(freq, g.d_small, 256 * sizeof(u64), hipMemcpyDeviceToHost, g.stream));
    return shafa_hipd_finish((shafa_hipd_batch *)g.batch, g.stream, 1, nullptr);
}

int shafa_hip_rle_encode(const uint8_t *in, size_t n, uint8_t *out, size_t out_cap, size_t *out_n,
                         uint64_t *freq_out)
{
    L1_ENTER();
    if (out_cap < 2 * n + 3) return SHAFA_LACK_OF_MEMORY;      // f.c:244 worst case
    if ((rc = upload(in, n))) return rc;
    const size_t cap = (2 * n + 3 + 15) & ~(size_t)15;
    if ((rc = ensure_dev(&g.d_b, &g.b_bytes, cap))) return rc;
    const u64 ioff[1] = {0}, ilen[1] = {n}, ooff[1] = {0}, ocap[1] = {cap};
    if ((rc = rleenc_launch(g.batch, g.stream, 1, g.d_a, ioff, ilen, g.d_b, ooff, ocap, g.d_small + 256,
                            freq_out ? g.d_small : nullptr))) return rc;
    u64 sz = 0;
    HIP_TRY(hipMemcpyAsync(&sz, g.d_small + 256, sizeof(u64), hipMemcpyDeviceToHost, g.stream));
    if (freq_out) HIP_TRY(hipMemcpyAsync(freq_out, g.d_small, 256 * sizeof(u64), hipMemcpyDeviceToHost, g.stream));
    rc = shafa_hipd_finish((shafa_hipd_batch *)g.batch, g.stream, 1, nullptr);
    if (rc) return rc;
    if (sz > out_cap) return SHAFA_LACK_OF_MEMORY;
    if (sz) HIP_TRY(hipMemcpy(out, g.d_b, sz, hipMemcpyDeviceToHost));
    *out_n = (size_t)sz;
    return SHAFA_SUCCESS;
}

int shafa_hip_sf_encode(const uint8_t *in, size_t n, const shafa_code_table *table,
                        uint8_t *out, size_t out_cap, size_t *out_n)
{
    L1_ENTER();
    if ((rc = upload(in, n))) return rc;
    const size_t cap = (out_cap + 15) & ~(size_t)15;
    if ((rc = ensure_dev(&g.d_b, &g.b_bytes, cap + 16))) return rc;
    // the device region is the caller's capacity (rounded down to keep the bound exact)
    const u64 ioff[1] = {0}, ilen[1] = {n}, ooff[1] = {0}, ocap[1] = {out_cap};
    if ((rc = sfenc_launch(g.batch, g.stream, 1, g.d_a, ioff, ilen, table, g.d_b, ooff, ocap, g.d_small + 256))) return rc;
    u64 sz = 0;
    HIP_TRY(hipMemcpyAsync(&sz, g.d_small + 256, sizeof(u64), hipMemcpyDeviceToHost, g.stream));
    rc = shafa_hipd_finish((shafa_hipd_batch *)g.batch, g.stream, 1, nullptr);
    if (rc) return rc;
    if (sz > out_cap) return SHAFA_LACK_OF_MEMORY;
    if (sz) HIP_TRY(hipMemcpy(out, g.d_b, sz, hipMemcpyDeviceToHost));
    *out_n = (size_t)sz;
    return SHAFA_SUCCESS;
}

int shafa_hip_sf_decode(const uint8_t *in, size_t in_n, const shafa_code_table *table,
                        uint8_t *out, size_t n_symbols)
{
    L1_ENTER();
    if ((rc = upload(in, in_n))) return rc;
    if ((rc = ensure_dev(&g.d_b, &g.b_bytes, n_symbols + 64))) return rc;
    const u64 ioff[1] = {0}, ilen[1] = {in_n}, ooff[1] = {0}, ns[1] = {n_symbols};
    if ((rc = sfdec_launch(g.batch, g.stream, 1, g.d_a, ioff, ilen, table, ns, g.d_b, ooff))) return rc;
    rc = shafa_hipd_finish((shafa_hipd_batch *)g.batch, g.stream, 1, nullptr);
    if (rc) return rc;
    if (n_symbols) HIP_TRY(hipMemcpy(out, g.d_b, n_symbols, hipMemcpyDeviceToHost));
    return SHAFA_SUCCESS;
}

int shafa_hip_rle_decode(const uint8_t *in, size_t in_n, uint8_t *out, size_t out_cap, size_t *out_n)
{
    L1_ENTER();
    if ((rc = upload(in, in_n))) return rc;
    size_t cap = out_cap < SHAFA_RLE_DECODE_MAX ? out_cap : SHAFA_RLE_DECODE_MAX;
    if ((rc = ensure_dev(&g.d_b, &g.b_bytes, cap + 64))) return rc;
    const u64 ioff[1] = {0}, ilen[1] = {in_n}, ooff[1] = {0}, ocap[1] = {cap};
    if ((rc = rledec_launch(g.batch, g.stream, 1, g.d_a, ioff, ilen, g.d_b, ooff, ocap, g.d_small + 256))) return rc;
    u64 sz = 0;
    HIP_TRY(hipMemcpyAsync(&sz, g.d_small + 256, sizeof(u64), hipMemcpyDeviceToHost, g.stream));
    rc = shafa_hipd_finish((shafa_hipd_batch *)g.batch, g.stream, 1, nullptr);
    if (rc) return rc;
    if (sz) HIP_TRY(hipMemcpy(out, g.d_b, sz, hipMemcpyDeviceToHost));
    *out_n = (size_t)sz;
    return SHAFA_SUCCESS;
}

}  // extern "C"

int api_lazy_init() { return lazy_init(); }     // layer 3 (pipe.hip)
// device of slot `slot` of a pipe with n_slots slots: the selected devices in turn, but no more devices than the pipe can
// keep busy with three blocks in flight each (a two-block file on an eight-GPU node opens one context, not eight)
int api_pipe_device(int slot, int n_slots)
{
    std::lock_guard<std::recursive_mutex> l1_lock(g_l1_mu);
    if (!g_ndevs) return g.device;
    int used = (n_slots + 2) / 3;
    if (used < 1) used = 1;
    if (used > g_ndevs) used = g_ndevs;
    return g_devs[slot % used];
}
