// internal.hpp — structures shared between the C-ABI layer (api.hip) and the kernel launchers.
#pragma once

#include "common.hpp"

#include <string.h>
#include <vector>

// one region of the pinned staging ring: [off, off + bytes), read by copies enqueued on `st`; `ev` is recorded on
// `st` when the next region is handed out (or at finish), i.e. after those copies
struct StageSeg {
    size_t off, bytes;
    hipStream_t st;
    hipEvent_t ev;
    bool sealed;
};

// Reusable per-batch context behind the opaque shafa_hipd_batch handle.
// parameter uploads of at most this many bytes go into the launch's stream (api.hip, batch_params_commit)
constexpr size_t PARAMS_INLINE_BYTES = 64 * 1024;

struct Batch {
    int device;            // the device the batch was created on: its workspace, error words and kernels live there
    hipStream_t last_st;   // the stream of the batch's last launch (a batch serves ONE stream at a time: a launch on a
    bool has_last;         //   different stream first waits for the previous one, see batch_enter)
    int max_blocks;
    size_t max_block_bytes;
    void *d_ws;            // device workspace (grow-only; grown outside timed regions by warm-up calls)
    size_t ws_bytes;
    u8 *h_stage;           // pinned host staging ring for parameter blocks / tables: every region handed out is
    size_t stage_bytes;    //   fenced by an event recorded on the stream that copies from it; a region is reused only
    size_t stage_used;     //   after ITS event (no device-wide synchronisation on the enqueue path)
    std::vector<StageSeg> *segs;
    void *d_par_hist;      // max_blocks * 48 B of hist256 parameters (separate from d_ws: hist256 may
                           //   run right after another op that still owns the workspace)
    int *d_err;            // one error code per block (first error wins)
    int *h_err;            // pinned mirror
    int *h_hosterr;        // errors found on the host while preparing a launch (malformed tables)
    // Parameter records and code tables of a launch travel on a SIDE stream into one of two device buffers, so that the
    // copy (and its hand-over between the copy engine and the compute queue, ~50-100 us either way) overlaps the kernels of
    // the launch before instead of sitting between two launches: batch_params_begin / _commit / _done.
    hipStream_t copy_st;
    void *d_par[2];
    size_t par_bytes[2];
    hipEvent_t par_ready[2], par_free[2];
    bool par_used[2];
    int par_turn, par_cur;
    bool par_inline;            // this launch's parameters are copied in the launch's own stream (few blocks), not on copy_st
    bool par_dma;               // the batch belongs to a pipe slot: parameters by hipMemcpyAsync on the side stream (api.hip)
};

// Every layer-2 entry point starts with this: checks that the calling thread's current device is the batch's, and
// serialises a change of stream (the workspace, the error words and the staging ring are shared by all launches of
// a batch, so work enqueued on another stream must have finished before the new stream's launch reuses them).
int batch_enter(Batch *b, hipStream_t st);
// shafa_hipd_finish, telling a failure of the call (stream / copy error: no block has a result, every block_err = the code)
// from the first block's error
int batch_finish(Batch *b, hipStream_t st, int nblocks, int *h_block_err, bool *call_failed);

// RAII: make `device` current for the calling thread, restore the caller's device on exit (a torch caller, or layer 1
// next to a multi-device pipe, must not see its current device change under it)
struct DeviceGuard {
    int prev;
    bool changed;
    explicit DeviceGuard(int device) : prev(-1), changed(false)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) changed = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// make sure the device workspace holds `bytes` (growing waits for `st`, the only stream that uses this batch's workspace)
int batch_reserve(Batch *b, hipStream_t st, size_t bytes);
// hand out `bytes` of the pinned staging ring for copies that the caller enqueues on `st`
void *batch_stage(Batch *b, hipStream_t st, size_t bytes);
// every copy enqueued on `st` so far has completed (the caller synchronised `st`): release its regions
void batch_stage_retire(Batch *b, hipStream_t st);

// Parameters of a launch (records + tables), uploaded off the launch stream:
//   d = batch_params_begin(b, bytes)          the device buffer the launch's parameters will live in (records may point into it)
//   hs = batch_stage(b, b->copy_st, bytes)    pinned staging, filled by the caller
//   batch_params_commit(b, st, hs, bytes)     copy on the side stream (after the kernels that last read this buffer); `st` waits for it
//   ... kernels on `st` ...
//   batch_params_done(b, st)                  the buffer is free again once these kernels have run
u8 *batch_params_begin(Batch *b, size_t bytes);
int batch_upload(Batch *b, hipStream_t st, void *dst, const void *src, size_t bytes);
int batch_params_commit(Batch *b, hipStream_t st, const void *hs, size_t bytes);
int batch_params_done(Batch *b, hipStream_t st);
// A launcher's early `return rc` after batch_params_begin must still mark the buffer's last reader: kernels already enqueued
// may be reading it, and the next launch that gets this buffer waits on par_free before it overwrites it.
struct ParamsScope {
    Batch *b;
    hipStream_t st;
    bool open;
    ParamsScope(Batch *b_, hipStream_t st_) : b(b_), st(st_), open(true) {}
    ~ParamsScope() { if (open) (void)batch_params_done(b, st); }
    int done() { open = false; return batch_params_done(b, st); }
    ParamsScope(const ParamsScope &) = delete;
    ParamsScope &operator=(const ParamsScope &) = delete;
};

// ---- per-op parameter records (device arrays) --------------------------------------------------
struct EncBlk {
    const u8 *in;
    u8 *out;
    u64 n;
    u64 out_cap;
    u64 *out_n;
    int *err;
    const void *lut;
    u32 desc_base;
    u32 n_tiles;
    u32 ticket;
    u32 pad;
    const void *thist;     // sfe6 (sf_encode6.hip): the block's tile histograms (256 x u16 per 32 KiB tile), else unused
};

// second chain (descriptors, tickets) and per-block flags of the one-pass encoder's encode-again fall-back (sf_encode4.hip)
struct SfeRedo {
    u64 *desc2;
    u32 *tickets2;
    u32 *redo;
};

// ---- launchers (one per reference function) ------------------------------------------------------
// Module T on the device (sf_tables.hip): nblocks x 256 counts -> nblocks tables, both in device memory
int sftab_launch(Batch *bt, hipStream_t st, int nblocks, const u64 *d_freq, shafa_code_table *d_tables);
int hist_launch(Batch *bt, hipStream_t st, int nblocks, const u8 *d_in, const u64 *h_in_off,
                const u64 *h_in_n, u64 *d_freq);
// d_thist / h_thist_off (both or neither): also write every 32 KiB tile's own histogram (256 x u16) to d_thist + h_thist_off[b]
int hist_launch_dev(Batch *bt, hipStream_t st, int nblocks, const u8 *d_in, const u64 *h_in_off,
                    const u64 *h_in_n, const u64 *d_n, u64 *d_freq, u8 *d_thist = nullptr, const u64 *h_thist_off = nullptr);
// d_thist / h_thist_off (both or neither): the tile histograms of shafa_hipd_hist256_tiles, block b's at d_thist + h_thist_off[b]
int sfenc_launch(Batch *bt, hipStream_t st, int nblocks, const u8 *d_in, const u64 *h_in_off,
                 const u64 *h_in_n, const shafa_code_table *h_tables, u8 *d_out, const u64 *h_out_off,
                 const u64 *h_out_cap, u64 *d_out_n, const u8 *d_thist = nullptr, const u64 *h_thist_off = nullptr);
int sfdec_launch(Batch *bt, hipStream_t st, int nblocks, const u8 *d_in, const u64 *h_in_off,
                 const u64 *h_in_n, const shafa_code_table *h_tables, const u64 *h_n_symbols, u8 *d_out,
                 const u64 *h_out_off);
int rleenc_launch(Batch *bt, hipStream_t st, int nblocks, const u8 *d_in, const u64 *h_in_off,
                  const u64 *h_in_n, u8 *d_out, const u64 *h_out_off, const u64 *h_out_cap, u64 *d_out_n,
                  u64 *d_freq, u8 *d_thist = nullptr, const u64 *h_thist_off = nullptr);
int rledec_launch(Batch *bt, hipStream_t st, int nblocks, const u8 *d_in, const u64 *h_in_off,
                  const u64 *h_in_n, u8 *d_out, const u64 *h_out_off, const u64 *h_out_cap, u64 *d_out_n);
void sfenc_configure(int sfe4_min_blocks);
void sfdec_configure(int speculate);
void sfdec_configure_path(int path);
void rleenc_configure(int force_general);
void rleenc_configure_one_pass(int on);
int gen_launch(hipStream_t st, u64 seed, u64 first, const u8 *d_map, u8 *d_out, size_t n);
