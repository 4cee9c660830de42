/*
 * shafa_hip.h — C-ABI of libshafa_hip.so: the MI355X (gfx950) implementation of Shafa's per-block
 * hot path (Modules F, C, D).  Plain C, plain pointers and sizes, no C++/torch types.
 *
 * The reference (Fytex/Shafa-CD) has no plugin/FFI interface; its seam is the per-block
 * `process` callback handed to multithread_create (utils/multithread.h:45) and the direct calls
 * in f.c.  Each entry point below replaces one of those static functions (cited per function,
 * paths relative to /root/reference/src/modules/).  Return values are the reference's
 * _modules_error numbers (utils/errors.h:5-16) plus SHAFA_DEVICE_ERROR.
 *
 * Two layers:
 *   1. host-buffer, one block per call, synchronous  (shafa_hip_*)   — what the C host's
 *      f/c/d drivers call in place of f.c:248,310,325, c.c:411, d.c:342,735.
 *   2. device-resident, many blocks per launch, asynchronous on a caller stream (shafa_hipd_*)
 *      — used by the streaming drivers, bench.py and the multi-GPU sharding; every pointer named
 *      d_* is a DEVICE pointer, every h_* a HOST pointer.
 *
 * Threading.  The reference calls the functions layer 1 replaces from one pthread per block at the same time
 * (utils/multithread.c:70-87 starts `process` per block; c.c:411 compress_to_buffer, d.c:735 process_shafa_decomp), so
 * LAYER 1 IS THREAD-SAFE: every shafa_hip_* entry point may be called from any number of host threads concurrently.
 * Layer 1 keeps one stream and one pair of staging buffers per process and serialises the calls on an internal lock
 * (one block's kernels fill the GPU; overlap of copies, kernels and I/O is what layer 3 is for).
 * Layers 2 and 3: a batch / a pipe is used by one thread at a time; different batches and pipes may be used from different
 * threads concurrently.  A batch serves one stream at a time: a launch on another stream first waits for the batch's
 * previous stream.  The calling thread's current HIP device must be the batch's device (the one current at
 * shafa_hipd_batch_create) for layer-2 calls; layer 1 and layer 3 select their devices themselves and NO entry point
 * leaves the calling thread's current device changed.  shafa_hip_set_option() and shafa_hip_init*() are configuration:
 * call them while no other call is in flight.  shafa_hip_last_error() is per calling thread.  A decode call of 16 blocks
 * or more prepares its tables on up to seven short-lived helper threads of its own (joined before the call returns).
 * No entry point throws or exits.
 */
#ifndef SHAFA_HIP_H
#define SHAFA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHAFA_HIP_ABI_VERSION 8

/* utils/errors.h:5-16 (_modules_error), same numbers */
enum shafa_error {
    SHAFA_SUCCESS = 0,
    SHAFA_OUTSIDE_MODULE = 1,
    SHAFA_LACK_OF_MEMORY = 2,
    SHAFA_FILE_INACCESSIBLE = 3,
    SHAFA_FILE_UNRECOGNIZABLE = 4,
    SHAFA_FILE_STREAM_FAILED = 5,
    SHAFA_FILE_TOO_SMALL = 6,
    SHAFA_THREAD_CREATION_FAILED = 7,
    SHAFA_THREAD_TERMINATION_FAILED = 8,
    SHAFA_DEVICE_ERROR = 9            /* HIP runtime failure / no GPU; see shafa_hip_last_error() */
};

/* One block's Shannon-Fano table, the binary form of one "c0;c1;...;c255" .cod block
 * (t.c:353-361; parsed by c.c:115-177 and d.c:466-504): len[s] = code length in bits
 * (0 = symbol absent, max 255), bits[s] = the code MSB-first, zero padded. */
typedef struct shafa_code_table {
    uint8_t len[256];
    uint8_t bits[256][32];
} shafa_code_table;

/* RLE decode output limit of the reference: 64 MiB + 1 KiB (d.c:129-169). */
#define SHAFA_RLE_DECODE_MAX ((size_t)67108864 + 1024)

/* ------------------------------------------------------------------ lifecycle */
int shafa_hip_abi_version(void);
int shafa_hip_device_count(void);            /* 0 when no GPU is visible; never fails */
int shafa_hip_init(int device);              /* select device, create the library's stream/workspace */
/* Select the devices of the block pipeline (layer 3): slot i of every pipe created afterwards lives on
 * devices[i % n_devices] (its stream, buffers and kernels), so the blocks of one file are spread over the GPUs of the
 * node while the caller still retires them in order (multithread.c:70-87) — of a pipe with n_slots slots only the first
 * ceil(n_slots / 3) selected devices are used (three blocks in flight keep a GPU busy; a short file does not open a
 * context on every GPU).  Layers 1 and 2 use devices[0].
 * n_devices == 0 selects every visible device.  Returns SHAFA_OUTSIDE_MODULE for an invalid device number. */
int shafa_hip_init_devices(const int *devices, int n_devices);
int shafa_hip_devices(void);                 /* number of devices selected for layer 3 (1 until shafa_hip_init_devices) */
void shafa_hip_shutdown(void);
const char *shafa_hip_last_error(void);      /* text of the last SHAFA_DEVICE_ERROR */

/* Tuning knobs (no reference counterpart).  Unknown names return SHAFA_OUTSIDE_MODULE.
 *   "sf_encode_one_pass_min_blocks": a shafa_hipd_sf_encode launch with at least this many blocks of <= 16-bit
 *       codes takes the one-pass encoder, smaller launches the count/scan/pack kernels; 0 (default) = the measured crossovers: 6 blocks
 *       where the 1024-lane form runs (every code <= 16 bits since round 3), 80 for the 256-lane form.
 *   "sf_decode_speculate": 1 (default) lets blocks whose code re-synchronises take the speculative entry kernels of
 *       the Shannon-Fano decoder (verified exactly; falls back to the exact kernels per block), 0 = exact kernels only,
 *       2 = speculate for every block the kernels apply to, whatever its code (for tests of the fall-back).
 *   "sf_encode_lanes": 0 (default) = the widest workgroup whose windows fit the CU's LDS (1024 lanes, 32 KiB tiles, for
 *       codes of <= 12 bits; 256 lanes otherwise), 256 / 512 = that width.
 * Test knobs that force the fall-back kernels the library otherwise takes by itself (tests/test_gpu_codec.py):
 *   "sf_decode_path": 0 (default) = the fastest kernels the tables allow, 1 = one code per look-up as for incomplete
 *       codes, 2 = the generic byte-map kernels of codes longer than 32 bits.
 *   "rle_encode_general": 1 = every tile takes the per-element general RLE code (long runs, ragged tiles), 0 = by data.
 *   "rle_encode_one_pass": 1 = block_compression and the histogram of its output in ONE pass over the input (chained
 *       32 KiB super-tiles, rle_encode.hip rle4_kernel: 1.1 x the algorithmic HBM bytes instead of 2.4 x — for callers whose
 *       memory system is the scarce resource), 0 (default) = the two-pass kernels (rle3_*) and a separate histogram pass,
 *       which are faster when the GPU is otherwise idle.
 *   "sf_encode_window_bits": bits per symbol the 1024-lane encoder's LDS windows are sized for; 0 (default) = the launch's
 *       longest code, at most 12.  A tile that needs more is not placed: its block is flagged on the device and encoded
 *       again by the 256-lane form in the same call (what happens to 13..16-bit codes whose rare symbols fill a whole
 *       32 KiB tile); small values make ordinary data take that path.
 * shafa_hip_init() reads the environment variables SHAFA_SF_ENCODE_ONE_PASS_MIN_BLOCKS and SHAFA_SF_DECODE_SPECULATE
 * once for the first two knobs. */
int shafa_hip_set_option(const char *name, long value);

/* ------------------------------------------------------------------ layer 1: host buffers, one block */

/* make_freq (f.c:63-79): 256-bin byte histogram, 64-bit bins. */
int shafa_hip_hist256(const uint8_t *in, size_t n, uint64_t freq[256]);

/* block_compression (f.c:29-55) [+ make_freq of the RLE bytes, f.c:310, when freq_out != NULL].
 * out_cap must be >= 2n+3 (f.c:244) or the worst case is refused with SHAFA_LACK_OF_MEMORY. */
int shafa_hip_rle_encode(const uint8_t *in, size_t n, uint8_t *out, size_t out_cap, size_t *out_n,
                         uint64_t *freq_out /* [256] or NULL */);

/* compress_to_buffer + binary_coding (c.c:52-237): concatenate the block's codes MSB-first,
 * zero-pad the last byte; *out_n = ceil(bits/8).  A data symbol with an empty code in a table that
 * holds non-empty codes is SHAFA_FILE_UNRECOGNIZABLE; an all-empty table gives 0 bytes (c.c:156).
 * A block with both faults — such a symbol AND an output that does not fit out_cap — is SHAFA_FILE_UNRECOGNIZABLE
 * (the symbols are looked at first), whichever the device notices first. */
int shafa_hip_sf_encode(const uint8_t *in, size_t n, const shafa_code_table *table,
                        uint8_t *out, size_t out_cap, size_t *out_n);

/* create_tree + shafa_block_decompressor (d.c:466-551): decode exactly n_symbols symbols.
 * Malformed table / missing branch / exhausted input is SHAFA_FILE_UNRECOGNIZABLE. */
int shafa_hip_sf_decode(const uint8_t *in, size_t in_n, const shafa_code_table *table,
                        uint8_t *out, size_t n_symbols);

/* rle_block_decompressor (d.c:116-197); more than SHAFA_RLE_DECODE_MAX bytes of output is
 * SHAFA_FILE_UNRECOGNIZABLE (d.c:165-168), more than out_cap is SHAFA_LACK_OF_MEMORY, a {0, symbol, count} cut by the end
 * of the block is SHAFA_FILE_UNRECOGNIZABLE.  A block with several of these reports the one a front-to-back decoder stops at
 * (the first token whose end passes out_cap or the maximum; a cut triple comes last), independent of timing. */
int shafa_hip_rle_decode(const uint8_t *in, size_t in_n, uint8_t *out, size_t out_cap, size_t *out_n);

/* ------------------------------------------------------------------ layer 2: device buffers, batches
 *
 * A batch is nblocks independent blocks.  Block b's input is the h_in_n[b] bytes at
 * d_in + h_in_off[b]; its output region is the h_out_cap[b] bytes at d_out + h_out_off[b]
 * (all four are host arrays of nblocks entries; every offset must be a multiple of 16).
 * Calls enqueue work on `stream` (a hipStream_t passed as void*; NULL = the null stream) and
 * return without synchronising; results in device memory are valid after the stream is synchronised.
 * shafa_hipd_finish() synchronises the stream and returns the batch's first error, filling the
 * optional host arrays.
 */
typedef struct shafa_hipd_batch shafa_hipd_batch;

/* Allocate a reusable batch context (device workspace, pinned staging) for up to max_blocks blocks
 * of up to max_block_bytes input bytes each (for RLE/SF decode: of the LARGER of input and output).
 * The device workspace grows on demand and is kept: the largest users are RLE encode (about 0.26 bytes per input
 * byte of a launch) and SF decode (about 0.1 bytes per byte of stream; 0.35 on the exact path of codes that do not
 * re-synchronise). */
int shafa_hipd_batch_create(int max_blocks, size_t max_block_bytes, shafa_hipd_batch **out);
void shafa_hipd_batch_destroy(shafa_hipd_batch *b);

/* make_freq per block: d_freq[b*256 + s], 64-bit counts. */
int shafa_hipd_hist256(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                       const uint64_t *h_in_off, const uint64_t *h_in_n, uint64_t *d_freq);

/* Module T's core on the device (t.c:74-210, the rule set of host/sfcodes.c shafa_sf_build_codes): d_freq = nblocks x 256
 * counts (what shafa_hipd_hist256 / _rle_encode leave), d_tables = nblocks tables in DEVICE memory, bit-identical to the
 * host's for counts whose sum fits 64 bits (a block's own histogram always does; else SHAFA_OUTSIDE_MODULE for the block
 * and an empty table).  One workgroup per block.  The encoder's and decoder's entry points take their tables from the
 * HOST (their launchers choose kernels by the longest code): this is for callers that keep histograms and tables on the
 * GPU, and the device-side half of a host-free F -> T -> C. */
int shafa_hipd_sf_build_codes(shafa_hipd_batch *b, void *stream, int nblocks, const uint64_t *d_freq,
                              shafa_code_table *d_tables);

/* block_compression per block; d_out_n[b] = RLE size; d_freq (may be NULL) = histogram of the RLE bytes. */
int shafa_hipd_rle_encode(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                          const uint64_t *h_in_off, const uint64_t *h_in_n, uint8_t *d_out,
                          const uint64_t *h_out_off, const uint64_t *h_out_cap,
                          uint64_t *d_out_n, uint64_t *d_freq);

/* binary_coding per block; d_out_n[b] = ceil(bits/8). */
int shafa_hipd_sf_encode(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                         const uint64_t *h_in_off, const uint64_t *h_in_n,
                         const shafa_code_table *h_tables, uint8_t *d_out, const uint64_t *h_out_off,
                         const uint64_t *h_out_cap, uint64_t *d_out_n);

/* ---- Tile histograms: Module F's by-product that lets Module C run without any tile waiting for another ----------------
 * The reference's F -> T -> C sequence reads a block three times: make_freq (f.c:63-79), then — with the codes of Module T —
 * binary_coding (c.c:52-83), whose output position of byte i depends on the code lengths of all bytes before it.  A caller
 * that keeps a block resident in HBM between F and C can hand C what F already saw: the histogram of every
 * SHAFA_TILE_BYTES (32 KiB) tile of the block, 256 x uint16_t per tile (a tile holds at most 32768 of one byte value),
 * shafa_hip_tile_hist_bytes(n) bytes for a block of n bytes, tile t's counts at offset 512 t.  With them the bit offset of
 * every tile in the encoded block is (tile histograms . code lengths), scanned — known BEFORE the encoder starts — and the
 * encoder is a one-shot grid of independent workgroups (sf_encode6.hip) instead of a chained scan.  The histograms depend
 * on the data only, not on the codes.  Block b's tile histograms live at d_tile_hist + h_tile_hist_off[b] (offsets
 * multiples of 16).  Results are identical to the entry points without `_tiles`; histograms that are not the block's own
 * are detected (SHAFA_OUTSIDE_MODULE for the block, nothing is written outside its output region). */
#define SHAFA_TILE_BYTES 32768
size_t shafa_hip_tile_hist_bytes(size_t n);

/* make_freq per block as shafa_hipd_hist256, plus the tile histograms. */
int shafa_hipd_hist256_tiles(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                             const uint64_t *h_in_off, const uint64_t *h_in_n, uint64_t *d_freq,
                             uint8_t *d_tile_hist, const uint64_t *h_tile_hist_off);

/* block_compression + make_freq of the RLE bytes as shafa_hipd_rle_encode (d_freq required), plus the tile histograms of
 * the RLE bytes: block b's region must hold shafa_hip_tile_hist_bytes(h_out_cap[b]) bytes. */
int shafa_hipd_rle_encode_tiles(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                                const uint64_t *h_in_off, const uint64_t *h_in_n, uint8_t *d_out,
                                const uint64_t *h_out_off, const uint64_t *h_out_cap, uint64_t *d_out_n,
                                uint64_t *d_freq, uint8_t *d_tile_hist, const uint64_t *h_tile_hist_off);

/* binary_coding per block as shafa_hipd_sf_encode, given the tile histograms of the input blocks. */
int shafa_hipd_sf_encode_tiles(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                               const uint64_t *h_in_off, const uint64_t *h_in_n, const shafa_code_table *h_tables,
                               const uint8_t *d_tile_hist, const uint64_t *h_tile_hist_off, uint8_t *d_out,
                               const uint64_t *h_out_off, const uint64_t *h_out_cap, uint64_t *d_out_n);

/* shafa_block_decompressor per block: block b decodes h_n_symbols[b] symbols from the h_in_n[b]
 * bytes at d_in + h_in_off[b] into d_out + h_out_off[b] (which must hold h_n_symbols[b] bytes). */
int shafa_hipd_sf_decode(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                         const uint64_t *h_in_off, const uint64_t *h_in_n,
                         const shafa_code_table *h_tables, const uint64_t *h_n_symbols,
                         uint8_t *d_out, const uint64_t *h_out_off);

/* rle_block_decompressor per block; d_out_n[b] = decoded size. */
int shafa_hipd_rle_decode(shafa_hipd_batch *b, void *stream, int nblocks, const uint8_t *d_in,
                          const uint64_t *h_in_off, const uint64_t *h_in_n, uint8_t *d_out,
                          const uint64_t *h_out_off, const uint64_t *h_out_cap, uint64_t *d_out_n);

/* Synchronise `stream`, return the first per-block error of the calls enqueued since the last
 * finish (SHAFA_SUCCESS if none).  h_block_err (nblocks ints, may be NULL) receives every block's code. */
int shafa_hipd_finish(shafa_hipd_batch *b, void *stream, int nblocks, int *h_block_err);

/* Synthetic byte streams for bench/tests (no reference counterpart; same stream as the oracle's
 * orc_gen_bytes): byte i = map[r16(seed, first_index + i)] or r16 >> 8 when d_map65536 == NULL. */
int shafa_hipd_gen_bytes(void *stream, uint64_t seed, uint64_t first_index,
                         const uint8_t *d_map65536, uint8_t *d_out, size_t n);

/* =================================================================================================
 * Layer 3 — bounded in-order block pipeline (host buffers, asynchronous).
 *
 * Stands where the reference's drivers start one thread per block and join them in order
 * (multithread.c:126-194 create, :70-87 ordered write chain; callers f.c:231-356, c.c:383-411,
 * d.c:330-345, d.c:690-740).  A pipe owns n_slots slots; each slot has a HIP stream, pinned host
 * input/output buffers and device buffers.  Usage per block: fill shafa_pipe_in(slot), submit, and
 * later wait on the slots in submission order.  Blocks in different slots overlap: host read,
 * H2D copy, kernels, D2H copy and host write of neighbouring blocks run concurrently, with at
 * most n_slots blocks in flight (the reference's read-ahead is unbounded, c.c:383).
 * ================================================================================================= */
typedef struct shafa_pipe shafa_pipe;

enum shafa_pipe_op {
    SHAFA_OP_HIST = 1,           /* make_freq of the input (f.c:325)                                  */
    SHAFA_OP_RLE_ENCODE = 2,     /* block_compression + make_freq of the RLE bytes (f.c:248,310)      */
    SHAFA_OP_SF_ENCODE = 3,      /* compress_to_buffer (c.c:91-237)                                   */
    SHAFA_OP_SF_DECODE = 4,      /* create_tree + shafa_block_decompressor (d.c:565-569)              */
    SHAFA_OP_RLE_DECODE = 5,     /* rle_block_decompressor (d.c:116-197)                              */
    SHAFA_OP_SF_RLE_DECODE = 6,  /* process_shafa_decomp with RLE (d.c:558-590), fused on the device  */
    SHAFA_OP_FTC = 7             /* Modules F and C on ONE residency of the block, Module T on the host in between:
                                    see "F -> T -> C" below                                                            */
};
#define SHAFA_PIPE_INPUT_HIST 1  /* with SHAFA_OP_RLE_ENCODE / SHAFA_OP_FTC: also the histogram of the input (-c f)  */
#define SHAFA_PIPE_FTC_RLE 2     /* SHAFA_OP_FTC: run block_compression (the block may be encoded from its RLE bytes) */
#define SHAFA_PIPE_FTC_PLAIN 4   /* SHAFA_OP_FTC: the block may be encoded as it is (tile histograms of the input)    */

typedef struct shafa_pipe_result {
    const uint8_t *out;          /* the slot's pinned result buffer (valid until the slot is reused)  */
    size_t out_n;                /* result bytes                                                      */
    size_t mid_n;                /* SF_RLE_DECODE: bytes after the SF stage                           */
    uint64_t freq[256];          /* HIST: of the input; RLE_ENCODE: of the RLE bytes                  */
    uint64_t freq_in[256];       /* RLE_ENCODE with SHAFA_PIPE_INPUT_HIST: of the input               */
} shafa_pipe_result;

int shafa_pipe_create(int n_slots, shafa_pipe **out);
void shafa_pipe_destroy(shafa_pipe *p);
int shafa_pipe_slots(const shafa_pipe *p);
int shafa_pipe_slot_device(const shafa_pipe *p, int slot);      /* the device the slot's work runs on */

/* Pinned input buffer of an idle slot, grown to hold `bytes`; NULL if the slot is busy or on failure. */
uint8_t *shafa_pipe_in(shafa_pipe *p, int slot, size_t bytes);

/* Enqueue `op` on the first in_n bytes of the slot's input buffer and return without waiting.
 * table: SF ops; n_symbols: SF decodes; out_cap: SF_ENCODE result capacity (the other ops size
 * their own results: 2n+3 for RLE_ENCODE, 64 MiB + 1 KiB for RLE decodes).  Errors of the block,
 * including malformed tables, are reported by shafa_pipe_wait so that they surface in block order. */
int shafa_pipe_submit(shafa_pipe *p, int slot, int op, size_t in_n, const shafa_code_table *table,
                      size_t n_symbols, size_t out_cap, int flags);

/* Wait for the slot's block, fetch its result into the pinned output buffer, mark the slot idle.
 * Returns the block's _modules_error number. */
int shafa_pipe_wait(shafa_pipe *p, int slot, shafa_pipe_result *res);

/* ---- F -> T -> C on one residency (the default `shafa file -b m|M`, shafa.c:293-298 with 157-198) ----------------------
 * The reference's default run reads a block in Module F, writes its .rle, and reads that again in Module C.  Here the block
 * is uploaded ONCE:
 *   shafa_pipe_submit(p, slot, SHAFA_OP_FTC, in_n, NULL, 0, 0, flags)   F on the device: with SHAFA_PIPE_FTC_RLE
 *        block_compression + make_freq of the RLE bytes (shafa_hipd_rle_encode_tiles: the 32 KiB tile histograms of the RLE
 *        bytes stay on the device), with SHAFA_PIPE_FTC_PLAIN / SHAFA_PIPE_INPUT_HIST make_freq of the input and its tile
 *        histograms (block 0, whose RLE size decides for the file, f.c:250-258, asks for both)
 *   shafa_pipe_wait(p, slot, &res)          res.out / out_n = the RLE bytes (for the .rle file), res.freq / freq_in; the
 *        slot stays reserved: the caller builds the block's codes (Module T) from the histogram of what will be encoded
 *   shafa_pipe_ftc_encode(p, slot, use_rle, &table, out_cap)   Module C from the bytes already on the device
 *        (shafa_hipd_sf_encode_tiles with the tile histograms of stage one)
 *   shafa_pipe_wait(p, slot, &res)          res.out / out_n = the .shaf payload (a second pinned buffer: the RLE bytes of
 *        the first wait stay valid until the slot is reused); the slot is idle again. */
int shafa_pipe_ftc_encode(shafa_pipe *p, int slot, int use_rle, const shafa_code_table *table, size_t out_cap);

/* ---- Groups: several consecutive blocks of a file in one slot -------------------------------------------------------------
 * One launch per block costs the submitting thread ~0.1 ms whatever the block's size, which is all of a file's time at the
 * reference's default block size (64 KiB, file.h).  A group puts up to SHAFA_PIPE_GROUP_MAX blocks into one slot: the caller
 * lays their inputs out in shafa_pipe_in(slot, total) at offsets that are multiples of 16, every kernel is launched once for
 * all of them, and the results come back together.  Same ops, same per-block results and error codes as the single-block
 * calls; a slot holds either a single block or a group. */
#define SHAFA_PIPE_GROUP_MAX 256

typedef struct shafa_pipe_block {
    size_t in_off, in_n;                 /* the block inside the slot's input buffer                               */
    const shafa_code_table *table;       /* SF ops                                                                 */
    size_t n_symbols;                    /* SF decodes                                                             */
    size_t out_cap;                      /* SF_ENCODE result capacity                                              */
} shafa_pipe_block;

int shafa_pipe_submit_group(shafa_pipe *p, int slot, int op, int nblocks, const shafa_pipe_block *blocks, int flags);

/* res[i] / block_rc[i]: block i's result and its _modules_error number (results of blocks behind a failed block are still
 * valid: the caller decides where to stop, in block order).  The return value is an error of the call itself. */
int shafa_pipe_wait_group(shafa_pipe *p, int slot, int nblocks, shafa_pipe_result *res, int *block_rc);

#ifdef __cplusplus
}
#endif
#endif /* SHAFA_HIP_H */
