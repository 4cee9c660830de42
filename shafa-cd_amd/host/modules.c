/*
 * modules.c — the five module entry points of the reference (f.h:16, t.h:11, c.h:11, d.h:14,22),
 * same names, arguments, path ownership and error numbers, with every per-block computation sent
 * to the GPU through the C-ABI of libshafa_hip.so:
 *
 *     reference call site                     here (ops of the layer-3 block pipeline, include/shafa_hip.h)
 *     f.c:248  block_compression          ->  SHAFA_OP_RLE_ENCODE   (+ fused make_freq, f.c:310)
 *     f.c:325  make_freq                  ->  SHAFA_OP_HIST / SHAFA_PIPE_INPUT_HIST
 *     c.c:411  compress_to_buffer         ->  SHAFA_OP_SF_ENCODE
 *     d.c:735  process_shafa_decomp       ->  SHAFA_OP_SF_DECODE / SHAFA_OP_SF_RLE_DECODE
 *     d.c:342  rle_block_decompressor     ->  SHAFA_OP_RLE_DECODE
 *
 * Where the reference starts a thread per block and joins them in order (multithread.c), the
 * drivers here keep PIPE_SLOTS blocks in flight: block b is read straight into a pinned buffer and
 * submitted, and the oldest block is retired (waited for and written) when the slots are full.
 * --no-multithread keeps one block in flight, like the reference's sequential mode.
 *
 * The host keeps what the reference's drivers do around those calls: block splitting, the block-0
 * RLE decision, the '@'-framed text files, ordered writes.  There is no CPU implementation of the
 * kernels here: without a GPU every module returns SHAFA_DEVICE_ERROR.
 */
#include "shafa_host.h"

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

bool NO_MULTITHREAD = false;
bool SHAFA_VERBOSE = true;

/* SHAFA_TRACE=1: stage timings on stderr (diagnostics only) */
static void trace(const char *what, double t0)
{
    static int on = -1;
    if (on < 0) on = getenv("SHAFA_TRACE") != NULL;
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    if (on) fprintf(stderr, "[trace] %-28s %9.2f ms\n", what, ts.tv_sec * 1e3 + ts.tv_nsec / 1e6 - t0);
}

enum { PIPE_SLOTS_PER_DEVICE = 3, PIPE_SLOTS = 64 };        /* PIPE_SLOTS: upper bound (ticket arrays) */
/* three blocks in flight per selected GPU (shafa_hip_init_devices): slot i works on device i mod n, results are retired
 * in submission order by the one writer thread as before */
/* One pipe is kept between the modules of a process (the default `shafa file` runs F, T and C one after the other, each
 * with the same three slots): its streams, pinned and device buffers are what a module's first blocks paid 150 ms for.
 * A module that fails destroys its pipe (slots may still be busy); shafa_host_release() drops the kept one. */
static shafa_pipe *g_kept_pipe = NULL;
static int pipe_get(int depth, shafa_pipe **out)
{
    if (g_kept_pipe && shafa_pipe_slots(g_kept_pipe) == depth) { *out = g_kept_pipe; g_kept_pipe = NULL; return SHAFA_SUCCESS; }
    if (g_kept_pipe) { shafa_pipe_destroy(g_kept_pipe); g_kept_pipe = NULL; }
    return shafa_pipe_create(depth, out);
}
static void pipe_put(shafa_pipe *p, int err)
{
    if (!p) return;
    if (err || g_kept_pipe) shafa_pipe_destroy(p);
    else g_kept_pipe = p;
}
void shafa_host_release(void)
{
    if (g_kept_pipe) { shafa_pipe_destroy(g_kept_pipe); g_kept_pipe = NULL; }
}

static int pipe_depth(uint64_t n_blocks)
{
    if (NO_MULTITHREAD) return 1;
    int d = PIPE_SLOTS_PER_DEVICE * shafa_hip_devices();
    if (d > PIPE_SLOTS) d = PIPE_SLOTS;
    if (n_blocks < (uint64_t)d) d = n_blocks ? (int)n_blocks : 1;      /* a short file does not open every GPU (layer 3 uses
                                                                          ceil(slots / 3) of the selected devices) */
    return d;
}

/* ------------------------------------------------------------------ block I/O
 * A 64 MiB block through one fread / fwrite is a single-threaded copy between the page cache and the pinned buffer:
 * about 12 ms per block, 5 GiB/s end to end, with the GPU pipeline (25-30 GiB/s PCIe-inclusive, LABNOTES §1.1) waiting.
 * Blocks are therefore moved with pread / pwrite at explicit offsets.  READS are split over a few helper threads for the
 * duration of the call (the reference, too, touches a block from its own thread: multithread.c:126-194): page-cache /
 * tmpfs reads scale with threads (tools/iobench/tmpfs_rw.c on the GPU box: 6.6 GiB/s with one thread, 15 with eight,
 * 21-23 with sixteen).  WRITES of a new file do not: 5.5 GiB/s with one thread and LESS with more (2.8 with eight: the
 * page allocations of one inode serialise) — so a block is written by one thread.  That write path is what bounds the
 * CLI end to end (LABNOTES §1.1).  --no-multithread: everything inline.  An input that cannot seek (a FIFO, /dev/stdin — the
 * reference's fread loop accepts them in Module C, c.c:392) is read front to back by one thread (io_all). */
/* (helper threads only for slices of 8 MiB: a block of -b m or less is read inline — eleven thread starts per small block
 * were pure overhead) */
enum { IO_THREADS = 12, IO_MIN_SLICE = 8 << 20 };
typedef struct { int fd; uint8_t *buf; size_t n; off_t off; bool write; bool ok; } io_slice;

/* An input that cannot seek (a FIFO, /dev/stdin: the reference's fread loops take them, c.c:392) is read in order: the drivers
 * ask for their blocks front to back, so `off` must be where the last read ended. */
static int seq_fd = -1, seq_back = -1;          /* seq_back: one byte read too far by a header parser (shaf_read_u64), given back */
static off_t seq_pos = 0;
/* The state belongs to ONE input at a time: every driver opens its input through in_open(), which forgets the stream before
 * (a second FIFO in the same process usually gets the same fd number).  Like the module entry points themselves (called from
 * the main thread only, multithread.c:55-60) it is not thread-safe. */
static int in_open(const char *path)
{
    seq_fd = -1;
    seq_back = -1;
    seq_pos = 0;
    return open(path, O_RDONLY);
}
static ssize_t seq_read(int fd, uint8_t *buf, size_t n, off_t off)
{
    if (seq_fd != fd) { seq_fd = fd; seq_pos = 0; seq_back = -1; }
    if (off != seq_pos || !n) return n ? -1 : 0;        /* not the next byte of the stream */
    ssize_t k;
    if (seq_back >= 0) { buf[0] = (uint8_t)seq_back; seq_back = -1; k = 1; }
    else k = read(fd, buf, n);
    if (k > 0) seq_pos += k;
    return k;
}
static bool io_all(int fd, uint8_t *buf, size_t n, off_t off, bool write)
{
    while (n) {
        ssize_t k = write ? pwrite(fd, buf, n, off) : pread(fd, buf, n, off);
        if (k < 0 && errno == ESPIPE && !write) k = seq_read(fd, buf, n, off);
        if (k <= 0) return false;                              /* error, or a file shorter than announced */
        buf += k; n -= (size_t)k; off += k;
    }
    return true;
}
static bool fd_seeks(int fd) { return lseek(fd, 0, SEEK_CUR) != (off_t)-1 || errno != ESPIPE; }
static void *io_slice_main(void *arg)
{
    io_slice *s = arg;
    s->ok = io_all(s->fd, s->buf, s->n, s->off, s->write);
    return NULL;
}
static bool par_io(int fd, const uint8_t *buf, size_t n, off_t off, bool write)
{
    size_t parts = (NO_MULTITHREAD || write || !fd_seeks(fd)) ? 1 : n / IO_MIN_SLICE;
    if (parts > IO_THREADS) parts = IO_THREADS;
    if (parts <= 1) return io_all(fd, (uint8_t *)buf, n, off, write);
    io_slice sl[IO_THREADS];
    pthread_t th[IO_THREADS];
    bool started[IO_THREADS];
    const size_t per = ((n / parts) + 4095) & ~(size_t)4095;
    bool ok = true;
    for (size_t i = 0; i < parts; ++i) {
        const size_t lo = i * per, hi = (i + 1 == parts || (i + 1) * per > n) ? n : (i + 1) * per;
        sl[i] = (io_slice){fd, (uint8_t *)buf + (lo < n ? lo : n), hi > lo ? hi - lo : 0, off + (off_t)lo, write, true};
        started[i] = i > 0 && sl[i].n && pthread_create(&th[i], NULL, io_slice_main, &sl[i]) == 0;
    }
    for (size_t i = 0; i < parts; ++i)                           /* slice 0, and any slice whose thread did not start, inline */
        if (!started[i] && sl[i].n) io_slice_main(&sl[i]);
    for (size_t i = 0; i < parts; ++i) {
        if (started[i]) pthread_join(th[i], NULL);
        ok = ok && sl[i].ok;
    }
    return ok;
}

/* ------------------------------------------------------------------ ordered writer
 * The reference's write callbacks run in block order on the worker threads (multithread.c:75-86).
 * Here one writer thread takes "@size@" + payload jobs in order and appends them at the file offset it keeps, so the
 * main thread can already read the next block while the previous result is written.  With --no-multithread the jobs
 * are written inline.  A job's payload is a slot's pinned result buffer: the slot must not be
 * submitted again before writer_wait() on the job's ticket. */
typedef struct { char hdr[40]; size_t hdr_n; const uint8_t *data; size_t n; } wjob;
enum { WRITER_Q = 64 };        /* jobs in flight: a group of small blocks pushes one job per block */
typedef struct {
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    wjob q[WRITER_Q];
    uint64_t pushed, done;
    int fd;                      /* every job of a writer goes to this file ... */
    off_t off;                   /* ... at this offset (advanced by the thread that runs the jobs) */
    int err;
    bool threaded, stop;
} writer_t;

static int wjob_run(writer_t *w, const wjob *j)
{
    if (w->fd < 0) return SHAFA_FILE_STREAM_FAILED;
    if (j->hdr_n && !io_all(w->fd, (uint8_t *)j->hdr, j->hdr_n, w->off, true)) return SHAFA_FILE_STREAM_FAILED;
    w->off += (off_t)j->hdr_n;
    if (j->n && !par_io(w->fd, j->data, j->n, w->off, true)) return SHAFA_FILE_STREAM_FAILED;
    w->off += (off_t)j->n;
    return SHAFA_SUCCESS;
}

static void *writer_main(void *arg)
{
    writer_t *w = arg;
    pthread_mutex_lock(&w->mu);
    for (;;) {
        while (w->done == w->pushed && !w->stop) pthread_cond_wait(&w->cv, &w->mu);
        if (w->done == w->pushed) break;
        const wjob j = w->q[w->done % WRITER_Q];
        pthread_mutex_unlock(&w->mu);
        const int e = wjob_run(w, &j);
        pthread_mutex_lock(&w->mu);
        if (e && !w->err) w->err = e;
        ++w->done;
        pthread_cond_broadcast(&w->cv);
    }
    pthread_mutex_unlock(&w->mu);
    return NULL;
}

static void writer_start(writer_t *w)
{
    memset(w, 0, sizeof(*w));
    w->fd = -1;
    pthread_mutex_init(&w->mu, NULL);
    pthread_cond_init(&w->cv, NULL);
    w->threaded = !NO_MULTITHREAD && pthread_create(&w->th, NULL, writer_main, w) == 0;
}

/* the file the jobs go to (set before the first push; `off` = bytes already in it) */
static void writer_target(writer_t *w, int fd, off_t off)
{
    pthread_mutex_lock(&w->mu);
    w->fd = fd;
    w->off = off;
    pthread_mutex_unlock(&w->mu);
}

/* (Allocating the file's pages ahead of the data with posix_fallocate on a helper thread was measured and removed: into
 * allocated pages one thread writes 7-7.7 GiB/s instead of 5.5, but the allocation runs on the same inode and slowed the
 * writer to 20 ms per block while it lasted, and an estimate's surplus has to be cut off at the end: -m c on an 8 GiB file
 * 2.0 s with it, 1.5 s without.) */
/* returns the job's ticket (>= 1) */
static uint64_t writer_push(writer_t *w, const char *hdr, const uint8_t *data, size_t n)
{
    wjob j = {.data = data, .n = n};
    j.hdr_n = hdr ? strlen(hdr) : 0;
    if (j.hdr_n) memcpy(j.hdr, hdr, j.hdr_n);
    if (!w->threaded) {
        const int e = wjob_run(w, &j);
        if (e && !w->err) w->err = e;
        return ++w->pushed, ++w->done;
    }
    pthread_mutex_lock(&w->mu);
    while (w->pushed - w->done == WRITER_Q) pthread_cond_wait(&w->cv, &w->mu);
    w->q[w->pushed % WRITER_Q] = j;
    const uint64_t t = ++w->pushed;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
    return t;
}

/* wait until job `ticket` is on disk; returns the first write error so far */
static int writer_wait(writer_t *w, uint64_t ticket)
{
    if (!w->threaded) return w->err;
    pthread_mutex_lock(&w->mu);
    while (w->done < ticket) pthread_cond_wait(&w->cv, &w->mu);
    const int e = w->err;
    pthread_mutex_unlock(&w->mu);
    return e;
}

static int writer_stop(writer_t *w)
{
    if (w->threaded) {
        pthread_mutex_lock(&w->mu);
        w->stop = true;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
        pthread_join(w->th, NULL);
    }
    pthread_mutex_destroy(&w->mu);
    pthread_cond_destroy(&w->cv);
    return w->err;
}

/* create / truncate an output file for the writer; the caller's text header (may be NULL) goes in first */
static int out_open(const char *path, const char *header, off_t *off)
{
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    *off = 0;
    if (fd >= 0 && header) {
        const size_t n = strlen(header);
        if (!io_all(fd, (uint8_t *)header, n, 0, true)) { close(fd); return -2; }
        *off = (off_t)n;
    }
    return fd;
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}

/* ------------------------------------------------------------------ tiny reader for '@' framed text */

typedef struct { char *buf; size_t len, pos; } text_t;

static int slurp(const char *path, text_t *t)
{
    FILE *f = fopen(path, "rb");
    if (!f) return SHAFA_FILE_INACCESSIBLE;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 0) { fclose(f); return SHAFA_FILE_STREAM_FAILED; }
    t->buf = malloc((size_t)sz + 1);
    if (!t->buf) { fclose(f); return SHAFA_LACK_OF_MEMORY; }
    t->len = fread(t->buf, 1, (size_t)sz, f);
    t->buf[t->len] = '\0';
    t->pos = 0;
    fclose(f);
    return SHAFA_SUCCESS;
}

static bool eat(text_t *t, char c)
{
    if (t->pos < t->len && t->buf[t->pos] == c) { ++t->pos; return true; }
    return false;
}

static bool read_u64(text_t *t, uint64_t *v)
{
    size_t p = t->pos;
    uint64_t x = 0;
    while (p < t->len && t->buf[p] >= '0' && t->buf[p] <= '9') x = x * 10 + (uint64_t)(t->buf[p++] - '0');
    if (p == t->pos) return false;
    *v = x;
    t->pos = p;
    return true;
}

/* Bytes of a regular file that have not been consumed yet: a block size announced by a header can never exceed them
 * (checked before a pinned buffer of that size is requested).  The size is taken ONCE with fstat after fopen and the
 * drivers count what they read: an ftell / fseek pair per block would drop the stdio read buffer on the path that feeds
 * the GPU pipeline.  A stream that is not a regular file (a pipe) has no bound: the fread that follows decides. */
typedef struct { uint64_t size, used; bool bounded; } in_budget;
static in_budget budget_of(int fd)
{
    in_budget b = {0, 0, false};
    struct stat sb;
    if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) { b.size = (uint64_t)sb.st_size; b.bounded = true; }
    return b;
}
static bool budget_has(const in_budget *b, uint64_t want) { return !b->bounded || want <= b->size - (b->used < b->size ? b->used : b->size); }

/* "@<R|N>@<n>" (f.c:289,294; t.c:302) */
static bool read_header(text_t *t, char *mode, uint64_t *n)
{
    if (!eat(t, '@') || t->pos >= t->len) return false;
    *mode = t->buf[t->pos++];
    if (!eat(t, '@') || !read_u64(t, n)) return false;
    /* every block needs at least "@<digit>@" in the rest of the text: a count beyond that is a corrupt header
     * (and would overflow the per-block size arrays the drivers allocate from it) */
    return *n <= (uint64_t)(t->len - t->pos) / 3;
}

/* "@<size>@<payload up to the next '@'>" : payload is NUL-terminated in place, *next = char it replaced */
static bool read_block(text_t *t, uint64_t *size, char **payload, size_t max_payload)
{
    if (!eat(t, '@') || !read_u64(t, size) || !eat(t, '@')) return false;
    size_t e = t->pos;
    while (e < t->len && t->buf[e] != '@') ++e;
    if (e == t->pos || e - t->pos > max_payload || e >= t->len) return false;   /* needs a following '@' */
    *payload = t->buf + t->pos;
    t->pos = e;
    return true;
}

/* ------------------------------------------------------------------ Module F (f.c:180-412) */

static void f_summary(uint64_t n_blocks, const uint64_t *sizes, uint64_t size_f, const uint64_t *rle_sizes,
                      double ms, const char *p_rle, const char *p_freq, const char *p_rle_freq)
{
    if (!SHAFA_VERBOSE) return;
    printf("Module: f (calculation of symbol frequencies)\nNumber of blocks: %lu\n", (unsigned long)n_blocks);
    printf("Size of blocks analyzed in the original file: ");
    for (uint64_t i = 0; i < n_blocks; ++i) printf(i + 1 == n_blocks ? "%lu\n" : "%lu/", (unsigned long)sizes[i]);
    if (p_rle) {
        uint64_t total = 0;
        for (uint64_t i = 0; i < n_blocks; ++i) total += rle_sizes[i];
        const float ratio = (float)((long)size_f - (long)total) / (float)size_f * 100.0f;
        printf("RLE Compression: %s (%f%% compression)\n", p_rle, ratio);
        printf("Size of blocks analyzed in the RLE file: ");
        for (uint64_t i = 0; i < n_blocks; ++i)
            printf(i + 1 == n_blocks ? "%lu bytes\n" : "%lu/", (unsigned long)rle_sizes[i]);
    }
    printf("Module runtime (milliseconds): %f\nGenerated files: ", ms);
    if (p_freq && p_rle_freq) printf("%s, %s\n", p_freq, p_rle_freq);
    else if (p_freq) printf("%s\n", p_freq);
    else if (p_rle_freq) printf("%s\n", p_rle_freq);
}

/* ------------------------------------------------------------------ groups of small blocks
 * One launch per block costs the submitting thread ~0.1 ms whatever the block's size: at the reference's default 64 KiB
 * blocks (file.h) that was all of a file's time.  Blocks of less than 2 MiB therefore go through the pipe in groups of
 * consecutive blocks (layer 3: shafa_pipe_submit_group, one launch per kernel for the whole group); the block structure
 * of the files, the order of the writes and the block whose error is reported stay what they are block by block.  The
 * module drivers describe their blocks through two callbacks, both called in block order. */
typedef struct {
    uint64_t in_n;                 /* the block's bytes in the input file ... */
    off_t file_off;                /* ... and where they start */
    shafa_code_table tab;          /* SF ops */
    uint64_t n_symbols;            /* SF decodes */
    size_t out_cap;                /* SF encode */
    int perr;                      /* the block's own error found on the host: surfaces when the block is retired */
    uint8_t *staged;               /* the block's bytes, already read (an input that cannot seek: a group's headers and payloads
                                      alternate in the stream, so prepare() has to take the payload with it); freed once copied */
} gblk;
typedef int (*g_prepare_fn)(void *ctx, uint64_t b, gblk *g);       /* an error stops further submissions; it is returned once
                                                                     the blocks before b are retired */
typedef int (*g_consume_fn)(void *ctx, uint64_t b, const shafa_pipe_result *r, uint64_t *ticket);

static int group_size(uint64_t block_bytes, bool rle_decode)
{
    if (NO_MULTITHREAD || block_bytes >= (2u << 20)) return 1;
    uint64_t g = block_bytes ? (16u << 20) / block_bytes : SHAFA_PIPE_GROUP_MAX;
    if (rle_decode) {                                           /* the pipe sizes a block's result for eight bytes out of one in
                                                                   (and decodes a group again, sized for 85, if one expands more) */
        const uint64_t cap = 8 * block_bytes + 4096;
        const uint64_t fit = (128u << 20) / (cap < ((64u << 20) + 1024) ? cap : ((64u << 20) + 1024));
        if (g > fit) g = fit;
    }
    if (g > SHAFA_PIPE_GROUP_MAX) g = SHAFA_PIPE_GROUP_MAX;
    return g < 2 ? 1 : (int)g;
}

/* blocks [first, end) through `pipe` in groups of at most G */
static int run_groups(shafa_pipe *pipe, int in_fd, uint64_t first, uint64_t end, int G, int op, int flags,
                      g_prepare_fn prepare, g_consume_fn consume, void *ctx, writer_t *wr, uint64_t *ticket)
{
    const uint64_t depth = (uint64_t)shafa_pipe_slots(pipe);
    gblk *gb = calloc(depth * (size_t)G, sizeof(gblk));
    shafa_pipe_block *pb = malloc((size_t)G * sizeof(*pb));
    shafa_pipe_result *res = malloc((size_t)G * sizeof(*res));
    int *brc = malloc((size_t)G * sizeof(int));
    shafa_code_table *no_codes = calloc(1, sizeof(*no_codes));  /* a block with an error of its own is submitted empty */
    uint64_t g_first[PIPE_SLOTS], g_cnt[PIPE_SLOTS];
    int err = (gb && pb && res && brc && no_codes) ? SHAFA_SUCCESS : SHAFA_LACK_OF_MEMORY;
    int deferred = 0;
    uint64_t sub = first, ret = first, sg = 0, rg = 0;
    /* G comes from the FIRST block's size, and the pipe sizes every row of a group for the group's largest block: a file whose
       later blocks are far larger than its first (8 MiB of zeros, then noise, at -b m; a crafted .cod) must not turn into
       165 rows of 64 MiB.  So a group also closes on a byte budget, and a large block goes on its own; the block that
       did not fit is kept (prepare() has consumed it) and opens the next group. */
    gblk held;
    bool have_held = false;
    while (!err && ret < end) {
        if (!deferred && sub < end && sg - rg < depth) {
            const int slot = (int)(sg % depth);
            gblk *g = gb + (size_t)slot * (size_t)G;
            uint64_t cnt = 0, sum_in = 0, max_out = 0;
            while (cnt < (uint64_t)G && sub + cnt < end) {
                if (have_held) { g[cnt] = held; have_held = false; }
                else {
                    free(g[cnt].staged);
                    memset(&g[cnt], 0, sizeof(gblk));
                    const int e = prepare(ctx, sub + cnt, &g[cnt]);
                    if (e) { deferred = e; break; }
                }
                uint64_t out_b = 0;                                      /* the row the pipe will size for this block */
                if (!g[cnt].perr) switch (op) {
                    case SHAFA_OP_RLE_ENCODE: out_b = 2 * g[cnt].in_n + 3; break;
                    case SHAFA_OP_SF_ENCODE: out_b = g[cnt].out_cap; break;
                    case SHAFA_OP_SF_DECODE: out_b = g[cnt].n_symbols; break;
                    case SHAFA_OP_RLE_DECODE: out_b = 8 * g[cnt].in_n + 4096; break;
                    case SHAFA_OP_SF_RLE_DECODE: out_b = 8 * g[cnt].n_symbols + 4096; break;
                    default: break;
                }
                const uint64_t in_b = g[cnt].perr ? 0 : g[cnt].in_n;
                const bool large = in_b >= (2u << 20) || out_b >= (16u << 20);
                const uint64_t widest = out_b > max_out ? out_b : max_out;
                if (cnt && (large || sum_in + in_b > (16u << 20) || (cnt + 1) * widest > (128u << 20))) {
                    held = g[cnt];                                       /* opens the next group */
                    g[cnt].staged = NULL;                                /* (the payload now belongs to `held` alone) */
                    have_held = true;
                    break;
                }
                sum_in += in_b;
                max_out = widest;
                ++cnt;
                if (large) break;                                        /* on its own */
            }
            size_t pos = 0;
            for (uint64_t i = 0; i < cnt; ++i) {
                pb[i].in_off = pos;
                pb[i].in_n = g[i].perr ? 0 : (size_t)g[i].in_n;
                pb[i].table = g[i].perr ? no_codes : &g[i].tab;
                pb[i].n_symbols = g[i].perr ? 0 : (size_t)g[i].n_symbols;
                pb[i].out_cap = g[i].perr ? 16 : g[i].out_cap;
                pos += (pb[i].in_n + 15) & ~(size_t)15;
            }
            uint8_t *buf = cnt ? shafa_pipe_in(pipe, slot, pos ? pos : 16) : NULL;
            if (cnt && !buf) { err = SHAFA_LACK_OF_MEMORY; break; }
            for (uint64_t i = 0; i < cnt; ++i) {
                if (g[i].staged) {
                    if (pb[i].in_n) memcpy(buf + pb[i].in_off, g[i].staged, pb[i].in_n);
                    free(g[i].staged);
                    g[i].staged = NULL;
                    continue;
                }
                if (pb[i].in_n && !io_all(in_fd, buf + pb[i].in_off, pb[i].in_n, g[i].file_off, false)) {
                    deferred = SHAFA_FILE_STREAM_FAILED;                  /* the file is shorter than announced: block i and on */
                    cnt = i;
                    break;
                }
            }
            if (!cnt) continue;
            if ((err = writer_wait(wr, ticket[slot]))) break;              /* the slot's previous results are on disk */
            if ((err = shafa_pipe_submit_group(pipe, slot, op, (int)cnt, pb, flags))) break;
            g_first[slot] = sub;
            g_cnt[slot] = cnt;
            sub += cnt;
            ++sg;
            continue;
        }
        if (sg == rg) { err = deferred ? deferred : SHAFA_FILE_STREAM_FAILED; break; }   /* nothing in flight any more */
        const int slot = (int)(rg % depth);
        const gblk *g = gb + (size_t)slot * (size_t)G;
        const uint64_t cnt = g_cnt[slot];
        if ((err = shafa_pipe_wait_group(pipe, slot, (int)cnt, res, brc))) break;
        for (uint64_t i = 0; i < cnt && !err; ++i) {
            if (g[i].perr) err = g[i].perr;
            else if (brc[i]) err = brc[i];
            else {
                uint64_t tk = 0;
                err = consume(ctx, g_first[slot] + i, &res[i], &tk);
                if (tk) ticket[slot] = tk;
                if (!err) ++ret;
            }
        }
        ++rg;
    }
    for (size_t i = 0; gb && i < depth * (size_t)G; ++i) free(gb[i].staged);      /* (left behind by an error) */
    if (have_held) free(held.staged);
    free(gb); free(pb); free(res); free(brc); free(no_codes);
    return err;
}

static int put_freq_block(FILE *f, uint64_t size, const uint64_t freq[256], bool last)
{
    char text[SHAFA_FREQ_BLOCK_MAX + 1];
    const size_t n = shafa_freq_format(freq, text);
    if (fprintf(f, "@%lu@", (unsigned long)size) < 2) return SHAFA_FILE_STREAM_FAILED;
    if (fwrite(text, 1, n, f) != n) return SHAFA_FILE_STREAM_FAILED;
    if (last && fputs("@0", f) < 0) return SHAFA_FILE_STREAM_FAILED;          /* f.c:112-116 */
    return SHAFA_SUCCESS;
}

typedef struct {
    uint64_t *sizes, *rle_sizes, bs, last, n_blocks;
    bool use_rle, force_freq;
    FILE *f_rle_freq, *f_freq;
    writer_t *wr;
} f_ctx;
static int f_prepare(void *vc, uint64_t b, gblk *g)
{
    f_ctx *c = vc;
    const uint64_t n = (b + 1 == c->n_blocks) ? c->last : c->bs;
    c->sizes[b] = n;
    g->in_n = n;
    g->file_off = (off_t)(b * c->bs);
    return SHAFA_SUCCESS;
}
static int f_consume(void *vc, uint64_t b, const shafa_pipe_result *r, uint64_t *ticket)
{
    f_ctx *c = vc;
    int err = SHAFA_SUCCESS;
    const bool is_last = b + 1 == c->n_blocks;
    if (c->use_rle) {
        c->rle_sizes[b] = r->out_n;
        *ticket = writer_push(c->wr, NULL, r->out, r->out_n);
        if ((err = put_freq_block(c->f_rle_freq, r->out_n, r->freq, is_last))) return err;
    }
    if (!c->use_rle || c->force_freq)                                           /* make_freq of the original, f.c:325 */
        err = put_freq_block(c->f_freq, c->sizes[b], c->use_rle ? r->freq_in : r->freq, is_last);
    return err;
}

_modules_error freq_rle_compress(char **path, bool force_rle, bool force_freq, unsigned long block_size)
{
    const double t0 = now_ms();
    int err = SHAFA_SUCCESS;
    const int in = in_open(*path);
    if (in < 0) return SHAFA_FILE_INACCESSIBLE;
    const in_budget whole = budget_of(in);                                      /* fsize (f.c:212) */
    uint64_t bs = block_size, last = 0;
    const uint64_t size_f = whole.bounded ? whole.size : 0;
    const uint64_t n_blocks = shafa_block_count(size_f, &bs, &last);
    if (size_f < SHAFA_1KiB) { close(in); return SHAFA_FILE_TOO_SMALL; }       /* f.c:220,366 */

    char *p_rle = shafa_add_ext(*path, SHAFA_RLE_EXT);
    char *p_rle_freq = p_rle ? shafa_add_ext(p_rle, SHAFA_FREQ_EXT) : NULL;
    char *p_freq = shafa_add_ext(*path, SHAFA_FREQ_EXT);
    uint64_t *sizes = malloc(n_blocks * sizeof(uint64_t)), *rle_sizes = malloc(n_blocks * sizeof(uint64_t));
    FILE *f_rle_freq = NULL, *f_freq = NULL;
    int f_rle = -1;
    bool use_rle = true;
    shafa_pipe *pipe = NULL;
    shafa_pipe_result *res = malloc(sizeof(*res));
    if (!p_rle || !p_rle_freq || !p_freq || !sizes || !rle_sizes || !res) err = SHAFA_LACK_OF_MEMORY;
    if (!err) err = pipe_get(pipe_depth(n_blocks), &pipe);
    const uint64_t depth = pipe ? (uint64_t)shafa_pipe_slots(pipe) : 1;

    /* submit block `sub`, retire block `ret`; block 0 is retired alone because it decides use_rle (f.c:250-258) */
    uint64_t sub = 0, ret = 0, ticket[PIPE_SLOTS] = {0};
    writer_t wr;
    writer_start(&wr);
    while (!err && ret < n_blocks) {
        const bool can_submit = sub < n_blocks && sub - ret < depth && (sub == 0 || ret > 0);
        if (can_submit) {
            const uint64_t n = (sub + 1 == n_blocks) ? last : bs;
            const int slot = (int)(sub % depth);
            uint8_t *buf = shafa_pipe_in(pipe, slot, n);
            if (!buf) { err = SHAFA_LACK_OF_MEMORY; break; }
            sizes[sub] = n;
            if (!par_io(in, buf, n, (off_t)(sub * bs), false)) { err = SHAFA_FILE_STREAM_FAILED; break; }
            /* block 0: RLE + both histograms (the decision is not known yet); later blocks: what is written */
            const int op = use_rle ? SHAFA_OP_RLE_ENCODE : SHAFA_OP_HIST;
            const int flags = (sub == 0 || force_freq) ? SHAFA_PIPE_INPUT_HIST : 0;
            if ((err = writer_wait(&wr, ticket[slot]))) break;                  /* the slot's previous result is on disk */
            err = shafa_pipe_submit(pipe, slot, op, n, NULL, 0, 0, flags);
            ++sub;
            continue;
        }
        const uint64_t bk = ret, n = sizes[bk];
        err = shafa_pipe_wait(pipe, (int)(bk % depth), res);
        if (err) break;
        if (bk == 0) {                                                          /* f.c:250-295 */
            use_rle = shafa_rle_worthwhile(n, res->out_n, force_rle);
            if (use_rle) {
                off_t o0 = 0;
                f_rle = out_open(p_rle, NULL, &o0);
                f_rle_freq = fopen(p_rle_freq, "wb");
                if (f_rle < 0 || !f_rle_freq) { err = SHAFA_FILE_INACCESSIBLE; break; }
                writer_target(&wr, f_rle, 0);
                if (fprintf(f_rle_freq, "@R@%lu", (unsigned long)n_blocks) < 4) { err = SHAFA_FILE_STREAM_FAILED; break; }
            }
            if (!use_rle || force_freq) {
                f_freq = fopen(p_freq, "wb");
                if (!f_freq) { err = SHAFA_FILE_INACCESSIBLE; break; }
                if (fprintf(f_freq, "@N@%lu", (unsigned long)n_blocks) < 4) { err = SHAFA_FILE_STREAM_FAILED; break; }
            }
        }
        const bool was_rle = bk == 0 || use_rle;                                /* op the block ran with */
        if (use_rle) {
            rle_sizes[bk] = res->out_n;
            ticket[bk % depth] = writer_push(&wr, NULL, res->out, res->out_n);
            err = put_freq_block(f_rle_freq, res->out_n, res->freq, bk + 1 == n_blocks);
            if (err) break;
        }
        if (!use_rle || force_freq)                                             /* make_freq of the original, f.c:325 */
            err = put_freq_block(f_freq, n, was_rle ? res->freq_in : res->freq, bk + 1 == n_blocks);
        ++ret;
        if (!err && bk == 0 && n_blocks > 1 && group_size(bs, false) > 1) {    /* small blocks: the rest in groups, with block 0's decision */
            f_ctx c = {sizes, rle_sizes, bs, last, n_blocks, use_rle, force_freq, f_rle_freq, f_freq, &wr};
            err = run_groups(pipe, in, 1, n_blocks, group_size(bs, false), use_rle ? SHAFA_OP_RLE_ENCODE : SHAFA_OP_HIST,
                             force_freq ? SHAFA_PIPE_INPUT_HIST : 0, f_prepare, f_consume, &c, &wr, ticket);
            ret = n_blocks;
            break;
        }
    }
    {
        const int werr = writer_stop(&wr);
        if (!err) err = werr;
    }
    pipe_put(pipe, err);
    free(res);
    if (f_rle >= 0) close(f_rle);
    if (f_rle_freq) fclose(f_rle_freq);
    if (f_freq) fclose(f_freq);
    close(in);

    if (!err) {
        const bool wrote_freq = !use_rle || force_freq;
        f_summary(n_blocks, sizes, size_f, rle_sizes, now_ms() - t0, use_rle ? p_rle : NULL,
                  wrote_freq ? p_freq : NULL, use_rle ? p_rle_freq : NULL);
        if (use_rle) {                                                           /* f.c:396-399 */
            free(*path);
            *path = p_rle;
            p_rle = NULL;
        }
    }
    free(p_rle); free(p_rle_freq); free(p_freq); free(sizes); free(rle_sizes);
    return (_modules_error)err;
}

/* ------------------------------------------------------------------ Module T (t.c:246-445) */

static void t_summary(uint64_t n_blocks, const uint64_t *sizes, double ms, const char *p_cod)      /* t.c:219-243 */
{
    if (!SHAFA_VERBOSE) return;
    printf("Module:T (Calculation of symbol codes)\nNumber of blocks: %lu\n"
           "Size of blocks analyzed in the symbol file: ", (unsigned long)n_blocks);
    for (uint64_t i = 0; i + 1 < n_blocks; ++i) printf("%lu/", (unsigned long)sizes[i]);
    if (n_blocks) printf("%lu bytes\n", (unsigned long)sizes[n_blocks - 1]);
    printf("Module runtime (milliseconds): %f\nGenerated file %s\n", ms, p_cod);
}

_modules_error get_shafa_codes(const char *path)
{
    const double t0 = now_ms();
    char *p_freq = shafa_add_ext(path, SHAFA_FREQ_EXT);
    if (!p_freq) return SHAFA_LACK_OF_MEMORY;
    text_t t = {0};
    int err = slurp(p_freq, &t);
    free(p_freq);
    if (err) return (_modules_error)err;

    char mode = 0;
    uint64_t n_blocks = 0;
    if (!read_header(&t, &mode, &n_blocks) || (mode != 'R' && mode != 'N')) { free(t.buf); return SHAFA_FILE_UNRECOGNIZABLE; }
    char *p_cod = shafa_add_ext(path, SHAFA_CODES_EXT);
    uint64_t *sizes = malloc((n_blocks ? n_blocks : 1) * sizeof(uint64_t));
    FILE *out = p_cod ? fopen(p_cod, "wb") : NULL;
    if (!p_cod || !sizes) err = SHAFA_LACK_OF_MEMORY;
    else if (!out) err = SHAFA_FILE_INACCESSIBLE;
    else if (fprintf(out, "@%c@%lu", mode, (unsigned long)n_blocks) < 3) err = SHAFA_FILE_STREAM_FAILED;

    char *cod = malloc(SHAFA_COD_BLOCK_MAX + 2);
    if (!cod && !err) err = SHAFA_LACK_OF_MEMORY;
    for (uint64_t b = 0; b < n_blocks && !err; ++b) {
        uint64_t size = 0;
        char *fields = NULL;
        if (!read_block(&t, &size, &fields, SHAFA_FREQ_BLOCK_MAX)) { err = SHAFA_FILE_STREAM_FAILED; break; }
        sizes[b] = size;
        const char keep = t.buf[t.pos];
        t.buf[t.pos] = '\0';
        uint64_t freq[256];
        err = shafa_freq_parse(fields, freq);
        t.buf[t.pos] = keep;
        if (err) break;
        shafa_code_table tab;
        shafa_sf_build_codes(freq, &tab);
        const size_t n = shafa_cod_format(&tab, cod);
        if (fprintf(out, "@%lu@", (unsigned long)size) < 2 || fwrite(cod, 1, n, out) != n) err = SHAFA_FILE_STREAM_FAILED;
    }
    if (!err && out) fputs("@0", out);                                          /* t.c:395-396 */
    if (out) fclose(out);
    if (!err) t_summary(n_blocks, sizes, now_ms() - t0, p_cod);
    free(cod); free(sizes); free(p_cod); free(t.buf);
    return (_modules_error)err;
}

/* ------------------------------------------------------------------ Module C (c.c:306-472) */

typedef struct { text_t *t; in_budget *left; uint64_t *in_sizes, *out_sizes; writer_t *wr; } c_ctx;
static int c_prepare(void *vc, uint64_t b, gblk *g)
{
    c_ctx *c = vc;
    uint64_t size = 0;
    char *codes = NULL;
    if (!read_block(c->t, &size, &codes, SHAFA_COD_BLOCK_MAX)) return SHAFA_FILE_STREAM_FAILED;       /* c.c:369 */
    const char keep = c->t->buf[c->t->pos];
    c->t->buf[c->t->pos] = '\0';
    g->perr = shafa_cod_parse(codes, &g->tab);                                                       /* c.c:115-177 */
    c->t->buf[c->t->pos] = keep;
    if (!budget_has(c->left, size)) return SHAFA_FILE_STREAM_FAILED;                                 /* fread would come up short */
    g->in_n = size;
    g->file_off = (off_t)c->left->used;
    c->left->used += size;
    unsigned lmax = 0;
    for (int q = 0; q < 256; ++q) lmax = g->tab.len[q] > lmax ? g->tab.len[q] : lmax;
    g->out_cap = (size_t)((size * (uint64_t)lmax + 7) / 8) + 16;
    c->in_sizes[b] = size;
    return SHAFA_SUCCESS;
}
static int c_consume(void *vc, uint64_t b, const shafa_pipe_result *r, uint64_t *ticket)
{
    c_ctx *c = vc;
    c->out_sizes[b] = r->out_n;
    char hdr[40];
    snprintf(hdr, sizeof(hdr), "@%lu@", (unsigned long)r->out_n);                                    /* c.c:256-258 */
    *ticket = writer_push(c->wr, hdr, r->out, r->out_n);
    return SHAFA_SUCCESS;
}
/* the size of the next block of a .cod / .freq text without consuming it */
static uint64_t peek_block_size(const text_t *t, size_t max_payload)
{
    text_t q = *t;
    uint64_t size = 0;
    char *skip = NULL;
    return read_block(&q, &size, &skip, max_payload) ? size : 0;
}

static void c_summary(uint64_t n_blocks, const uint64_t *in_sizes, const uint64_t *out_sizes, double ms, const char *p_shaf)   /* c.c:282-303 */
{
    if (!SHAFA_VERBOSE) return;
    printf("Module: C (Symbol codes' codification)\nNumber of blocks: %lu\n", (unsigned long)n_blocks);
    for (uint64_t i = 0; i < n_blocks; ++i)
        printf("Size before/after & compression rate (Block %lu): %lu/%lu -> %d%%\n", (unsigned long)i,
               (unsigned long)in_sizes[i], (unsigned long)out_sizes[i], (int)(((float)out_sizes[i] / in_sizes[i]) * 100));
    printf("Module runtime (milliseconds): %f\nGenerated file %s\n", ms, p_shaf);
}

_modules_error shafa_compress(char **path)
{
    const double t0 = now_ms();
    char *p_cod = shafa_add_ext(*path, SHAFA_CODES_EXT);
    if (!p_cod) return SHAFA_LACK_OF_MEMORY;
    text_t t = {0};
    int err = slurp(p_cod, &t);
    free(p_cod);
    if (err) return (_modules_error)err;
    char mode = 0;
    uint64_t n_blocks = 0;
    if (!read_header(&t, &mode, &n_blocks)) { free(t.buf); return SHAFA_FILE_UNRECOGNIZABLE; }   /* c.c:333,447 */

    const int in = in_open(*path);
    if (in < 0) { free(t.buf); return SHAFA_FILE_INACCESSIBLE; }
    in_budget left = budget_of(in);
    char *p_shaf = shafa_add_ext(*path, SHAFA_SHAFA_EXT);
    char head[32];
    snprintf(head, sizeof(head), "@%lu", (unsigned long)n_blocks);               /* c.c:351 */
    off_t out_off = 0;
    const int out = p_shaf ? out_open(p_shaf, head, &out_off) : -1;
    uint64_t *in_sizes = malloc((n_blocks ? n_blocks : 1) * 2 * sizeof(uint64_t));
    uint64_t *out_sizes = in_sizes ? in_sizes + n_blocks : NULL;
    if (!p_shaf || !in_sizes) err = SHAFA_LACK_OF_MEMORY;
    else if (out == -2) err = SHAFA_FILE_STREAM_FAILED;
    else if (out < 0) err = SHAFA_FILE_INACCESSIBLE;

    shafa_pipe *pipe = NULL;
    shafa_pipe_result *res = malloc(sizeof(*res));
    if (!err && !res) err = SHAFA_LACK_OF_MEMORY;
    trace("C: files open", t0);
    if (!err) err = pipe_get(pipe_depth(n_blocks), &pipe);
    trace("C: pipe created", t0);
    const uint64_t depth = pipe ? (uint64_t)shafa_pipe_slots(pipe) : 1;
    uint64_t sub = 0, ret = 0, ticket[PIPE_SLOTS] = {0};
    writer_t wr;
    writer_start(&wr);
    writer_target(&wr, out, out_off);
    const int G = (!err && n_blocks > 1) ? group_size(peek_block_size(&t, SHAFA_COD_BLOCK_MAX), false) : 1;
    if (!err && G > 1) {                                                                /* small blocks: in groups */
        c_ctx c = {&t, &left, in_sizes, out_sizes, &wr};
        err = run_groups(pipe, in, 0, n_blocks, G, SHAFA_OP_SF_ENCODE, 0, c_prepare, c_consume, &c, &wr, ticket);
        ret = n_blocks;
    }
    while (!err && ret < n_blocks) {
        if (sub < n_blocks && sub - ret < depth) {
            uint64_t size = 0;
            char *codes = NULL;
            if (!read_block(&t, &size, &codes, SHAFA_COD_BLOCK_MAX)) { err = SHAFA_FILE_STREAM_FAILED; break; }   /* c.c:369 */
            const char keep = t.buf[t.pos];
            t.buf[t.pos] = '\0';
            shafa_code_table tab;
            const int perr = shafa_cod_parse(codes, &tab);                       /* c.c:115-177 */
            t.buf[t.pos] = keep;
            const int slot = (int)(sub % depth);
            if (!budget_has(&left, size)) { err = SHAFA_FILE_STREAM_FAILED; break; }                  /* fread would come up short */
            uint8_t *buf = shafa_pipe_in(pipe, slot, size);
            if (!buf) { err = SHAFA_LACK_OF_MEMORY; break; }
            if (!par_io(in, buf, size, (off_t)left.used, false)) { err = SHAFA_FILE_STREAM_FAILED; break; }   /* c.c:392 */
            left.used += size;
            unsigned lmax = 0;
            for (int q = 0; q < 256; ++q) lmax = tab.len[q] > lmax ? tab.len[q] : lmax;
            const size_t need = (size_t)((size * (uint64_t)lmax + 7) / 8) + 16;  /* exact upper bound, not 1.05 n (c.c:58) */
            in_sizes[sub] = size;
            /* a malformed table is this block's error: it must surface after the earlier blocks were written */
            out_sizes[sub] = (uint64_t)perr;
            trace("C:   read", t0);
            if ((err = writer_wait(&wr, ticket[slot]))) break;                  /* the slot's previous result is on disk */
            trace("C:   slot's write done", t0);
            err = shafa_pipe_submit(pipe, slot, perr ? SHAFA_OP_HIST : SHAFA_OP_SF_ENCODE, perr ? 0 : size, &tab, 0, need, 0);
            trace("C:   submitted", t0);
            ++sub;
            continue;
        }
        const int perr = (int)out_sizes[ret];
        err = shafa_pipe_wait(pipe, (int)(ret % depth), res);                    /* binary_coding on the GPU */
        trace("C:   waited", t0);
        if (perr) err = perr;
        if (err) break;
        out_sizes[ret] = res->out_n;
        char hdr[40];
        snprintf(hdr, sizeof(hdr), "@%lu@", (unsigned long)res->out_n);         /* c.c:256-258 */
        ticket[ret % depth] = writer_push(&wr, hdr, res->out, res->out_n);
        trace("C:   write queued", t0);
        ++ret;
    }
    {
        const int werr = writer_stop(&wr);
        if (!err) err = werr;
    }
    trace("C: loop done", t0);
    pipe_put(pipe, err);
    trace("C: pipe destroyed", t0);
    free(res);
    if (out >= 0) close(out);
    close(in);
    if (!err) {
        c_summary(n_blocks, in_sizes, out_sizes, now_ms() - t0, p_shaf);
        free(*path);
        *path = p_shaf;
        p_shaf = NULL;
    }
    free(p_shaf); free(in_sizes); free(t.buf);
    return (_modules_error)err;
}

/* ------------------------------------------------------------------ F -> T -> C on one residency of every block
 * The default `shafa file` (shafa.c:293-298) runs freq_rle_compress, get_shafa_codes and shafa_compress one after the other:
 * every block is read twice (by F, and as .rle by C) and crosses the link twice.  When the three modules run in one process on
 * blocks of 2 MiB or more, this driver does their work with ONE upload per block (layer 3: SHAFA_OP_FTC): F on the device,
 * the block's histogram back (2 KB), Module T here, then Module C from the bytes that are still on the device.  Every file is
 * what the three modules write — .rle, .rle.freq and / or .freq, .cod, .shaf, in that order per block — and the three
 * summaries are printed in module order at the end.
 * Returns SHAFA_FTC_NOT_TAKEN (nothing touched) when the run is not one it takes, and also on ANY error: the caller then runs
 * the three modules, whose files (rewritten from scratch), messages and exit code are the reference's. */
int shafa_ftc_compress(char **path, bool force_rle, bool force_freq, unsigned long block_size)
{
    if (NO_MULTITHREAD || block_size < (2u << 20)) return SHAFA_FTC_NOT_TAKEN;
    const double t0 = now_ms();
    const int in = in_open(*path);
    if (in < 0) return SHAFA_FTC_NOT_TAKEN;
    const in_budget whole = budget_of(in);
    uint64_t bs = block_size, last = 0;
    const uint64_t size_f = whole.bounded ? whole.size : 0;
    const uint64_t n_blocks = shafa_block_count(size_f, &bs, &last);
    if (size_f < SHAFA_1KiB || !n_blocks || bs < (2u << 20)) { close(in); return SHAFA_FTC_NOT_TAKEN; }

    int err = SHAFA_SUCCESS;
    char *p_rle = shafa_add_ext(*path, SHAFA_RLE_EXT);
    char *p_rle_freq = p_rle ? shafa_add_ext(p_rle, SHAFA_FREQ_EXT) : NULL;
    char *p_freq = shafa_add_ext(*path, SHAFA_FREQ_EXT);
    char *p_cod = NULL, *p_shaf = NULL;                 /* named once block 0 has decided: X[.rle].cod / .shaf */
    uint64_t *sizes = malloc(n_blocks * 4 * sizeof(uint64_t));
    uint64_t *rle_sizes = sizes ? sizes + n_blocks : NULL, *enc_in = sizes ? sizes + 2 * n_blocks : NULL,
             *enc_out = sizes ? sizes + 3 * n_blocks : NULL;
    char *cod = malloc(SHAFA_COD_BLOCK_MAX + 2);
    shafa_pipe_result *res = malloc(sizeof(*res));
    FILE *f_rle_freq = NULL, *f_freq = NULL, *f_cod = NULL;
    int f_rle = -1, f_shaf = -1;
    bool use_rle = true;
    shafa_pipe *pipe = NULL;
    if (!p_rle || !p_rle_freq || !p_freq || !sizes || !cod || !res) err = SHAFA_LACK_OF_MEMORY;
    if (!err) err = pipe_get(pipe_depth(n_blocks), &pipe);
    const uint64_t depth = pipe ? (uint64_t)shafa_pipe_slots(pipe) : 1;
    double t_f = 0, t_t = 0;                            /* time spent in the F and T parts (the C summary gets the rest) */

    /* sub1: blocks whose F stage was submitted; ret1: whose F stage was retired (T done, C stage submitted); ret2: done.
       Block 0 goes alone until its RLE size has decided for the file (f.c:250-258). */
    uint64_t sub1 = 0, ret1 = 0, ret2 = 0, tk_rle[PIPE_SLOTS] = {0}, tk_shaf[PIPE_SLOTS] = {0};
    writer_t wr_rle, wr_shaf;
    writer_start(&wr_rle);
    writer_start(&wr_shaf);
    while (!err && ret2 < n_blocks) {
        if (sub1 < n_blocks && sub1 - ret2 < depth && (sub1 == 0 || ret1 > 0)) {
            const uint64_t n = (sub1 + 1 == n_blocks) ? last : bs;
            const int slot = (int)(sub1 % depth);
            uint8_t *buf = shafa_pipe_in(pipe, slot, n);
            if (!buf) { err = SHAFA_LACK_OF_MEMORY; break; }
            sizes[sub1] = n;
            if (!par_io(in, buf, n, (off_t)(sub1 * bs), false)) { err = SHAFA_FILE_STREAM_FAILED; break; }
            int flags;
            if (sub1 == 0) flags = SHAFA_PIPE_FTC_RLE | SHAFA_PIPE_FTC_PLAIN | SHAFA_PIPE_INPUT_HIST;
            else if (use_rle) flags = SHAFA_PIPE_FTC_RLE | (force_freq ? SHAFA_PIPE_INPUT_HIST : 0);
            else flags = SHAFA_PIPE_FTC_PLAIN;
            if ((err = writer_wait(&wr_rle, tk_rle[slot])) || (err = writer_wait(&wr_shaf, tk_shaf[slot]))) break;
            err = shafa_pipe_submit(pipe, slot, SHAFA_OP_FTC, n, NULL, 0, 0, flags);
            ++sub1;
            continue;
        }
        if (ret1 < sub1) {                              /* F's results of block ret1; its codes; its C stage */
            const uint64_t bk = ret1, n = sizes[bk];
            const int slot = (int)(bk % depth);
            const double ta = now_ms();
            if ((err = shafa_pipe_wait(pipe, slot, res))) break;
            if (bk == 0) {                              /* f.c:250-295 */
                use_rle = shafa_rle_worthwhile(n, res->out_n, force_rle);
                char head[48];
                off_t o0 = 0;
                if (use_rle) {
                    f_rle = out_open(p_rle, NULL, &o0);
                    f_rle_freq = fopen(p_rle_freq, "wb");
                    if (f_rle < 0 || !f_rle_freq) { err = SHAFA_FILE_INACCESSIBLE; break; }
                    writer_target(&wr_rle, f_rle, 0);
                    if (fprintf(f_rle_freq, "@R@%lu", (unsigned long)n_blocks) < 4) { err = SHAFA_FILE_STREAM_FAILED; break; }
                }
                if (!use_rle || force_freq) {
                    f_freq = fopen(p_freq, "wb");
                    if (!f_freq) { err = SHAFA_FILE_INACCESSIBLE; break; }
                    if (fprintf(f_freq, "@N@%lu", (unsigned long)n_blocks) < 4) { err = SHAFA_FILE_STREAM_FAILED; break; }
                }
                const char *stem = use_rle ? p_rle : *path;
                p_cod = shafa_add_ext(stem, SHAFA_CODES_EXT);
                p_shaf = shafa_add_ext(stem, SHAFA_SHAFA_EXT);
                f_cod = p_cod ? fopen(p_cod, "wb") : NULL;
                snprintf(head, sizeof(head), "@%lu", (unsigned long)n_blocks);                 /* c.c:351 */
                f_shaf = p_shaf ? out_open(p_shaf, head, &o0) : -1;
                if (!p_cod || !p_shaf) { err = SHAFA_LACK_OF_MEMORY; break; }
                if (!f_cod || f_shaf < 0) { err = SHAFA_FILE_INACCESSIBLE; break; }
                writer_target(&wr_shaf, f_shaf, o0);
                if (fprintf(f_cod, "@%c@%lu", use_rle ? 'R' : 'N', (unsigned long)n_blocks) < 3) { err = SHAFA_FILE_STREAM_FAILED; break; }   /* t.c:302 */
            }
            const bool last_blk = bk + 1 == n_blocks;
            const bool ran_rle = bk == 0 || use_rle;    /* stage one's layout: freq = of the RLE bytes, freq_in = of the input */
            const uint64_t *freq_enc = use_rle ? res->freq : (ran_rle ? res->freq_in : res->freq);
            if (use_rle) {
                rle_sizes[bk] = res->out_n;
                tk_rle[slot] = writer_push(&wr_rle, NULL, res->out, res->out_n);
                if ((err = put_freq_block(f_rle_freq, res->out_n, res->freq, last_blk))) break;
            }
            if (!use_rle || force_freq)                 /* make_freq of the original, f.c:325 */
                if ((err = put_freq_block(f_freq, n, ran_rle ? res->freq_in : res->freq, last_blk))) break;
            const double tb = now_ms();
            t_f += tb - ta;
            /* Module T (t.c:335-361) on the histogram of the bytes that are encoded */
            const uint64_t enc_n = use_rle ? res->out_n : n;
            shafa_code_table tab;
            shafa_sf_build_codes(freq_enc, &tab);
            const size_t cn = shafa_cod_format(&tab, cod);
            if (fprintf(f_cod, "@%lu@", (unsigned long)enc_n) < 2 || fwrite(cod, 1, cn, f_cod) != cn) { err = SHAFA_FILE_STREAM_FAILED; break; }
            if (last_blk && fputs("@0", f_cod) < 0) { err = SHAFA_FILE_STREAM_FAILED; break; }      /* t.c:395-396 */
            t_t += now_ms() - tb;
            /* Module C (c.c:91-237) from the bytes on the device */
            unsigned lmax = 0;
            for (int q = 0; q < 256; ++q) lmax = tab.len[q] > lmax ? tab.len[q] : lmax;
            enc_in[bk] = enc_n;
            err = shafa_pipe_ftc_encode(pipe, slot, use_rle, &tab, (size_t)((enc_n * (uint64_t)lmax + 7) / 8) + 16);
            ++ret1;
            continue;
        }
        {                                               /* Module C's payload of block ret2 (c.c:256-258) */
            const int slot = (int)(ret2 % depth);
            if ((err = shafa_pipe_wait(pipe, slot, res))) break;
            enc_out[ret2] = res->out_n;
            char hdr[40];
            snprintf(hdr, sizeof(hdr), "@%lu@", (unsigned long)res->out_n);
            tk_shaf[slot] = writer_push(&wr_shaf, hdr, res->out, res->out_n);
            ++ret2;
        }
    }
    {
        const int e1 = writer_stop(&wr_rle), e2 = writer_stop(&wr_shaf);
        if (!err) err = e1 ? e1 : e2;
    }
    pipe_put(pipe, err);
    free(res); free(cod);
    if (f_rle >= 0) close(f_rle);
    if (f_shaf >= 0) close(f_shaf);
    if (f_rle_freq && fclose(f_rle_freq) && !err) err = SHAFA_FILE_STREAM_FAILED;
    if (f_freq && fclose(f_freq) && !err) err = SHAFA_FILE_STREAM_FAILED;
    if (f_cod && fclose(f_cod) && !err) err = SHAFA_FILE_STREAM_FAILED;
    close(in);
    if (!err) {
        const double total = now_ms() - t0;
        const bool wrote_freq = !use_rle || force_freq;
        f_summary(n_blocks, sizes, size_f, rle_sizes, t_f, use_rle ? p_rle : NULL, wrote_freq ? p_freq : NULL, use_rle ? p_rle_freq : NULL);
        t_summary(n_blocks, enc_in, t_t, p_cod);
        c_summary(n_blocks, enc_in, enc_out, total - t_f - t_t, p_shaf);
        free(*path);
        *path = p_shaf;
        p_shaf = NULL;
    }
    free(p_rle); free(p_rle_freq); free(p_freq); free(p_cod); free(p_shaf); free(sizes);
    return err ? SHAFA_FTC_NOT_TAKEN : SHAFA_SUCCESS;
}

/* ------------------------------------------------------------------ Module D (d.c:232-388, 628-834) */

static void d_summary(double ms, const uint64_t *before, const uint64_t *after, uint64_t n, const char *path, int algo)
{
    if (!SHAFA_VERBOSE) return;
    printf(algo == 0 ? "Module: D (RLE decoding)\n" : algo == 1 ? "Module: D (SHAFA decoding)\n"
                                                               : "Module: D (SHAFA & RLE decoding)\n");
    for (uint64_t i = 0; i < n; ++i)
        printf("Size before/after generating file (block %lu): %lu/%lu\n", (unsigned long)(i + 1),
               (unsigned long)before[i], (unsigned long)after[i]);
    printf("Module runtime (in milliseconds): %f\nGenerated file %s\n", ms, path);
}

typedef struct { in_budget *left; uint64_t *sizes, *finals; writer_t *wr; } rd_ctx;
static int rd_prepare(void *vc, uint64_t b, gblk *g)
{
    rd_ctx *c = vc;
    if (!budget_has(c->left, c->sizes[b])) return SHAFA_FILE_STREAM_FAILED;
    g->in_n = c->sizes[b];
    g->file_off = (off_t)c->left->used;
    c->left->used += c->sizes[b];
    return SHAFA_SUCCESS;
}
static int rd_consume(void *vc, uint64_t b, const shafa_pipe_result *r, uint64_t *ticket)
{
    rd_ctx *c = vc;
    c->finals[b] = r->out_n;
    *ticket = writer_push(c->wr, NULL, r->out, r->out_n);
    return SHAFA_SUCCESS;
}

_modules_error rle_decompress(char **path)
{
    const double t0 = now_ms();
    const int in = in_open(*path);
    if (in < 0) return SHAFA_FILE_INACCESSIBLE;
    char *p_out = shafa_rm_ext(*path);
    off_t out_off = 0;
    const int out = p_out ? out_open(p_out, NULL, &out_off) : -1;                /* d.c:256 */
    in_budget left = budget_of(in);
    char *p_freq = shafa_add_ext(*path, SHAFA_FREQ_EXT);
    int err = (!p_out || !p_freq) ? SHAFA_LACK_OF_MEMORY : (out < 0 ? SHAFA_FILE_INACCESSIBLE : SHAFA_SUCCESS);
    text_t t = {0};
    if (!err) err = slurp(p_freq, &t);
    char mode = 0;
    uint64_t n_blocks = 0;
    if (!err && !read_header(&t, &mode, &n_blocks)) err = SHAFA_FILE_STREAM_FAILED;   /* d.c:268,296 */
    if (!err && mode != 'R') err = SHAFA_FILE_UNRECOGNIZABLE;                          /* d.c:270,292 */
    uint64_t *sizes = !err ? malloc((n_blocks ? n_blocks : 1) * 2 * sizeof(uint64_t)) : NULL;
    uint64_t *finals = sizes ? sizes + n_blocks : NULL;
    if (!err && !sizes) err = SHAFA_LACK_OF_MEMORY;
    for (uint64_t b = 0; b < n_blocks && !err; ++b) {                                  /* d.c:277-281 */
        char *skip = NULL;
        if (!read_block(&t, &sizes[b], &skip, SHAFA_FREQ_BLOCK_MAX)) err = SHAFA_FILE_STREAM_FAILED;
    }
    shafa_pipe *pipe = NULL;
    shafa_pipe_result *res = malloc(sizeof(*res));
    if (!err && !res) err = SHAFA_LACK_OF_MEMORY;
    if (!err) err = pipe_get(pipe_depth(n_blocks), &pipe);
    const uint64_t depth = pipe ? (uint64_t)shafa_pipe_slots(pipe) : 1;
    uint64_t sub = 0, ret = 0, ticket[PIPE_SLOTS] = {0};
    writer_t wr;
    writer_start(&wr);
    writer_target(&wr, out, out_off);
    const int G = (!err && n_blocks > 1) ? group_size(sizes[0], true) : 1;
    if (!err && G > 1) {                                                                /* small blocks: in groups */
        rd_ctx c = {&left, sizes, finals, &wr};
        err = run_groups(pipe, in, 0, n_blocks, G, SHAFA_OP_RLE_DECODE, 0, rd_prepare, rd_consume, &c, &wr, ticket);
        ret = n_blocks;
    }
    while (!err && ret < n_blocks) {
        if (sub < n_blocks && sub - ret < depth) {
            const int slot = (int)(sub % depth);
            if (!budget_has(&left, sizes[sub])) { err = SHAFA_FILE_STREAM_FAILED; break; }
            uint8_t *buf = shafa_pipe_in(pipe, slot, sizes[sub]);
            if (!buf) { err = SHAFA_LACK_OF_MEMORY; break; }
            if (!par_io(in, buf, sizes[sub], (off_t)left.used, false)) { err = SHAFA_FILE_STREAM_FAILED; break; }   /* d.c:84 */
            left.used += sizes[sub];
            if ((err = writer_wait(&wr, ticket[slot]))) break;
            err = shafa_pipe_submit(pipe, slot, SHAFA_OP_RLE_DECODE, sizes[sub], NULL, 0, 0, 0);   /* rle_block_decompressor */
            ++sub;
            continue;
        }
        err = shafa_pipe_wait(pipe, (int)(ret % depth), res);
        if (err) break;
        finals[ret] = res->out_n;
        ticket[ret % depth] = writer_push(&wr, NULL, res->out, res->out_n);
        ++ret;
    }
    {
        const int werr = writer_stop(&wr);
        if (!err) err = werr;
    }
    pipe_put(pipe, err);
    free(res);
    if (out >= 0) close(out);
    close(in);
    if (!err) {
        d_summary(now_ms() - t0, sizes, finals, n_blocks, p_out, 0);
        free(*path);
        *path = p_out;
        p_out = NULL;
    }
    free(p_out); free(p_freq); free(sizes); free(t.buf);
    return (_modules_error)err;
}

/* "@<n>" then "@<size>@" + size raw bytes per block, from a binary stream (c.c:351,256) */
static bool shaf_read_u64(int fd, off_t *off, char lead, uint64_t *v, bool trailing_at)
{
    char b[32];
    ssize_t got = pread(fd, b, sizeof(b), *off);
    if (got < 0 && errno == ESPIPE) {                   /* a FIFO (d.c:673,697 read it with fscanf): byte by byte, one byte given back */
        got = 0;
        while (got < (ssize_t)sizeof(b) && seq_read(fd, (uint8_t *)b + got, 1, *off + got) == 1) {
            ++got;
            if (got >= 2 && (b[got - 1] < '0' || b[got - 1] > '9')) break;     /* the byte behind the digits */
        }
        ssize_t used = 0;                               /* as the parse below will find: lead, digits, '@' if asked for */
        if (got >= 2 && b[0] == lead) {
            used = 1;
            while (used < got && b[used] >= '0' && b[used] <= '9') ++used;
            if (trailing_at && used < got && b[used] == '@') ++used;
        }
        if (used < got && got >= 1) { seq_back = (unsigned char)b[got - 1]; --seq_pos; }   /* at most the last byte was not used */
    }
    if (got < 2 || b[0] != lead) return false;
    ssize_t i = 1;
    int digits = 0;
    uint64_t x = 0;
    while (i < got && b[i] >= '0' && b[i] <= '9' && digits < 20) { x = x * 10 + (uint64_t)(b[i] - '0'); ++digits; ++i; }
    if (!digits) return false;
    if (trailing_at) {
        if (i >= got || b[i] != '@') return false;
        ++i;
    }
    *off += i;
    *v = x;
    return true;
}

typedef struct { int in; off_t *in_off; text_t *t; in_budget *left; uint64_t *sf_sizes, *sizes, *finals; writer_t *wr; } d_ctx;
static int d_prepare(void *vc, uint64_t b, gblk *g)
{
    d_ctx *c = vc;
    uint64_t sf_n = 0, n_sym = 0;
    if (!shaf_read_u64(c->in, c->in_off, '@', &sf_n, true)) return SHAFA_FILE_STREAM_FAILED;         /* d.c:697 */
    if (!budget_has(c->left, sf_n)) return SHAFA_FILE_STREAM_FAILED;
    g->in_n = sf_n;
    g->file_off = *c->in_off;
    if (!fd_seeks(c->in) && sf_n) {                     /* the payload sits between this header and the next one */
        g->staged = malloc(sf_n);
        if (!g->staged) return SHAFA_LACK_OF_MEMORY;
        if (!io_all(c->in, g->staged, sf_n, *c->in_off, false)) { free(g->staged); g->staged = NULL; return SHAFA_FILE_STREAM_FAILED; }
    }
    *c->in_off += (off_t)sf_n;
    c->left->used += sf_n;
    char *codes = NULL;
    if (!read_block(c->t, &n_sym, &codes, SHAFA_COD_BLOCK_MAX)) return SHAFA_FILE_STREAM_FAILED;     /* d.c:709,716 */
    const char keep = c->t->buf[c->t->pos];
    c->t->buf[c->t->pos] = '\0';
    g->perr = shafa_cod_parse(codes, &g->tab);
    c->t->buf[c->t->pos] = keep;
    g->n_symbols = n_sym;
    c->sf_sizes[b] = sf_n;
    c->sizes[b] = n_sym;
    return SHAFA_SUCCESS;
}
static int d_consume(void *vc, uint64_t b, const shafa_pipe_result *r, uint64_t *ticket)
{
    d_ctx *c = vc;
    c->finals[b] = r->out_n;
    *ticket = writer_push(c->wr, NULL, r->out, r->out_n);
    return SHAFA_SUCCESS;
}

_modules_error shafa_decompress(char **path, bool decompress_rle)
{
    const double t0 = now_ms();
    const int in = in_open(*path);
    if (in < 0) return SHAFA_FILE_INACCESSIBLE;
    int err = SHAFA_SUCCESS;
    in_budget left = budget_of(in);
    off_t in_off = 0;
    char *p_tmp = shafa_rm_ext(*path);                                                /* X[.rle] */
    char *p_out = p_tmp ? (decompress_rle ? shafa_rm_ext(p_tmp) : shafa_add_ext(p_tmp, "")) : NULL;
    char *p_cod = p_tmp ? shafa_add_ext(p_tmp, SHAFA_CODES_EXT) : NULL;
    if (!p_tmp || !p_out || !p_cod) err = SHAFA_LACK_OF_MEMORY;
    off_t out_off = 0;
    const int out = !err ? out_open(p_out, NULL, &out_off) : -1;                      /* d.c:663 (before any check) */
    if (!err && out < 0) err = SHAFA_FILE_INACCESSIBLE;
    text_t t = {0};
    if (!err) err = slurp(p_cod, &t);
    uint64_t n_shaf = 0, n_blocks = 0;
    char mode = 0;
    if (!err && !shaf_read_u64(in, &in_off, '@', &n_shaf, false)) err = SHAFA_FILE_STREAM_FAILED; /* d.c:673 */
    if (!err && !read_header(&t, &mode, &n_blocks)) err = SHAFA_FILE_STREAM_FAILED;               /* d.c:676: .cod's count wins */
    if (!err && !((mode == 'N' && !decompress_rle) || mode == 'R')) err = SHAFA_FILE_UNRECOGNIZABLE;   /* d.c:678 */

    uint64_t *sf_sizes = !err ? malloc((n_blocks ? n_blocks : 1) * 3 * sizeof(uint64_t)) : NULL;
    uint64_t *sizes = sf_sizes ? sf_sizes + n_blocks : NULL, *finals = sf_sizes ? sf_sizes + 2 * n_blocks : NULL;
    if (!err && !sf_sizes) err = SHAFA_LACK_OF_MEMORY;
    shafa_pipe *pipe = NULL;
    shafa_pipe_result *res = malloc(sizeof(*res));
    if (!err && !res) err = SHAFA_LACK_OF_MEMORY;
    if (!err) err = pipe_get(pipe_depth(n_blocks), &pipe);
    const uint64_t depth = pipe ? (uint64_t)shafa_pipe_slots(pipe) : 1;
    uint64_t sub = 0, ret = 0, ticket[PIPE_SLOTS] = {0};
    writer_t wr;
    writer_start(&wr);
    writer_target(&wr, out, out_off);
    const int G = (!err && n_blocks > 1) ? group_size(peek_block_size(&t, SHAFA_COD_BLOCK_MAX), decompress_rle) : 1;
    if (!err && G > 1) {                                                                /* small blocks: in groups */
        d_ctx c = {in, &in_off, &t, &left, sf_sizes, sizes, finals, &wr};
        err = run_groups(pipe, in, 0, n_blocks, G, decompress_rle ? SHAFA_OP_SF_RLE_DECODE : SHAFA_OP_SF_DECODE, 0,
                         d_prepare, d_consume, &c, &wr, ticket);
        ret = n_blocks;
    }
    while (!err && ret < n_blocks) {
        if (sub < n_blocks && sub - ret < depth) {
            uint64_t sf_n = 0, n_sym = 0;
            if (!shaf_read_u64(in, &in_off, '@', &sf_n, true)) { err = SHAFA_FILE_STREAM_FAILED; break; }   /* d.c:697 */
            const int slot = (int)(sub % depth);
            if (!budget_has(&left, sf_n)) { err = SHAFA_FILE_STREAM_FAILED; break; }
            uint8_t *payload = shafa_pipe_in(pipe, slot, sf_n);
            if (!payload) { err = SHAFA_LACK_OF_MEMORY; break; }
            if (!par_io(in, payload, sf_n, in_off, false)) { err = SHAFA_FILE_STREAM_FAILED; break; }   /* d.c:706 */
            trace("D:   read", t0);
            in_off += (off_t)sf_n;
            left.used += sf_n;
            char *codes = NULL;
            if (!read_block(&t, &n_sym, &codes, SHAFA_COD_BLOCK_MAX)) { err = SHAFA_FILE_STREAM_FAILED; break; }   /* d.c:709,716 */
            const char keep = t.buf[t.pos];
            t.buf[t.pos] = '\0';
            shafa_code_table tab;
            const int perr = shafa_cod_parse(codes, &tab);
            t.buf[t.pos] = keep;
            sf_sizes[sub] = sf_n;
            sizes[sub] = n_sym;
            finals[sub] = (uint64_t)perr;                /* this block's own error, reported in block order */
            /* shafa_block_decompressor (+ rle_block_decompressor, d.c:574-586) */
            if ((err = writer_wait(&wr, ticket[slot]))) break;
            trace("D:   slot's write done", t0);
            err = shafa_pipe_submit(pipe, slot, perr ? SHAFA_OP_HIST : (decompress_rle ? SHAFA_OP_SF_RLE_DECODE : SHAFA_OP_SF_DECODE),
                                    perr ? 0 : sf_n, &tab, n_sym, 0, 0);
            trace("D:   submitted", t0);
            ++sub;
            continue;
        }
        const int perr = (int)finals[ret];
        err = shafa_pipe_wait(pipe, (int)(ret % depth), res);
        trace("D:   waited", t0);
        if (perr) err = perr;
        if (err) break;
        finals[ret] = res->out_n;
        ticket[ret % depth] = writer_push(&wr, NULL, res->out, res->out_n);
        ++ret;
    }
    {
        const int werr = writer_stop(&wr);
        if (!err) err = werr;
    }
    trace("D: loop done", t0);
    pipe_put(pipe, err);
    trace("D: pipe destroyed", t0);
    free(res);
    if (out >= 0) close(out);
    close(in);
    if (!err) {
        d_summary(now_ms() - t0, sf_sizes, decompress_rle ? finals : sizes, n_blocks, p_out, decompress_rle ? 2 : 1);
        free(*path);
        *path = p_out;
        p_out = NULL;
    }
    free(p_out); free(p_tmp); free(p_cod); free(sf_sizes); free(t.buf);
    return (_modules_error)err;
}
