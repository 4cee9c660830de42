/*
 * sfcodes.c — Module T core: Shannon-Fano code construction for one 256-bin histogram
 * (reference t.c:74-210).  Host-side by design: 256 symbols per block is
 * microseconds of work and the result (one shafa_code_table) feeds the HIP encode/decode kernels.
 *
 * Rule set (what makes the output bit-exact with the reference's .cod):
 *   - symbols ordered by frequency, descending, ties by ascending symbol (stable sort, t.c:87);
 *   - only symbols with a non-zero count take part (t.c:202-210); one such symbol => empty code;
 *   - a range [a,b] of ranks is cut after rank d, the first d at which |2*left - total| stops
 *     strictly decreasing (t.c:138-149): '0' is appended to a..d, '1' to d+1..b, then both halves
 *     are cut the same way (t.c:187-193).
 */
#include "shafa_host.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t count; int sym; } ranked;

static int by_count_desc(const void *pa, const void *pb)
{
    const ranked *a = pa, *b = pb;
    if (a->count != b->count) return a->count < b->count ? 1 : -1;
    return a->sym - b->sym;
}

void shafa_sf_build_codes(const uint64_t freq[256], shafa_code_table *out)
{
    ranked r[256];
    int used = 0;
    for (int s = 0; s < 256; ++s) {
        r[s].count = freq[s];
        r[s].sym = s;
        used += freq[s] != 0;
    }
    qsort(r, 256, sizeof(ranked), by_count_desc);

    /* cum[i] = sum of the first i ranked counts: range totals in O(1).  128-bit sums: a hand-made .freq file may carry
     * any 64-bit counts (the reference's `int total` overflows there, t.c:133: undefined; here the same rule on exact sums) */
    unsigned __int128 cum[257];
    cum[0] = 0;
    for (int i = 0; i < 256; ++i) cum[i + 1] = cum[i] + r[i].count;

    memset(out, 0, sizeof(*out));
    if (used < 2) return;

    /* explicit stack of rank ranges; a parent is always cut before its halves */
    struct { int a, b; } stack[512];
    int top = 0;
    stack[top].a = 0; stack[top].b = used - 1; ++top;
    while (top) {
        --top;
        const int a = stack[top].a, b = stack[top].b;
        if (a == b) continue;
        const unsigned __int128 total = cum[b + 1] - cum[a];
        int cut = a;
        unsigned __int128 best = total;
        for (int i = a; i <= b; ++i) {
            const unsigned __int128 left = cum[i + 1] - cum[a], rest = total - left;
            const unsigned __int128 d = left > rest ? left - rest : rest - left;        /* |2 left - total| */
            if (d >= best) break;
            best = d;
            cut = i;
        }
        for (int i = a; i <= b; ++i) {
            const int s = r[i].sym;
            const unsigned n = out->len[s];
            if (i > cut) out->bits[s][n / 8] |= (uint8_t)(0x80 >> (n % 8));
            out->len[s] = (uint8_t)(n + 1);
        }
        stack[top].a = cut + 1; stack[top].b = b; ++top;
        stack[top].a = a; stack[top].b = cut; ++top;
    }
}

/* Module T for the n histograms of a launch (n x 256 counts in, n tables out): the blocks are independent (t.c:286-300 builds
 * them one after the other), so they are handed out by a counter to the caller and up to seven helper threads — a helper
 * is worth starting for sixteen blocks or more; the result does not depend on how many could be started. */
typedef struct { const uint64_t *freq; shafa_code_table *out; int n; atomic_int next; } t_batch;

static void *t_batch_main(void *arg)
{
    t_batch *tb = arg;
    for (;;) {
        const int b = atomic_fetch_add_explicit(&tb->next, 1, memory_order_relaxed);
        if (b >= tb->n) break;
        shafa_sf_build_codes(tb->freq + (size_t)b * 256, &tb->out[b]);
    }
    return NULL;
}

void shafa_sf_build_codes_batch(const uint64_t *freq, int n, shafa_code_table *out)
{
    if (n <= 0) return;
    t_batch tb = {freq, out, n, 0};
    pthread_t th[7];
    int helpers = n / 16, started = 0;
    if (helpers > 7) helpers = 7;
    for (int i = 0; i < helpers; ++i) {
        if (pthread_create(&th[started], NULL, t_batch_main, &tb) != 0) break;
        ++started;
    }
    t_batch_main(&tb);
    for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
}
