/*
 * pipe_rate.c — PCIe-inclusive throughput of the layer-3 block pipeline (include/shafa_hip.h):
 * pinned host buffers in, pinned host buffers out, no file I/O.  Measurement tool only.
 *   usage: pipe_rate [blocks=32] [block MiB=64] [slots=3]
 */
#include "shafa_host.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}

int main(int argc, char **argv)
{
    const int nb = argc > 1 ? atoi(argv[1]) : 32, slots = argc > 3 ? atoi(argv[3]) : 3;
    const size_t n = (size_t)(argc > 2 ? atoi(argv[2]) : 64) << 20;
    uint8_t *src = malloc(n), *enc = NULL;
    if (!src) return 2;
    /* skewed bytes: min of three uniform draws, squared down — any skewed stream will do here */
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const unsigned a = x & 255, b = (x >> 8) & 255, c = (x >> 16) & 255;
        unsigned m = a < b ? a : b;
        m = m < c ? m : c;
        src[i] = (uint8_t)((m * m) >> 8);
    }
    uint64_t freq[256];
    int rc = shafa_hip_hist256(src, n, freq);
    if (rc) { fprintf(stderr, "hist256: %d %s\n", rc, shafa_hip_last_error()); return 1; }
    shafa_code_table tab;
    shafa_sf_build_codes(freq, &tab);
    unsigned lmax = 0;
    for (int s = 0; s < 256; ++s) lmax = tab.len[s] > lmax ? tab.len[s] : lmax;
    const size_t cap = (n * lmax + 7) / 8 + 16;

    shafa_pipe *p = NULL;
    if ((rc = shafa_pipe_create(slots, &p))) return 1;
    shafa_pipe_result *res = malloc(sizeof(*res));
    for (int s = 0; s < slots; ++s) memcpy(shafa_pipe_in(p, s, n), src, n);
    size_t enc_n = 0;
    for (int pass = 0; pass < 2; ++pass) {               /* pass 0 warms up (allocations), pass 1 is timed */
        const double t0 = now_ms();
        int sub = 0, ret = 0;
        const int trace = pass && getenv("SHAFA_PIPE_TRACE") != NULL;      /* host time of every call of the timed pass */
        while (ret < nb) {
            const double c0 = now_ms();
            if (sub < nb && sub - ret < slots) {
                if (!shafa_pipe_in(p, sub % slots, n)) return 3;      /* input already there: only the H2D is paid */
                if ((rc = shafa_pipe_submit(p, sub % slots, SHAFA_OP_SF_ENCODE, n, &tab, 0, cap, 0))) return 1;
                if (trace) printf("  %8.3f ms submit %2d (slot %d) took %.3f ms\n", c0 - t0, sub, sub % slots, now_ms() - c0);
                ++sub;
            } else {
                if ((rc = shafa_pipe_wait(p, ret % slots, res))) { fprintf(stderr, "encode wait: %d\n", rc); return 1; }
                if (trace) printf("  %8.3f ms wait   %2d (slot %d) took %.3f ms\n", c0 - t0, ret, ret % slots, now_ms() - c0);
                enc_n = res->out_n;
                if (!enc) { enc = malloc(enc_n); memcpy(enc, res->out, enc_n); }
                ++ret;
            }
        }
        const double ms = now_ms() - t0;
        if (pass) printf("sf_encode  %d x %zu MiB, %d slots: %8.2f ms  %7.2f GiB/s (host pinned -> host pinned, H2D + kernels + D2H)\n",
                         nb, n >> 20, slots, ms, (double)nb * n / 1073741824.0 / (ms / 1e3));
    }
    for (int s = 0; s < slots; ++s) memcpy(shafa_pipe_in(p, s, enc_n), enc, enc_n);
    int bad = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const double t0 = now_ms();
        int sub = 0, ret = 0;
        while (ret < nb) {
            if (sub < nb && sub - ret < slots) {
                if (!shafa_pipe_in(p, sub % slots, enc_n)) return 3;
                if ((rc = shafa_pipe_submit(p, sub % slots, SHAFA_OP_SF_DECODE, enc_n, &tab, n, 0, 0))) return 1;
                ++sub;
            } else {
                if ((rc = shafa_pipe_wait(p, ret % slots, res))) { fprintf(stderr, "decode wait: %d\n", rc); return 1; }
                if (!pass && ret == 0) bad = res->out_n != n || memcmp(res->out, src, n) != 0;
                ++ret;
            }
        }
        const double ms = now_ms() - t0;
        if (pass) printf("sf_decode  %d x %zu MiB, %d slots: %8.2f ms  %7.2f GiB/s (of decoded bytes)\n",
                         nb, n >> 20, slots, ms, (double)nb * n / 1073741824.0 / (ms / 1e3));
    }
    printf("round trip %s; compressed ratio %.4f\n", bad ? "DIFFERS" : "identical", (double)enc_n / n);
    shafa_pipe_destroy(p);
    free(res); free(src); free(enc);
    return bad;
}
