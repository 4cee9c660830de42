/* san_corpus.c — the host's text parsers and Module T under AddressSanitizer + UBSan (CPU only; built by
 * `make -C shafa-cd_amd/host SAN=1` together with the sanitized host sources it links: formats.c, sfcodes.c).
 * Feeds shafa_freq_parse / shafa_cod_parse (t.c:27-65, c.c:115-177 restated in host/formats.c) a corpus of malformed
 * block texts — truncations at every position of a valid text, doubled / missing separators, non-digits, overlong fields,
 * huge numbers, embedded NULs — and every well-formed text round trip; Module T (host/sfcodes.c) gets degenerate
 * histograms (single symbol, all ties, Fibonacci depths, 2^63 counts).  A sanitizer report aborts the run; the program
 * prints what it exercised.  Reference parsers that the corpus mirrors: t.c:27-65, c.c:115-177. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../shafa-cd_amd/host/shafa_host.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void)
{
    uint64_t x = (rng_state += 0x9E3779B97F4A7C15ull);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static long n_freq = 0, n_cod = 0, n_build = 0, n_ok = 0;

static void try_freq(const char *text)
{
    uint64_t f[256];
    char *copy = strdup(text);                 /* heap copy of exact size: an over-read is a report */
    if (shafa_freq_parse(copy, f) == 0) ++n_ok;
    free(copy);
    ++n_freq;
}

static void try_cod(const char *text)
{
    shafa_code_table t;
    char *copy = strdup(text);
    if (shafa_cod_parse(copy, &t) == 0) ++n_ok;
    free(copy);
    ++n_cod;
}

static void mutate(const char *good, void (*fn)(const char *))
{
    const size_t n = strlen(good);
    char *buf = malloc(n + 64);
    /* truncation at every position (at most 4000 of them, spread) */
    const size_t step = n > 4000 ? n / 4000 : 1;
    for (size_t cut = 0; cut <= n; cut += step) {
        memcpy(buf, good, cut);
        buf[cut] = 0;
        fn(buf);
    }
    /* single-character substitutions */
    static const char subs[] = ";;@@-x 92\n\t:01";
    for (int k = 0; k < 3000; ++k) {
        memcpy(buf, good, n + 1);
        if (n) buf[rnd() % n] = subs[rnd() % (sizeof(subs) - 1)];
        fn(buf);
    }
    /* insertions: doubled separators, overlong digit strings */
    for (int k = 0; k < 500 && n; ++k) {
        const size_t at = rnd() % n;
        memcpy(buf, good, at);
        const char *ins = (k & 1) ? ";;;" : "99999999999999999999999999";
        const size_t m = strlen(ins) < 60 ? strlen(ins) : 60;
        memcpy(buf + at, ins, m);
        memcpy(buf + at + m, good + at, n - at + 1);
        fn(buf);
    }
    free(buf);
}

int main(void)
{
    static char text[SHAFA_COD_BLOCK_MAX + 64];
    uint64_t f[256];
    shafa_code_table t;

    /* ---- histograms: degenerate and random; format -> parse round trip; Module T on each -------- */
    for (int c = 0; c < 400; ++c) {
        memset(f, 0, sizeof(f));
        switch (c % 8) {
        case 0: f[rnd() & 255] = 1 + (rnd() & 0xFFFF); break;                                   /* single symbol */
        case 1: for (int s = 0; s < 256; ++s) f[s] = 100; break;                                /* all ties */
        case 2: { uint64_t a = 1, b = 1; for (int i = 0; i < 80; ++i) { f[(i * 7) & 255] = a; const uint64_t n = a + b; a = b; b = n; } } break;
        case 3: f[3] = 1ull << 62; f[200] = 1; f[201] = 1; break;                               /* huge counts */
        case 4: for (int s = 0; s < 256; ++s) f[s] = rnd() & 1 ? 0 : rnd() % 1000; break;
        case 5: for (int s = 0; s < 256; ++s) f[s] = 1ull << (s & 31); break;
        case 6: for (int s = 0; s < 256; ++s) f[s] = (s % 3 == 0) ? 5 : 0; break;
        default: for (int s = 0; s < 256; ++s) f[s] = rnd() % 70000; break;
        }
        const size_t n = shafa_freq_format(f, text);
        if (n > SHAFA_FREQ_BLOCK_MAX) { fprintf(stderr, "freq text too long: %zu\n", n); return 1; }
        uint64_t g[256];
        if (shafa_freq_parse(text, g) != 0 || memcmp(f, g, sizeof(f)) != 0) { fprintf(stderr, "freq round trip %d\n", c); return 1; }
        if (c < 24) mutate(text, try_freq);
        shafa_sf_build_codes(f, &t);
        ++n_build;
        const size_t m = shafa_cod_format(&t, text);
        if (m > SHAFA_COD_BLOCK_MAX) { fprintf(stderr, "cod text too long: %zu\n", m); return 1; }
        shafa_code_table u;
        int nz = 0;
        for (int s = 0; s < 256; ++s) nz += f[s] != 0;
        if (shafa_cod_parse(text, &u) != 0 || memcmp(t.len, u.len, 256) != 0) {
            if (nz > 1 || m != 255) { fprintf(stderr, "cod round trip %d\n", c); return 1; }    /* single symbol: 255 separators, all empty */
        }
        if (c < 24) mutate(text, try_cod);
    }
    /* ---- hand-made extremes ---------------------------------------------------------------------- */
    static const char *odd[] = {"", ";", "@", "0", "1;", ";;;;", "18446744073709551616", "-1;2", "1;2;3", "1 ;2",
                                "0101;x", "010101010101010101010101010101010101010101010101010101010101010101010101010101010101", NULL};
    for (int i = 0; odd[i]; ++i) { try_freq(odd[i]); try_cod(odd[i]); }
    /* 255-bit code: the longest the table holds (c.c:18) and one bit more */
    for (int extra = 0; extra < 2; ++extra) {
        size_t p = 0;
        for (int i = 0; i < 255 + extra; ++i) text[p++] = (i & 1) ? '1' : '0';
        for (int s = 1; s < 256; ++s) text[p++] = ';';
        text[p] = 0;
        try_cod(text);
    }
    uint64_t bs = 65536, last = 0;
    for (uint64_t n = 0; n < 300000; n += 4099) (void)shafa_block_count(n, &bs, &last);
    bs = 1; (void)shafa_block_count(1000, &bs, &last);
    bs = ~0ull; (void)shafa_block_count(~0ull, &bs, &last);
    (void)shafa_rle_worthwhile(0, 0, false);
    (void)shafa_rle_worthwhile(100, 200, false);
    char *e = shafa_add_ext("a", SHAFA_RLE_EXT); char *r = shafa_rm_ext(e); free(e); free(r);
    r = shafa_rm_ext("noext"); free(r);
    printf("san_corpus: %ld .freq texts, %ld .cod texts (%ld accepted), %ld code tables built: no sanitizer report\n",
           n_freq, n_cod, n_ok, n_build);
    return 0;
}
