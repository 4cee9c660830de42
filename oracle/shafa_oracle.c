/*
 * shafa_oracle.c — CPU restatement of the Shafa hot path.  TEST INFRASTRUCTURE ONLY
 * (see shafa_oracle.h).  Written from the behavioural spec, pinned against
 * the compiled reference through tests/golden/ (tests/test_oracle_golden.py).
 *
 * Reference paths are relative to /root/reference/src/modules/.
 */
#include "shafa_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ histogram */

/* f.c:63-79 make_freq: 256 bins, one increment per byte. */
void orc_hist256(const uint8_t *in, size_t n, uint64_t freq[256])
{
    memset(freq, 0, 256 * sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i) freq[in[i]]++;
}

/* ------------------------------------------------------------------ RLE encode */

/* f.c:29-55 block_compression, stated per maximal run:
 * a run of byte s with length L gives q = L/255 triples {0,s,255}; the remainder m = L%255 gives
 * one triple {0,s,m} when s == 0 (m > 0) or m >= 4, else m literal bytes.  Runs never cross the
 * block end (f.c:38 "j<block_size"). */
size_t orc_rle_encode(const uint8_t *in, size_t n, uint8_t *out)
{
    size_t o = 0, i = 0;
    while (i < n) {
        const uint8_t s = in[i];
        size_t e = i + 1;
        while (e < n && in[e] == s) ++e;
        size_t L = e - i;
        for (; L >= 255; L -= 255) { out[o++] = 0; out[o++] = s; out[o++] = 255; }
        if (L && (s == 0 || L >= 4)) { out[o++] = 0; out[o++] = s; out[o++] = (uint8_t)L; }
        else for (; L; --L) out[o++] = s;
        i = e;
    }
    return o;
}

/* Per-element closed form: with r = (i - head(i)) mod 255 and
 * e = min(255, end(i) - i) the element emits
 *   r == 0:  3 bytes {0,s,e} if s == 0 or e >= 4, else the literal s
 *   r  > 0:  the literal s if s != 0 and r + e < 4, else nothing. */
size_t orc_rle_encode_elementwise(const uint8_t *in, size_t n, uint8_t *out)
{
    if (!n) return 0;
    uint32_t *head = malloc(n * sizeof(uint32_t)), *end = malloc(n * sizeof(uint32_t));
    uint8_t *emit = malloc(n);
    if (!head || !end || !emit) { free(head); free(end); free(emit); return (size_t)-1; }
    for (size_t i = 0; i < n; ++i) head[i] = (i && in[i] == in[i - 1]) ? head[i - 1] : (uint32_t)i;
    for (size_t i = n; i-- > 0;) end[i] = (i + 1 < n && in[i] == in[i + 1]) ? end[i + 1] : (uint32_t)(i + 1);
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) {
        const unsigned r = (unsigned)((i - head[i]) % 255);
        const unsigned e = (end[i] - i) > 255 ? 255u : (unsigned)(end[i] - i);
        const uint8_t s = in[i];
        emit[i] = r == 0 ? ((s == 0 || e >= 4) ? 3 : 1) : ((s != 0 && r + e < 4) ? 1 : 0);
        total += emit[i];
    }
    size_t o = 0;
    for (size_t i = 0; i < n; ++i) {
        if (emit[i] == 3) {
            const unsigned e = (end[i] - i) > 255 ? 255u : (unsigned)(end[i] - i);
            out[o++] = 0; out[o++] = in[i]; out[o++] = (uint8_t)e;
        } else if (emit[i] == 1) out[o++] = in[i];
    }
    free(head); free(end); free(emit);
    return o == total ? o : (size_t)-1;
}

/* f.c:250-258: ratio = (float)(long)(n0 - rle0) / (float)n0 ; rejected iff ratio < 0.05 (the float is
 * promoted to double for the compare) and RLE is not forced. */
int orc_rle_accept(size_t n0, size_t rle0, int force_rle)
{
    const long gain = (long)n0 - (long)rle0;
    const float ratio = (float)gain / (float)n0;
    return !(ratio < 0.05 && !force_rle);
}

/* ------------------------------------------------------------------ .freq text */

static size_t put_u64(char *dst, uint64_t v)
{
    char tmp[24]; size_t k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (size_t j = 0; j < k; ++j) dst[j] = tmp[k - 1 - j];
    return k;
}

/* f.c:89-119 write_freq: a value equal to its predecessor is written as an empty field;
 * 255 separators, none after symbol 255. */
size_t orc_freq_write_block(const uint64_t freq[256], char *dst)
{
    size_t o = 0;
    for (int s = 0; s < 256; ++s) {
        if (s == 0 || freq[s] != freq[s - 1]) o += put_u64(dst + o, freq[s]);
        if (s != 255) dst[o++] = ';';
    }
    dst[o] = '\0';
    return o;
}

/* t.c:27-65 read_block: 256 ';'-separated fields, field 0 numeric, empty field = previous value. */
int orc_freq_parse_block(const char *t, uint64_t freq[256])
{
    for (int s = 0; s < 256; ++s) {
        if (*t >= '0' && *t <= '9') {
            uint64_t v = 0;
            while (*t >= '0' && *t <= '9') v = v * 10 + (uint64_t)(*t++ - '0');
            freq[s] = v;
        } else if (s > 0) freq[s] = freq[s - 1];
        else return ORC_FILE_UNRECOGNIZABLE;
        if (s < 255) { if (*t != ';') return ORC_FILE_UNRECOGNIZABLE; ++t; }
    }
    return *t == '\0' ? ORC_SUCCESS : ORC_FILE_UNRECOGNIZABLE;
}

/* ------------------------------------------------------------------ Shannon-Fano codes (Module T) */

typedef struct { uint64_t f[256]; int sym[256]; uint8_t len[256]; uint8_t bits[256][32]; } sf_work;

static void sf_append(sf_work *w, int rank, int bit)
{
    const unsigned l = w->len[rank];
    if (bit) w->bits[rank][l >> 3] |= (uint8_t)(0x80u >> (l & 7));
    w->len[rank] = (uint8_t)(l + 1);
}

/* t.c:130-152 best_Division + t.c:182-195 sf_codes: grow the left group while |2*g1 - total|
 * strictly decreases (a tie stops); '0' to the left group, '1' to the right; recurse left, right. */
static void sf_split(sf_work *w, int a, int b)
{
    if (a >= b) return;
    int64_t total = 0;
    for (int i = a; i <= b; ++i) total += (int64_t)w->f[i];
    int64_t g = 0, best = total;
    int d = a;
    for (int i = a; i <= b; ++i) {
        g += (int64_t)w->f[i];
        int64_t dif = 2 * g - total; if (dif < 0) dif = -dif;
        if (dif < best) { best = dif; d = i; } else break;
    }
    for (int i = a; i <= d; ++i) sf_append(w, i, 0);
    for (int i = d + 1; i <= b; ++i) sf_append(w, i, 1);
    sf_split(w, a, d);
    sf_split(w, d + 1, b);
}

/* t.c:74-104 insert_sort (descending, stable: ties keep the lower symbol first),
 * t.c:202-210 not_null (only non-zero frequencies take part), t.c:347 sf_codes(0, k-1). */
void orc_sf_build(const uint64_t freq[256], orc_code_table *tab)
{
    sf_work *w = calloc(1, sizeof(sf_work));
    int k = 0;
    for (int s = 0; s < 256; ++s) {           /* stable insertion by descending frequency */
        int j = s;
        while (j > 0 && w->f[j - 1] < freq[s]) { w->f[j] = w->f[j - 1]; w->sym[j] = w->sym[j - 1]; --j; }
        w->f[j] = freq[s]; w->sym[j] = s;
        if (freq[s]) ++k;
    }
    memset(tab, 0, sizeof(*tab));
    if (k >= 2) sf_split(w, 0, k - 1);        /* k == 1: the only symbol keeps the empty code */
    for (int r = 0; r < k; ++r) {
        tab->len[w->sym[r]] = w->len[r];
        memcpy(tab->bits[w->sym[r]], w->bits[r], 32);
    }
    free(w);
}

/* ------------------------------------------------------------------ .cod text */

/* t.c:353-361: codes as '0'/'1' strings joined by ';' (255 separators). */
size_t orc_cod_write_block(const orc_code_table *tab, char *dst)
{
    size_t o = 0;
    for (int s = 0; s < 256; ++s) {
        for (unsigned b = 0; b < tab->len[s]; ++b)
            dst[o++] = (tab->bits[s][b >> 3] >> (7 - (b & 7))) & 1 ? '1' : '0';
        if (s != 255) dst[o++] = ';';
    }
    dst[o] = '\0';
    return o;
}

/* c.c:115-177: exactly 256 fields of '0'/'1'; anything else is _FILE_UNRECOGNIZABLE. */
int orc_cod_parse_block(const char *t, orc_code_table *tab)
{
    memset(tab, 0, sizeof(*tab));
    for (int s = 0; s < 256; ++s) {
        unsigned l = 0;
        while (*t == '0' || *t == '1') {
            if (l >= 255) return ORC_FILE_UNRECOGNIZABLE;
            if (*t == '1') tab->bits[s][l >> 3] |= (uint8_t)(0x80u >> (l & 7));
            ++l; ++t;
        }
        tab->len[s] = (uint8_t)l;
        if (s < 255) { if (*t != ';') return ORC_FILE_UNRECOGNIZABLE; ++t; }
    }
    return *t == '\0' ? ORC_SUCCESS : ORC_FILE_UNRECOGNIZABLE;
}

/* ------------------------------------------------------------------ SF encode (Module C) */

/* c.c:52-83 binary_coding / c.c:91-237 compress_to_buffer.  Net effect (verified
 * against the reference through tests/golden): the block's code bit-strings concatenated MSB-first
 * into a zeroed buffer; size = ceil(bits / 8).
 * Defined here where the reference is undefined: a data symbol whose code is empty
 * while the table holds non-empty codes is _FILE_UNRECOGNIZABLE; an all-empty table (single-symbol
 * block) encodes to 0 bytes as the reference does. */
int orc_sf_encode(const uint8_t *in, size_t n, const orc_code_table *tab,
                  uint8_t *out, size_t out_cap, size_t *out_n)
{
    int any = 0;
    for (int s = 0; s < 256; ++s) any |= tab->len[s] != 0;
    *out_n = 0;
    if (!any) return ORC_SUCCESS;
    uint64_t bits = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!tab->len[in[i]]) return ORC_FILE_UNRECOGNIZABLE;
        bits += tab->len[in[i]];
    }
    const size_t bytes = (size_t)((bits + 7) / 8);
    if (bytes > out_cap) return ORC_LACK_OF_MEMORY;
    memset(out, 0, bytes);
    uint64_t pos = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t s = in[i];
        for (unsigned b = 0; b < tab->len[s]; ++b, ++pos)
            if ((tab->bits[s][b >> 3] >> (7 - (b & 7))) & 1)
                out[pos >> 3] |= (uint8_t)(0x80u >> (pos & 7));
    }
    *out_n = bytes;
    return ORC_SUCCESS;
}

/* ------------------------------------------------------------------ SF decode (Module D) */

/* d.c:422-504 add_tree/create_tree ('0' = left, '1' = right) and d.c:514-551
 * shafa_block_decompressor (bit-serial MSB-first walk, stop after n_symbols symbols, pad ignored).
 * Defined where the reference is undefined: a missing branch, a non-prefix-free table, an empty
 * table with n_symbols > 0 (single-symbol block) or running out of input bits
 * return _FILE_UNRECOGNIZABLE instead of crashing. */
int orc_sf_decode(const uint8_t *in, size_t in_n, const orc_code_table *tab,
                  uint8_t *out, size_t n_symbols)
{
    /* node 0 = root; child < 0: none; leaf[] = symbol or -1 */
    static const int MAXN = 256 * 256;
    int (*child)[2] = malloc(sizeof(int[2]) * MAXN);
    int *leaf = malloc(sizeof(int) * MAXN);
    if (!child || !leaf) { free(child); free(leaf); return ORC_LACK_OF_MEMORY; }
    int nn = 1, rc = ORC_SUCCESS;
    child[0][0] = child[0][1] = -1; leaf[0] = -1;
    for (int s = 0; s < 256 && !rc; ++s) {
        if (!tab->len[s]) continue;
        int cur = 0;
        for (unsigned b = 0; b < tab->len[s]; ++b) {
            const int bit = (tab->bits[s][b >> 3] >> (7 - (b & 7))) & 1;
            if (leaf[cur] >= 0) { rc = ORC_FILE_UNRECOGNIZABLE; break; }
            if (child[cur][bit] < 0) {
                child[nn][0] = child[nn][1] = -1; leaf[nn] = -1;
                child[cur][bit] = nn++;
            }
            cur = child[cur][bit];
        }
        if (!rc) {
            if (leaf[cur] >= 0 || child[cur][0] >= 0 || child[cur][1] >= 0) rc = ORC_FILE_UNRECOGNIZABLE;
            else leaf[cur] = s;
        }
    }
    if (!rc && n_symbols && nn == 1) rc = ORC_FILE_UNRECOGNIZABLE;
    size_t l = 0; uint64_t pos = 0; const uint64_t nbits = (uint64_t)in_n * 8;
    int cur = 0;
    while (!rc && l < n_symbols) {
        if (pos >= nbits) { rc = ORC_FILE_UNRECOGNIZABLE; break; }
        const int bit = (in[pos >> 3] >> (7 - (pos & 7))) & 1; ++pos;
        cur = child[cur][bit];
        if (cur < 0) { rc = ORC_FILE_UNRECOGNIZABLE; break; }
        if (leaf[cur] >= 0) { out[l++] = (uint8_t)leaf[cur]; cur = 0; }
    }
    free(child); free(leaf);
    return rc;
}

/* ------------------------------------------------------------------ RLE decode (Module D) */

/* d.c:116-197 rle_block_decompressor: b != 0 is a literal; b == 0 starts {0, sym, cnt}:
 * cnt copies of sym, and cnt == 0 behaves as one literal sym (d.c:179-184).  Output beyond
 * 64 MiB + 1 KiB is _FILE_UNRECOGNIZABLE (d.c:165-168).  Defined where the reference over-reads:
 * a triple cut by the block end is _FILE_UNRECOGNIZABLE. */
int orc_rle_decode(const uint8_t *in, size_t in_n, uint8_t *out, size_t out_cap, size_t *out_n)
{
    size_t l = 0;
    *out_n = 0;
    for (size_t i = 0; i < in_n; ++i) {
        uint8_t s = in[i];
        size_t cnt = 1;
        if (s == 0) {
            if (i + 2 >= in_n) return ORC_FILE_UNRECOGNIZABLE;
            s = in[i + 1];
            cnt = in[i + 2] ? in[i + 2] : 1;
            i += 2;
        }
        if (l + cnt > ORC_RLE_DECODE_MAX) return ORC_FILE_UNRECOGNIZABLE;
        if (l + cnt > out_cap) return ORC_LACK_OF_MEMORY;
        memset(out + l, s, cnt);
        l += cnt;
    }
    *out_n = l;
    return ORC_SUCCESS;
}

/* ------------------------------------------------------------------ synthetic streams */

static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

void orc_gen_bytes(uint64_t seed, uint64_t first_index, const uint8_t *map, uint8_t *out, size_t n)
{
    for (size_t k = 0; k < n; ++k) {
        const uint64_t i = first_index + k;
        const uint64_t w = splitmix64(seed + (i >> 2));
        const unsigned r16 = (unsigned)(w >> (16 * (i & 3))) & 0xFFFFu;
        out[k] = map ? map[r16] : (uint8_t)(r16 >> 8);
    }
}
