/*
 * formats.c — text formats and small utilities of the C host.
 * .freq / .cod are '@'-framed ASCII; .shaf is '@n' + n * ('@size@' + raw bytes).
 */
#include "shafa_host.h"

#include <stdlib.h>
#include <string.h>

/* utils/errors.c:12-20: same texts (they are part of the CLI's stderr contract) */
const char *shafa_error_msg(int code)
{
    switch (code) {
    case SHAFA_SUCCESS: return "No error\n";
    case SHAFA_LACK_OF_MEMORY: return "Not enough memory for allocation\n";
    case SHAFA_FILE_INACCESSIBLE: return "File can't be accessed. Either lack of permissions or file doesn't exist\n";
    case SHAFA_FILE_UNRECOGNIZABLE: return "File not recognized\n";
    case SHAFA_FILE_STREAM_FAILED: return "Can't communicate properly with file's stream\n";
    case SHAFA_FILE_TOO_SMALL: return "File too small for decompression\n";
    case SHAFA_THREAD_CREATION_FAILED: return "Thread couldn't be created\n";
    case SHAFA_THREAD_TERMINATION_FAILED: return "Thread didn't terminate properly\n";
    case SHAFA_DEVICE_ERROR: return "GPU device error\n";
    default: return "Unknown error";
    }
}

/* utils/file.c:52-117 */
uint64_t shafa_block_count(uint64_t total, uint64_t *block_size, uint64_t *last)
{
    uint64_t bs = *block_size;
    if (bs == 0) bs = 524288;          /* FSIZE_DEFAULT_BLOCK_SIZE */
    if (bs < 512) bs = 512;            /* FSIZE_MIN_BLOCK_SIZE */
    if (bs > SHAFA_64MiB) bs = SHAFA_64MiB;
    *block_size = bs;
    uint64_t nb = total / bs;
    const uint64_t rem = total - nb * bs;
    if (rem) { ++nb; *last = rem; } else *last = bs;
    return nb;
}

bool shafa_has_ext(const char *path, const char *ext)
{
    if (!path) return false;
    const size_t lp = strlen(path), le = strlen(ext);
    return lp >= le && memcmp(path + lp - le, ext, le) == 0;
}

char *shafa_add_ext(const char *path, const char *ext)
{
    const size_t lp = strlen(path), le = strlen(ext);
    char *p = malloc(lp + le + 1);
    if (!p) return NULL;
    memcpy(p, path, lp);
    memcpy(p + lp, ext, le + 1);
    return p;
}

char *shafa_rm_ext(const char *path)
{
    const char *dot = strrchr(path, '.');
    const size_t keep = dot ? (size_t)(dot - path) : strlen(path);
    char *p = malloc(keep + 1);
    if (!p) return NULL;
    memcpy(p, path, keep);
    p[keep] = '\0';
    return p;
}

/* ------------------------------------------------------------------ .freq */

static char *emit_dec(char *p, uint64_t v)
{
    char rev[24];
    int k = 0;
    do rev[k++] = (char)('0' + (int)(v % 10)); while ((v /= 10) != 0);
    while (k) *p++ = rev[--k];
    return p;
}

/* f.c:89-119: walk the histogram run by run; a run of equal counts prints its value once and then
 * only separators. */
size_t shafa_freq_format(const uint64_t freq[256], char *dst)
{
    char *p = dst;
    int s = 0;
    while (s < 256) {
        int e = s + 1;
        while (e < 256 && freq[e] == freq[s]) ++e;
        p = emit_dec(p, freq[s]);
        for (int k = s; k < e; ++k)
            if (k != 255) *p++ = ';';
        s = e;
    }
    *p = '\0';
    return (size_t)(p - dst);
}

/* t.c:27-65 */
int shafa_freq_parse(const char *text, uint64_t freq[256])
{
    const char *p = text;
    uint64_t prev = 0;
    for (int field = 0; field < 256; ++field) {
        if (*p >= '0' && *p <= '9') {
            uint64_t v = 0;
            do v = v * 10u + (uint64_t)(*p - '0'); while (*++p >= '0' && *p <= '9');
            prev = v;
        } else if (field == 0) return SHAFA_FILE_UNRECOGNIZABLE;
        freq[field] = prev;
        if (field != 255 && *p++ != ';') return SHAFA_FILE_UNRECOGNIZABLE;
    }
    return *p ? SHAFA_FILE_UNRECOGNIZABLE : SHAFA_SUCCESS;
}

/* ------------------------------------------------------------------ .cod */

size_t shafa_cod_format(const shafa_code_table *t, char *dst)
{
    char *p = dst;
    for (int s = 0; s < 256; ++s) {
        const unsigned n = t->len[s];
        for (unsigned b = 0; b < n; ++b) *p++ = (char)('0' + ((t->bits[s][b / 8] >> (7 - b % 8)) & 1));
        if (s != 255) *p++ = ';';
    }
    *p = '\0';
    return (size_t)(p - dst);
}

/* c.c:115-177: 256 fields over {'0','1'}; anything else, or a wrong field count, is unrecognizable */
int shafa_cod_parse(const char *text, shafa_code_table *t)
{
    memset(t, 0, sizeof(*t));
    const char *p = text;
    int s = 0;
    unsigned n = 0;
    for (;; ++p) {
        const char c = *p;
        if (c == '0' || c == '1') {
            if (n == 255) return SHAFA_FILE_UNRECOGNIZABLE;
            if (c == '1') t->bits[s][n / 8] |= (uint8_t)(0x80 >> (n % 8));
            ++n;
        } else if (c == ';' || c == '\0') {
            t->len[s] = (uint8_t)n;
            n = 0;
            if (c == '\0') break;
            if (++s == 256) return SHAFA_FILE_UNRECOGNIZABLE;
        } else return SHAFA_FILE_UNRECOGNIZABLE;
    }
    return s == 255 ? SHAFA_SUCCESS : SHAFA_FILE_UNRECOGNIZABLE;
}

/* f.c:250-258 */
bool shafa_rle_worthwhile(uint64_t n0, uint64_t rle0, bool force_rle)
{
    if (force_rle) return true;
    const long saved = (long)n0 - (long)rle0;
    const float ratio = (float)saved / (float)n0;
    return !(ratio < 0.05);
}
