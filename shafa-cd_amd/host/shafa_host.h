/*
 * shafa_host.h — the C host of the MI355X Shafa codec: on-disk formats, Shannon-Fano code
 * construction (Module T) and the module drivers that stand behind the reference's entry points
 * (f.h:16, t.h:11, c.h:11, d.h:14,22).  The per-block compute is dispatched to libshafa_hip.so
 * through include/shafa_hip.h; nothing here computes RLE / histograms / bit packing on the CPU.
 */
#ifndef SHAFA_HOST_H
#define SHAFA_HOST_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/shafa_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum shafa_error _modules_error;     /* same numbers as utils/errors.h:5-16 */

/* ---- utils/file.h:6-12 block sizes, utils/extensions.h:7-10 suffixes --------------------------- */
enum { SHAFA_1KiB = 1024, SHAFA_64KiB = 65536, SHAFA_640KiB = 655360, SHAFA_8MiB = 8388608, SHAFA_64MiB = 67108864 };
#define SHAFA_RLE_EXT ".rle"
#define SHAFA_FREQ_EXT ".freq"
#define SHAFA_CODES_EXT ".cod"
#define SHAFA_SHAFA_EXT ".shaf"
#define SHAFA_COD_BLOCK_MAX 33151              /* c.c:362: longest "c0;...;c255" text */
#define SHAFA_FREQ_BLOCK_MAX (256 * 20 + 255)  /* 256 decimal fields + separators */

/* utils/errors.c:12-20 */
const char *shafa_error_msg(int code);

/* utils/file.c:52-117 fsize semantics: block size clamped to [512, 64 MiB]; n_blocks = ceil(total/bs);
 * *last = size of the last block (== bs when total is a multiple).  Returns n_blocks (0 for empty). */
uint64_t shafa_block_count(uint64_t total_bytes, uint64_t *block_size, uint64_t *last);

/* utils/extensions.c */
bool shafa_has_ext(const char *path, const char *ext);
char *shafa_add_ext(const char *path, const char *ext);   /* malloc'd */
char *shafa_rm_ext(const char *path);                     /* malloc'd; strips from the last '.' */

/* ---- .freq / .cod text -------------------------------------------------------- */
/* f.c:89-119: one block's 256 fields ("repeat = empty field"); returns chars written (dst NUL-terminated,
 * needs SHAFA_FREQ_BLOCK_MAX+1 bytes). */
size_t shafa_freq_format(const uint64_t freq[256], char *dst);
/* t.c:27-65 */
int shafa_freq_parse(const char *text, uint64_t freq[256]);
/* t.c:353-361: needs SHAFA_COD_BLOCK_MAX+1 bytes */
size_t shafa_cod_format(const shafa_code_table *t, char *dst);
/* c.c:115-177 */
int shafa_cod_parse(const char *text, shafa_code_table *t);

/* ---- Module T core (t.c:74-210): Shannon-Fano codes of one histogram ---------------------------- */
void shafa_sf_build_codes(const uint64_t freq[256], shafa_code_table *out);
/* the same for n histograms (n x 256 counts, n tables), blocks spread over up to eight threads */
void shafa_sf_build_codes_batch(const uint64_t *freq, int n, shafa_code_table *out);

/* f.c:250-258: block-0 rule that switches RLE on/off for the whole file */
bool shafa_rle_worthwhile(uint64_t n0, uint64_t rle0, bool force_rle);

/* ---- module entry points: same names, arguments and ownership as the reference ----------------- */
/* f.h:16 */ _modules_error freq_rle_compress(char **path, bool force_rle, bool force_freq, unsigned long block_size);
/* t.h:11 */ _modules_error get_shafa_codes(const char *path);
/* c.h:11 */ _modules_error shafa_compress(char **path);
/* d.h:14 */ _modules_error shafa_decompress(char **path, bool decompress_rle);
/* d.h:22 */ _modules_error rle_decompress(char **path);

/* F, T and C of the default run with one upload per block (blocks of 2 MiB or more, all three modules in one process):
 * the same files and summaries as the three calls above.  SHAFA_FTC_NOT_TAKEN: not such a run, or something failed — the
 * caller runs the three modules (which rewrite every file and report the error as the reference does). */
#define SHAFA_FTC_NOT_TAKEN (-1)
int shafa_ftc_compress(char **path, bool force_rle, bool force_freq, unsigned long block_size);

/* The module drivers keep one block pipeline (streams, pinned and device buffers) between calls of a process; this
 * releases it (no reference counterpart: the reference allocates per block). */
void shafa_host_release(void);

/* multithread.h:19: kept for CLI compatibility; the GPU path has no per-block host threads, the flag
 * only selects one-block-at-a-time dispatch instead of batched dispatch. */
extern bool NO_MULTITHREAD;

/* When true the module drivers print the reference's stdout summaries (without author banners). */
extern bool SHAFA_VERBOSE;

#ifdef __cplusplus
}
#endif
#endif
