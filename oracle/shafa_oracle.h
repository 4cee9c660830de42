/*
 * shafa_oracle.h — CPU restatement of the Shafa hot path (Modules F, T, C, D).
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: a plain-C restatement of the
 * reference's per-block algorithms, written from the behavioural spec in SURVEY.md §9.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (shafa-cd_amd/) never links, imports or calls anything in oracle/.
 *
 * Pinning: the reference ships no tests or golden vectors, so the oracle
 * is pinned against the reference itself: oracle/Makefile compiles the reference's own
 * sources from /root/reference into oracle/_ref/shafa, tests/golden/make_golden.py runs
 * it on deterministic inputs and commits inputs + outputs under tests/golden/, and
 * tests/test_oracle_golden.py checks every function here against those files.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/src/modules/).
 */
#ifndef SHAFA_ORACLE_H
#define SHAFA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* _modules_error values, utils/errors.h:5-16 */
enum {
    ORC_SUCCESS = 0,
    ORC_OUTSIDE_MODULE = 1,
    ORC_LACK_OF_MEMORY = 2,
    ORC_FILE_INACCESSIBLE = 3,
    ORC_FILE_UNRECOGNIZABLE = 4,
    ORC_FILE_STREAM_FAILED = 5,
    ORC_FILE_TOO_SMALL = 6
};

/* One block's code table: len[s] in bits (0 = symbol absent), bits[s] MSB-first, zero padded. */
typedef struct {
    uint8_t len[256];
    uint8_t bits[256][32];
} orc_code_table;

/* RLE decode output limit: 64 MiB + 1 KiB (d.c:129-169). */
#define ORC_RLE_DECODE_MAX ((size_t)67108864 + 1024)

/* f.c:63-79 make_freq */
void orc_hist256(const uint8_t *in, size_t n, uint64_t freq[256]);

/* f.c:29-55 block_compression. out must hold 2n+3 bytes (f.c:244). Returns RLE size. */
size_t orc_rle_encode(const uint8_t *in, size_t n, uint8_t *out);

/* Same result via the per-element closed form of SURVEY.md §9.1 (the form the HIP kernel uses). */
size_t orc_rle_encode_elementwise(const uint8_t *in, size_t n, uint8_t *out);

/* f.c:250-258: RLE accepted for the whole file? (block 0 sizes; float ratio < 0.05 => rejected) */
int orc_rle_accept(size_t n0, size_t rle0, int force_rle);

/* f.c:89-119 write_freq (one block's 256 fields, without the "@size@" prefix and "@0" suffix).
 * dst must hold 256*21 bytes. Returns the number of chars written (no NUL counted). */
size_t orc_freq_write_block(const uint64_t freq[256], char *dst);

/* t.c:27-65 read_block: parse one block's field text (NUL terminated). */
int orc_freq_parse_block(const char *text, uint64_t freq[256]);

/* t.c:74-210: stable descending sort + recursive Shannon-Fano split. */
void orc_sf_build(const uint64_t freq[256], orc_code_table *tab);

/* t.c:353-361: "c0;c1;...;c255" text of one block. dst must hold 33152 bytes. Returns length. */
size_t orc_cod_write_block(const orc_code_table *tab, char *dst);

/* c.c:115-177 / d.c:466-504: parse "c0;...;c255" into a table. */
int orc_cod_parse_block(const char *text, orc_code_table *tab);

/* c.c:52-83 binary_coding (net effect: concatenate code bit-strings MSB-first, zero pad).
 * out must hold orc_sf_encoded_size() bytes. Returns ORC_* ; *out_n = ceil(bits/8). */
int orc_sf_encode(const uint8_t *in, size_t n, const orc_code_table *tab,
                  uint8_t *out, size_t out_cap, size_t *out_n);

/* d.c:466-551 create_tree + shafa_block_decompressor. Decodes exactly n_symbols symbols. */
int orc_sf_decode(const uint8_t *in, size_t in_n, const orc_code_table *tab,
                  uint8_t *out, size_t n_symbols);

/* d.c:116-197 rle_block_decompressor. */
int orc_rle_decode(const uint8_t *in, size_t in_n, uint8_t *out, size_t out_cap, size_t *out_n);

/* Deterministic synthetic byte streams shared by tests, bench and the HIP generator
 * (no reference counterpart).  For byte index i:
 *   w   = splitmix64(seed + (i >> 2));  r16 = (w >> (16 * (i & 3))) & 0xFFFF
 *   map == NULL: byte = r16 >> 8 (uniform);  else byte = map[r16] (e.g. a Zipf inverse CDF,
 *   built by the caller so that CPU and GPU use the very same table bytes). */
void orc_gen_bytes(uint64_t seed, uint64_t first_index, const uint8_t *map65536,
                   uint8_t *out, size_t n);

#ifdef __cplusplus
}
#endif
#endif
