"""Deterministic synthetic byte streams: the host-side (numpy) statement of the stream that
shafa_hipd_gen_bytes produces on the device and oracle/shafa_oracle.c's orc_gen_bytes on the CPU — byte i =
map[r16(seed, i)], r16 = 16 bits of splitmix64(seed + i / 4) — and the byte maps / stream shapes bench.py and the
fixtures use.  No reference counterpart (the reference has no generator); numpy only."""
import numpy as np


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)).astype(np.uint64)
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)).astype(np.uint64)
    return x ^ (x >> np.uint64(31))


def gen_bytes(seed, n, table=None, first=0):
    """bytes [first, first + n) of stream `seed`; table = 65536-entry byte map (None: the high byte of r16)."""
    with np.errstate(over="ignore"):
        i = np.arange(first, first + n, dtype=np.uint64)
        w = splitmix64(np.uint64(seed) + (i >> np.uint64(2)))
        r16 = ((w >> (np.uint64(16) * (i & np.uint64(3)))) & np.uint64(0xFFFF)).astype(np.int64)
    if table is None:
        return (r16 >> 8).astype(np.uint8)
    return table[r16]


def _inverse_cdf(p):
    cdf = np.cumsum(p) / np.sum(p)
    edges = np.minimum(np.floor(cdf * 65536.0 + 0.5).astype(np.int64), 65536)
    edges[-1] = 65536
    table = np.zeros(65536, dtype=np.uint8)
    lo = 0
    for k in range(len(p)):
        table[lo:edges[k]] = k
        lo = max(lo, edges[k])
    return table


def zipf_table(s=1.2, nsym=256):
    """2^16-entry inverse CDF of Zipf(s) truncated to nsym ranks: table[r] = smallest k with cdf(k) * 65536 > r."""
    return _inverse_cdf(np.arange(1, nsym + 1, dtype=np.float64) ** (-s))


def zipf_mod256_table(s=1.2):
    """2^16-entry inverse CDF of "Zipf(s) over the positive integers, taken mod 256" (SURVEY.md §8(d) config 4):
    P(byte b) = sum_j (b + 256 j)^-s / zeta(s).  Shannon-Fano output of such a 64 MiB block is 0.812 n."""
    p = np.zeros(256, dtype=np.float64)
    j = np.arange(0, 200000, dtype=np.float64)
    for b in range(256):
        k = (b if b else 256) + 256.0 * j
        # tail of the series beyond the summed terms: integral of x^-s from the last term + 128
        p[b] = np.sum(k ** (-s)) + (k[-1] + 128.0) ** (1.0 - s) / ((s - 1.0) * 256.0)
    return _inverse_cdf(p)


def runs_stream(seed, n, table, p=0.35):
    """Symbols of `table` repeated for geometric run lengths (gives RLE something to do: cfg-0 / pipeline shape)."""
    syms = gen_bytes(seed, n, table)
    u = gen_bytes(seed ^ 0x5DEECE66D, n).astype(np.float64) / 256.0 + 1.0 / 512.0
    runlen = (np.floor(np.log(u) / np.log(1.0 - p)) + 1).astype(np.int64)
    out = np.repeat(syms, runlen)[:n]
    assert out.size == n
    return out.astype(np.uint8)
