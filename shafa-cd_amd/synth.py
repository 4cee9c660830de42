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


# ---------------------------------------------------------------------------------------------------------------------
# Structured stand-in for a real file (BASELINE config[2], Silesia is not available offline): dictionary text with
# order-1 letter statistics, and a binary section of records with zero / 0xFF padding runs.  Unlike the i.i.d. streams
# above these have real-file structure: words repeat, bytes depend on their neighbours, runs cluster.
# ---------------------------------------------------------------------------------------------------------------------
def _dictionary(seed, nwords=4096):
    """pool of pseudo-words (lower-case letters, order-1 Markov inside a word) and their offsets / lengths"""
    rnd = gen_bytes(seed, 64 * nwords).astype(np.int64)
    # letter transition preferences: a fixed pseudo-random 26 x 26 table with a few strong successors per letter
    pref = gen_bytes(seed ^ 0xA5A5, 26 * 26).reshape(26, 26).astype(np.int64)
    order = np.argsort(-pref, axis=1, kind="stable")          # successors of each letter, most likely first
    pick = np.minimum((gen_bytes(seed ^ 0x5A5A, 64 * nwords).astype(np.int64) ** 2) >> 11, 25)   # skewed rank 0..25
    words, k = [], 0
    for w in range(nwords):
        ln = 2 + int(rnd[k]) % (4 if w < 64 else 11)          # frequent words are short
        c = int(rnd[k + 1]) % 26
        letters = [c]
        for j in range(1, ln):
            c = int(order[c, pick[k + 1 + j]])
            letters.append(c)
        k += 64
        words.append(bytes(97 + x for x in letters))
    pool = np.frombuffer(b"".join(words), dtype=np.uint8)
    lens = np.array([len(w) for w in words], dtype=np.int64)
    offs = np.concatenate(([0], np.cumsum(lens)[:-1]))
    return pool, offs, lens


def text_stream(seed, n):
    """n bytes of dictionary text: Zipf-distributed words, separators ' ', ', ', '. ' and newlines, capital after a stop"""
    pool, offs, lens = _dictionary(seed)
    nwords = lens.size
    wmap = np.zeros(65536, dtype=np.int64)                      # r16 -> word id, Zipf(1.1) over the dictionary
    w = np.arange(1, nwords + 1, dtype=np.float64) ** (-1.1)
    edges = np.minimum(np.floor(np.cumsum(w) / np.sum(w) * 65536.0 + 0.5).astype(np.int64), 65536)
    edges[-1] = 65536
    lo = 0
    for k in range(nwords):
        wmap[lo:edges[k]] = k
        lo = max(lo, edges[k])
    seps = [b" ", b" ", b" ", b" ", b" ", b" ", b", ", b". ", b" ", b" ", b" ", b"\n", b" ", b"; ", b" ", b".\n\n"]
    spool = np.frombuffer(b"".join(seps), dtype=np.uint8)
    slens = np.array([len(x) for x in seps], dtype=np.int64)
    soffs = np.concatenate(([0], np.cumsum(slens)[:-1]))
    out = np.empty(n, dtype=np.uint8)
    lens32, offs32, slens32, soffs32 = (x.astype(np.int32) for x in (lens, offs, slens, soffs))
    pos, first, chunk = 0, 0, 1 << 20
    while pos < n:
        with np.errstate(over="ignore"):
            i = np.arange(first, first + chunk, dtype=np.uint64)
            h = splitmix64(np.uint64(seed) + i)
        wid = wmap[(h & np.uint64(0xFFFF)).astype(np.int64)].astype(np.int32)
        sid = ((h >> np.uint64(16)) & np.uint64(15)).astype(np.int32)
        wl = lens32[wid]
        tl = wl + slens32[sid]
        toff = (np.cumsum(tl, dtype=np.int64) - tl).astype(np.int32)
        total = int(toff[-1]) + int(tl[-1])
        tok = np.repeat(np.arange(chunk, dtype=np.int32), tl)   # token of every byte
        rel = np.arange(total, dtype=np.int32) - toff[tok]      # offset inside the token
        in_word = rel < wl[tok]
        src_w = offs32[wid][tok] + rel
        src_s = soffs32[sid][tok] + rel - wl[tok]
        buf = np.where(in_word, pool[np.where(in_word, src_w, 0)], spool[np.where(in_word, 0, src_s)])
        stop = (sid == 7) | (sid == 15)                         # capital letter after a full stop: the word that follows
        cap = toff[1:][stop[:-1]]
        buf[cap] -= 32
        take = min(total, n - pos)
        out[pos:pos + take] = buf[:take]
        pos += take
        first += chunk
    return out


def binary_stream(seed, n):
    """n bytes of a binary section: 32-byte records of little-endian fields (small values: zero high bytes), zero padding
    and 0xFF fill runs of geometric length between groups of records"""
    nrec = n // 32 + 2
    r = gen_bytes(seed, nrec * 32).reshape(nrec, 32).copy()
    r[:, 2:4] = 0                                              # u32 fields with small values
    r[:, 6:8] = 0
    r[:, 9:16] = 0                                             # a u64 that fits a byte
    r[:, 16] &= 0x0F
    r[:, 20:24] = r[:, 16:20]                                  # repeated field
    r[:, 28:32] = 0xFF                                         # sentinel
    body = r.reshape(-1)
    # padding: every 8th record is replaced by a run of zeros or 0xFF
    pad = gen_bytes(seed ^ 0x77, nrec)
    for k, fill in ((0, 0x00), (1, 0xFF)):
        sel = np.nonzero((np.arange(nrec) % 8 == 3 + k) & (pad < 160))[0]
        body.reshape(nrec, 32)[sel, :] = fill
    return body[:n].copy()


def mixed_file_stream(seed, n, binary_first=True):
    """a file whose first third is a binary section and whose rest is text (or the other way round): the RLE verdict of
    block 0 (reference f.c:250-258) is then applied to blocks that would have chosen differently"""
    nb = n // 3
    a = binary_stream(seed, nb)
    b = text_stream(seed + 1, n - nb)
    return np.concatenate((a, b) if binary_first else (b, a))
