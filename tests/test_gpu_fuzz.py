"""Corrupted inputs (-m gpu): a decoder that guesses chunk entries from the code's self-synchronisation must stay EXACT
on streams that do not behave — flipped bits, bursts of noise, splices of other streams — because its guesses are
verified, not trusted.  The reference decodes whatever bits it is given (d.c:514-551) and so does the oracle: for every
corrupted stream the HIP decoder must return the oracle's return code and, on success, the oracle's bytes.  The same for
rle_decode on arbitrary byte strings (every byte string is a token stream, d.c:116-197)."""
import numpy as np
import pytest

from test_gpu_parity import first_diff, to_shafa_table

pytestmark = pytest.mark.gpu


def _corrupt(rng, enc, kind):
    e = enc.copy()
    n = e.size
    if kind == "bitflips":
        for _ in range(int(rng.integers(1, 6))):
            i = int(rng.integers(0, n))
            e[i] ^= np.uint8(1 << int(rng.integers(0, 8)))
    elif kind == "burst":
        i = int(rng.integers(0, max(1, n - 600)))
        e[i:i + 512] = rng.integers(0, 256, size=e[i:i + 512].size, dtype=np.uint8)
    elif kind == "splice":                      # the second half shifted by a few bytes: every chunk entry after the cut moves
        i = int(rng.integers(n // 4, n // 2))
        k = int(rng.integers(1, 5))
        e[i:n - k] = enc[i + k:n]
    elif kind == "zeros":
        i = int(rng.integers(0, max(1, n - 5000)))
        e[i:i + 4096] = 0
    elif kind == "ones":
        i = int(rng.integers(0, max(1, n - 5000)))
        e[i:i + 4096] = 0xFF
    return e


@pytest.mark.parametrize("speculate", [1, 2, 0])
def test_sf_decode_of_corrupted_streams_equals_the_oracle(oracle, shafa, speculate):
    shafa.lib().shafa_hip_init(0)
    shafa.set_option("sf_decode_speculate", speculate)
    rng = np.random.default_rng(20260 + speculate)
    try:
        for s_zipf, n in ((1.2, 300001), (2.0, 200000), (1.2, 1 << 20)):
            zt = shafa.zipf_table(s_zipf)
            data = oracle.gen_bytes(5150 + n, n, zt)
            otab = oracle.sf_build(oracle.hist256(data))
            t = to_shafa_table(shafa, otab)
            rc, enc = oracle.sf_encode(data, otab)
            assert rc == 0
            for kind in ("bitflips", "burst", "splice", "zeros", "ones", "bitflips", "burst"):
                bad = _corrupt(rng, enc, kind)
                want_rc, want = oracle.sf_decode(bad, otab, n)
                got_rc, got = shafa.sf_decode(bad, t, n, raw_rc=True)
                assert got_rc == want_rc, f"s={s_zipf} n={n} {kind}: rc {got_rc}, oracle {want_rc}"
                if want_rc == 0:
                    assert got.tobytes() == want.tobytes(), f"s={s_zipf} n={n} {kind}: {first_diff(got, want)}"
    finally:
        shafa.set_option("sf_decode_speculate", 1)


@pytest.mark.parametrize("speculate", [1, 2, 0])
def test_sf_decode_of_streams_cut_around_tile_and_unit_edges(oracle, shafa, speculate):
    """A stream that ends a few bytes around an 8 KiB tile edge or a 16 KiB unit edge of the decoder — truncated (fewer symbols
    fit than announced: the oracle's FILE_UNRECOGNIZABLE) and whole (exactly the symbols that fit are asked for).  One to three
    bytes into a unit the unit before is left without sfd_scan's look-ahead word: sfd_ends takes both; the symbol pass must
    not walk on zero fill past a lane's count."""
    shafa.lib().shafa_hip_init(0)
    shafa.set_option("sf_decode_speculate", speculate)
    try:
        for s_zipf, n in ((1.2, 200000), (2.0, 400000)):
            zt = shafa.zipf_table(s_zipf)
            data = oracle.gen_bytes(777 + n, n, zt)
            otab = oracle.sf_build(oracle.hist256(data))
            t = to_shafa_table(shafa, otab)
            rc, enc = oracle.sf_encode(data, otab)
            assert rc == 0 and enc.size > 5 * 16384
            lens = otab.lens().astype(np.int64)
            ends = np.cumsum(lens[data])                                   # bit at which every symbol's code ends
            for edge in (8192, 16384, 3 * 8192, 4 * 16384, 5 * 16384):
                for d in (-3, -1, 0, 1, 2, 3, 5):
                    cut = edge + d
                    part = enc[:cut].copy()
                    fit = int(np.searchsorted(ends, cut * 8, side="right"))   # symbols whose codes end inside the cut stream
                    for nsym in (fit, fit + 1, n):                         # all that fit; one too many; the announced count
                        want_rc, want = oracle.sf_decode(part, otab, nsym)
                        got_rc, got = shafa.sf_decode(part, t, nsym, raw_rc=True)
                        assert got_rc == want_rc, f"s={s_zipf} cut={cut} nsym={nsym} (fit {fit}): rc {got_rc}, oracle {want_rc}"
                        if want_rc == 0:
                            assert got.tobytes() == want.tobytes(), f"s={s_zipf} cut={cut} nsym={nsym}: {first_diff(got, want)}"
    finally:
        shafa.set_option("sf_decode_speculate", 1)


def test_rle_decode_of_arbitrary_bytes_equals_the_oracle(oracle, shafa):
    rng = np.random.default_rng(77)
    for n, pz in ((1000, 0.1), (70000, 0.02), (300000, 0.3), (65536, 0.5), (200001, 0.005)):
        a = rng.integers(1, 256, size=n, dtype=np.uint8)
        a[rng.random(n) < pz] = 0                      # zeros start triples: symbol and count bytes are whatever follows
        want_rc, want = oracle.rle_decode(a)
        got_rc, got = shafa.rle_decode(a, raw_rc=True)
        assert got_rc == want_rc, f"n={n} pz={pz}: rc {got_rc}, oracle {want_rc}"
        if want_rc == 0:
            assert got.tobytes() == want.tobytes(), f"n={n} pz={pz}: {first_diff(got, want)}"


def test_blocks_with_two_faults_report_the_one_a_sequential_coder_meets_first(oracle, shafa):
    """Two faults of different kinds in one block are found by different workgroups in no fixed order; the return code must
    still be the oracle's — the fault a sequential coder stops at — every time.  rle_decode: output past out_cap
    (LACK_OF_MEMORY) in front of a triple cut by the end of the block (FILE_UNRECOGNIZABLE), and out_cap within one token's
    length of SHAFA_RLE_DECODE_MAX (the token that passes out_cap decides; one that also passes the maximum is
    FILE_UNRECOGNIZABLE).  sf_encode: a symbol without a code outranks an output that does not fit (the oracle looks at
    every symbol first)."""
    shafa.lib().shafa_hip_init(0)
    rng = np.random.default_rng(6061)
    # rle_decode: many tiles of triples and literals, the last bytes a cut triple
    for n_tok in (50, 3000, 200000):
        body = np.zeros((n_tok, 3), dtype=np.uint8)
        body[:, 1] = rng.integers(0, 256, size=n_tok)
        body[:, 2] = rng.integers(0, 256, size=n_tok)
        stream = np.concatenate([body.reshape(-1), rng.integers(1, 256, size=777).astype(np.uint8), np.array([0, 65], dtype=np.uint8)])
        rc_full, full = oracle.rle_decode(stream[:-2], cap=int(shafa.RLE_DECODE_MAX))
        total = full.size if rc_full == 0 else None
        for cap in ([total // 2, total - 1, total, total + 5] if total is not None else [1000, 1 << 20]):
            want_rc, _ = oracle.rle_decode(stream, cap=cap)
            for rep in range(4):
                got_rc, _ = shafa.rle_decode(stream, cap=cap, raw_rc=True)
                assert got_rc == want_rc, f"rle_decode n_tok={n_tok} cap={cap} (run {rep}): rc {got_rc}, oracle {want_rc}"
    # out_cap just under the maximum, output past both: runs of 255 (and of 1..254 in front, so that the edges fall inside tokens)
    MAX = int(shafa.RLE_DECODE_MAX)
    n_tok = MAX // 255 + 40
    body = np.zeros((n_tok, 3), dtype=np.uint8)
    body[:, 1] = 9
    body[:, 2] = 255
    for lead in (0, 1, 100, 254):
        stream = np.concatenate([np.full(lead, 7, dtype=np.uint8), body.reshape(-1)])
        for cap in (MAX - 1, MAX - 100, MAX - 254, MAX - 255, MAX - 256, MAX - 400, MAX, MAX + 10):
            want_rc, _ = oracle.rle_decode(stream, cap=cap)
            got_rc, _ = shafa.rle_decode(stream, cap=cap, raw_rc=True)
            assert got_rc == want_rc, f"rle_decode lead={lead} cap=MAX{cap - MAX:+d}: rc {got_rc}, oracle {want_rc}"
    # sf_encode: the table of other data (symbol 200 has no code), an output region that is too small
    zt = shafa.zipf_table(1.2)
    for n in (3000, 500000):
        data = oracle.gen_bytes(4 + n, n, zt) % 128
        otab = oracle.sf_build(oracle.hist256(data))
        t = to_shafa_table(shafa, otab)
        _, enc = oracle.sf_encode(data, otab)
        for where in (5, n // 2, n - 1):
            bad = data.copy()
            bad[where] = 200
            for cap in (enc.size // 3, enc.size - 1, enc.size + 64):
                want_rc, _ = oracle.sf_encode(bad, otab, cap=cap)
                for rep in range(3):
                    got_rc, _ = shafa.sf_encode(bad, t, cap=cap, raw_rc=True)
                    assert got_rc == want_rc, f"sf_encode n={n} bad at {where} cap={cap} (run {rep}): rc {got_rc}, oracle {want_rc}"
