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
