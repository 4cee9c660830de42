"""GPU parity tests, part 2: RLE encode/decode (Module F/D) and Shannon-Fano decode (Module D),
through the C-ABI, bit-exact against the oracle and the reference-generated golden files."""
import os

import numpy as np
import pytest

import oracle_lib
from oracle_lib import parse_blocks_text, parse_shaf
from test_gpu_parity import first_diff, long_code_case, rd, streams, to_shafa_table

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 15, 16, 17, 255, 256, 257, 4095, 4096, 4097, 8191, 12288, 65536, 262144 + 5, 1048576 + 77]


def rle_inputs(oracle, shafa):
    import golden.make_golden as mg
    zt = shafa.zipf_table(1.2)
    s = streams(oracle, shafa)
    cases = {"edges": mg.edge_stream()}
    for n in SIZES:
        cases[f"runs_{n}"] = mg.runs_stream(31 + n, n, zt)
        cases[f"uniform_{n}"] = s["uniform"](n)
        cases[f"two_{n}"] = s["two"](n)
    for n in [1, 254, 255, 256, 4096, 4097, 8192 + 3, 255 * 40, 300000, 1 << 20]:
        cases[f"zeros_{n}"] = np.zeros(n, dtype=np.uint8)          # one giant run of the escaped byte
        cases[f"same_{n}"] = np.full(n, 0x5A, dtype=np.uint8)      # one giant run of a plain byte
    # runs that start/stop exactly at tile edges, and a 0/1 alternation (max expansion)
    a = np.full(3 * 4096, 7, dtype=np.uint8)
    a[4096:8192] = 9
    cases["tile_aligned_runs"] = a
    alt = np.zeros(20000, dtype=np.uint8)
    alt[1::2] = 1
    cases["alternating_zero"] = alt
    b = np.full(9000, 3, dtype=np.uint8)
    b[4095] = 4
    b[4096 + 254] = 4
    cases["run_cut_near_tile_edge"] = b
    # text-like data: pairs / tiles whose every byte is a literal take the copy paths of rle3_emit and rle_decode_kernel;
    # sparse tokens in between move the output off the input's alignment and switch between the paths, also right at the
    # edges of a pair (8 KiB) and with the tokens that only exist because of the bytes AROUND a pair
    rng = np.random.default_rng(77)
    for n in [8192, 8192 * 3, 8192 * 5 + 4096, 100000, (1 << 20) + 777]:
        t = (rng.integers(1, 255, size=n) | 1).astype(np.uint8)              # odd bytes: no zeros
        t[1:] = np.where(t[1:] == t[:-1], t[1:] ^ 2, t[1:])                   # and no two equal neighbours
        cases[f"literal_{n}"] = t.copy()
        u = t.copy()
        for pos_ in range(5000, n - 300, 20011):
            kind = (pos_ // 20011) % 4
            if kind == 0:
                u[pos_] = 0                                                   # a single zero: {0,0,1}
            elif kind == 1:
                u[pos_:pos_ + 4] = 0x41                                       # a run of four
            elif kind == 2:
                u[pos_:pos_ + 300] = 0x42                                     # a run over the 255 cap
            else:
                u[pos_:pos_ + 3] = 0x43                                       # a run of three stays literals
        cases[f"literal_sparse_tokens_{n}"] = u
        if n > 8192 * 2:
            e = t.copy()
            e[8190:8194] = 0x51                                               # a run of four across the first pair's end
            e[8192 * 2 - 1] = 0                                               # a zero as the last byte of the second pair
            e[8192 * 2 + 1:8192 * 2 + 4] = e[8192 * 2]                        # a run of four that starts on a pair's first byte
            cases[f"literal_pair_edges_{n}"] = e
    # the one-pass encoder's 32 KiB super-tiles: a run that ENTERS one with L bytes behind it is known from the 64 bytes in
    # front (L < 64), sends its unit to the general code (L >= 60) or needs the state chain (L >= 64: the halo is all run);
    # runs that swallow whole super-tiles (transparent ones: chain aggregates), of a plain byte and of the escaped one
    base = (rng.integers(1, 255, size=32768 * 4 + 1000) | 1).astype(np.uint8)
    base[1:] = np.where(base[1:] == base[:-1], base[1:] ^ 2, base[1:])
    for L in (1, 3, 59, 60, 63, 64, 65, 254, 255, 256, 300, 8192, 33000, 70000):
        for sym in (0x33, 0):
            d = base.copy()
            a0 = 32768 * 3 - L if L < 32768 * 2 else 32768 - (L - 65536) // 2
            d[a0:a0 + L + 10] = sym                     # L bytes in front of a super-tile border, ten behind it (or far more)
            d[a0 + L + 10] = 0x35
            cases[f"supertile_enter_{L}_{sym}"] = d
    for n in (32768, 32768 + 3, 32768 * 2 - 1, 32768 * 2 + 4, 32768 * 3 + 8192 + 2):      # ragged ends around the borders
        cases[f"supertile_ragged_{n}"] = mg.runs_stream(4000 + n, n, zt)
    return cases


# ----------------------------------------------------------------------------- K2 rle_encode
def test_rle_encode_matches_oracle(oracle, shafa):
    bad = []
    for name, data in rle_inputs(oracle, shafa).items():
        want = oracle.rle_encode(data)
        got, freq = shafa.rle_encode(data, want_freq=True)
        if got.tobytes() != want.tobytes():
            bad.append(f"{name}: {first_diff(got, want)}")
        elif not (freq == oracle.hist256(want)).all():
            bad.append(f"{name}: fused histogram differs")
    assert not bad, "\n".join(bad[:12])


def test_rle_encode_one_pass_matches_oracle(oracle, shafa):
    """The chained one-pass form (rle4_kernel, option rle_encode_one_pass) on every input of the default form: run states that
    enter a 32 KiB super-tile (halo / state chain), transparent super-tiles, ragged ends, the fused histogram of the output."""
    shafa.lib().shafa_hip_init(0)
    shafa.set_option("rle_encode_one_pass", 1)
    try:
        bad = []
        for name, data in rle_inputs(oracle, shafa).items():
            want = oracle.rle_encode(data)
            got, freq = shafa.rle_encode(data, want_freq=True)
            if got.tobytes() != want.tobytes():
                bad.append(f"{name}: {first_diff(got, want)}")
            elif not (freq == oracle.hist256(want)).all():
                bad.append(f"{name}: fused histogram differs")
            elif shafa.rle_encode(data).tobytes() != want.tobytes():             # without the histogram
                bad.append(f"{name}: differs when no histogram is asked for")
        assert not bad, "\n".join(bad[:12])
    finally:
        shafa.set_option("rle_encode_one_pass", 0)


@pytest.mark.parametrize("case,fn", [("runs_default", "x"), ("edges_forced_rle", "e"), ("uniform_forced_both", "v"),
                                     ("runs_force_freq", "w"), ("tiny_1024", "a")])
def test_rle_encode_matches_reference_files(shafa, case, fn):
    data, ref = rd(case, fn), rd(case, fn + ".rle")
    _, fblocks = parse_blocks_text(rd(case, fn + ".rle.freq"))
    pos = 0
    for i, (size, ftext) in enumerate(fblocks):
        blk = data[i * 65536:(i + 1) * 65536]
        got, freq = shafa.rle_encode(blk, want_freq=True)
        assert got.size == size and got.tobytes() == ref[pos:pos + size], f"{case} block {i}: {first_diff(got, ref[pos:pos + size])}"
        assert shafa.freq_format(freq) == ftext
        pos += size


# ----------------------------------------------------------------------------- K5 rle_decode
def test_rle_decode_matches_oracle(oracle, shafa):
    bad = []
    for name, data in rle_inputs(oracle, shafa).items():
        rle = oracle.rle_encode(data)
        got = shafa.rle_decode(rle)
        if got.tobytes() != np.asarray(data).tobytes():
            bad.append(f"{name}: {first_diff(got, data)}")
    assert not bad, "\n".join(bad[:12])


def test_rle_decode_hand_made_streams(oracle, shafa):
    cases = {
        "count0_is_literal": bytes([0, 65, 0, 66, 0, 67, 3]),
        "zero_sym": bytes([0, 0, 5, 1, 0, 0, 1]),
        "triples_only": bytes([0, 0, 255] * 5000),               # non-zero bytes never adjacent: chained maps
        "zero_count_zero_sym": bytes([0, 0, 0] * 3000 + [7]),
        "mixed": bytes(([0, 9, 200] + [1, 2, 3] + [0, 0, 1]) * 3000),
        # many tiles whose 32 bytes in front never settle the state (every tile looks back over tile maps: the path that the
        # local entry state, rle_decode.hip, leaves to zero-heavy data), in all three alignments against the 8 KiB tiles
        "triples_many_tiles": bytes([0, 0, 3] * 200000),
        "all_zero_many_tiles": bytes(300001 // 3 * 3),
        "zeros_shifted_1": bytes([5]) + bytes([0, 0, 2] * 100000) + bytes([7, 8, 9] * 5000),
        "zeros_shifted_2": bytes([5, 6]) + bytes([0, 0, 2] * 100000) + bytes([0, 7, 200] * 300),
        "text_zeros_text": bytes([65 + (i * 7) % 26 for i in range(50000)]) + bytes([0, 0, 1] * 30000) + bytes([97 + (i * 5) % 26 for i in range(50000)]),
    }
    for name, raw in cases.items():
        rc, want = oracle.rle_decode(raw)
        assert rc == 0
        got = shafa.rle_decode(raw)
        assert got.tobytes() == want.tobytes(), f"{name}: {first_diff(got, want)}"
    # a triple cut by the block end is refused (the reference over-reads there)
    for raw in (bytes([65, 0, 66]), bytes([65, 0]), bytes([1] * 5000 + [0, 3])):
        rc, _ = shafa.rle_decode(raw, raw_rc=True)
        assert rc == shafa.FILE_UNRECOGNIZABLE
    # output limit 64 MiB + 1 KiB (d.c:165-168) and caller capacity
    n_tr = (shafa.RLE_DECODE_MAX // 255) + 1
    big = np.tile(np.array([0, 1, 255], dtype=np.uint8), n_tr)
    rc, _ = shafa.rle_decode(big, cap=shafa.RLE_DECODE_MAX, raw_rc=True)
    assert rc == shafa.FILE_UNRECOGNIZABLE
    rc, _ = shafa.rle_decode(bytes([0, 1, 255] * 100), cap=1000, raw_rc=True)
    assert rc == shafa.LACK_OF_MEMORY
    rc, out = shafa.rle_decode(b"", raw_rc=True)
    assert rc == 0 and out.size == 0


# ----------------------------------------------------------------------------- K4 sf_decode
def decode_roundtrip(oracle, shafa, data, otab):
    rc, enc = oracle.sf_encode(data, otab)
    assert rc == 0
    got = shafa.sf_decode(enc, to_shafa_table(shafa, otab), len(data))
    return got, enc


@pytest.mark.parametrize("kind", ["uniform", "zipf", "two"])
def test_sf_decode_matches_oracle_sizes(oracle, shafa, kind):
    gen = streams(oracle, shafa)[kind]
    bad = []
    for n in SIZES:
        if n < 2:
            continue
        data = gen(n)
        otab = oracle.sf_build(oracle.hist256(data))
        if otab.lens().max() == 0:
            continue
        got, enc = decode_roundtrip(oracle, shafa, data, otab)
        if got.tobytes() != data.tobytes():
            bad.append(f"{kind} n={n} enc={enc.size}: {first_diff(got, data)}")
    assert not bad, "\n".join(bad[:12])


def test_sf_decode_long_codes(oracle, shafa):
    import golden.make_golden as mg
    data = mg.textlike_stream(5, 200000)                      # lengths up to 16: LUT + trie
    otab = oracle.sf_build(oracle.hist256(data))
    got, _ = decode_roundtrip(oracle, shafa, data, otab)
    assert got.tobytes() == data.tobytes(), first_diff(got, data)
    for nsyms, n in ((30, 150000), (60, 50000)):              # up to 29 / 52 bits: trie path, R = 32 / 64
        otab, data = long_code_case(oracle, n, nsyms, 0.5, 21)
        got, _ = decode_roundtrip(oracle, shafa, data, otab)
        assert got.tobytes() == data.tobytes(), f"nsyms={nsyms} lmax={otab.lens().max()}: {first_diff(got, data)}"


def test_sf_decode_ignores_padding_and_trailing_bytes(oracle, shafa):
    data = streams(oracle, shafa)["zipf"](50001)
    otab = oracle.sf_build(oracle.hist256(data))
    rc, enc = oracle.sf_encode(data, otab)
    tab = to_shafa_table(shafa, otab)
    padded = np.concatenate([enc, np.zeros(100, dtype=np.uint8)])
    assert shafa.sf_decode(padded, tab, data.size).tobytes() == data.tobytes()
    assert shafa.sf_decode(enc, tab, 1234).tobytes() == data[:1234].tobytes()    # stop on symbol count


def test_sf_decode_error_semantics(oracle, shafa):
    f = np.zeros(256, dtype=np.uint64)
    f[7] = 100
    tab = to_shafa_table(shafa, oracle.sf_build(f))                 # single symbol: all codes empty
    rc, _ = shafa.sf_decode(b"", tab, 100, raw_rc=True)
    assert rc == shafa.FILE_UNRECOGNIZABLE
    rc, out = shafa.sf_decode(b"", tab, 0, raw_rc=True)
    assert rc == 0
    bad = shafa.CodeTable.from_strings(["0", "01"] + [""] * 254)     # not prefix-free
    rc, _ = shafa.sf_decode(bytes(10), bad, 5, raw_rc=True)
    assert rc == shafa.FILE_UNRECOGNIZABLE
    inc = shafa.CodeTable.from_strings(["00", "01", "10"] + [""] * 253)   # incomplete tree: '11' missing
    rc, out = shafa.sf_decode(bytes([0b00011000]), inc, 4, raw_rc=True)
    assert rc == 0 and out.tolist() == [0, 1, 2, 0]
    rc, _ = shafa.sf_decode(bytes([0b00110000]), inc, 3, raw_rc=True)
    assert rc == shafa.FILE_UNRECOGNIZABLE
    data = streams(oracle, shafa)["uniform"](5000)
    otab = oracle.sf_build(oracle.hist256(data))
    rc, enc = oracle.sf_encode(data, otab)
    rc, _ = shafa.sf_decode(enc[:2000], to_shafa_table(shafa, otab), 5000, raw_rc=True)   # stream too short
    assert rc == shafa.FILE_UNRECOGNIZABLE


@pytest.mark.parametrize("case,stem", [("runs_default", "x.rle"), ("edges_forced_rle", "e.rle"),
                                       ("uniform_no_rle", "u"), ("textlike_m", "t"), ("tiny_1024", "a.rle")])
def test_sf_decode_matches_reference_files(shafa, case, stem):
    """HIP decode of the reference's .shaf payloads with its .cod == the file the reference encoded."""
    data = rd(case, stem)
    _, cblocks = parse_blocks_text(rd(case, stem + ".cod"))
    payloads = parse_shaf(rd(case, stem + ".shaf"))
    pos = 0
    for i, ((size, ctext), payload) in enumerate(zip(cblocks, payloads)):
        rc, tab = shafa.cod_parse(ctext)
        got = shafa.sf_decode(payload, tab, size)
        assert got.tobytes() == data[pos:pos + size], f"{case} block {i}: {first_diff(got, data[pos:pos + size])}"
        pos += size


def test_full_pipeline_roundtrip_batch(oracle, shafa):
    """F -> T -> C -> D on device-resident blocks in single launches (the shape bench.py uses)."""
    import torch
    import golden.make_golden as mg
    dev = torch.device("cuda:0")
    zt = shafa.zipf_table(1.2)
    nb, bs = 5, 1 << 20
    host = np.concatenate([mg.runs_stream(100 + b, bs, zt) for b in range(nb)])
    host[3 * bs:4 * bs] = 0x33                                       # one block that is a single run
    d_in = torch.from_numpy(host).to(dev)
    st = torch.cuda.Stream()
    bt = shafa.Batch(nb, 2 * bs + 16)
    off = [b * bs for b in range(nb)]
    rcap = 2 * bs + 16
    roff = [b * rcap for b in range(nb)]
    d_rle = torch.empty(nb * rcap, dtype=torch.uint8, device=dev)
    d_rn = torch.zeros(nb, dtype=torch.int64, device=dev)
    d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
    bt.rle_encode(st, d_in, off, [bs] * nb, d_rle, roff, [rcap] * nb, d_rn, d_freq)
    bt.finish(st, nb)
    rn = d_rn.cpu().numpy().astype(np.int64)
    freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
    rle_host = d_rle.cpu().numpy()
    for b in range(nb):
        want = oracle.rle_encode(host[b * bs:(b + 1) * bs])
        got = rle_host[roff[b]:roff[b] + rn[b]]
        assert got.tobytes() == want.tobytes(), f"rle block {b}: {first_diff(got, want)}"
        assert (freq[b] == oracle.hist256(want)).all()
    tables = [shafa.sf_build_codes(freq[b]) for b in range(nb)]
    d_sf = torch.empty(nb * rcap, dtype=torch.uint8, device=dev)
    d_sn = torch.zeros(nb, dtype=torch.int64, device=dev)
    bt.sf_encode(st, d_rle, roff, rn, tables, d_sf, roff, [rcap] * nb, d_sn)
    bt.finish(st, nb)
    sn = d_sn.cpu().numpy().astype(np.int64)
    d_back = torch.empty(nb * rcap, dtype=torch.uint8, device=dev)
    bt.sf_decode(st, d_sf, roff, sn, tables, rn, d_back, roff)
    bt.finish(st, nb)
    back = d_back.cpu().numpy()
    for b in range(nb):
        assert back[roff[b]:roff[b] + rn[b]].tobytes() == rle_host[roff[b]:roff[b] + rn[b]].tobytes(), f"sf roundtrip block {b}"
    d_orig = torch.empty(nb * bs, dtype=torch.uint8, device=dev)
    d_on = torch.zeros(nb, dtype=torch.int64, device=dev)
    bt.rle_decode(st, d_back, roff, rn, d_orig, off, [bs] * nb, d_on)
    bt.finish(st, nb)
    assert (d_on.cpu().numpy() == bs).all()
    assert torch.equal(d_orig, d_in), "D(C(T(F(x)))) != x"


def test_sf_decode_codes_of_14_to_16_bits_fast_path(oracle, shafa):
    """Complete codes with 13 < Lmax <= 16 (steep Zipf): the nibble-ring DP, the counting automaton and the
    three-code write pass with the long codes resolved from the LDS prefix table."""
    for s, seed in ((1.6, 31), (2.0, 32), (2.5, 33)):
        zt = shafa.zipf_table(s)
        for n in (4096, 70001, (1 << 21) + 13):
            data = oracle.gen_bytes(seed + n, n, zt)
            otab = oracle.sf_build(oracle.hist256(data))
            got, enc = decode_roundtrip(oracle, shafa, data, otab)
            assert got.tobytes() == data.tobytes(), f"s={s} n={n} lmax={otab.lens().max()}: {first_diff(got, data)}"
    # the table of a big block (Lmax 15/16) applied to a stream that is dense in its rarest symbols
    zt = shafa.zipf_table(2.0)
    big = oracle.gen_bytes(77, 1 << 24, zt)
    otab = oracle.sf_build(oracle.hist256(big))
    lens = otab.lens()
    assert 13 < lens.max() <= 16, lens.max()
    rare = np.nonzero(lens >= 13)[0].astype(np.uint8)
    mix = big[:300000].copy()
    mix[::3] = rare[np.arange(mix[::3].size) % rare.size]          # every third symbol has a 13..16-bit code
    got, enc = decode_roundtrip(oracle, shafa, mix, otab)
    assert got.tobytes() == mix.tobytes(), first_diff(got, mix)
    # truncated stream on this path is still an error
    rc, _ = shafa.sf_decode(enc[: enc.size // 2], to_shafa_table(shafa, otab), len(mix), raw_rc=True)
    assert rc == shafa.FILE_UNRECOGNIZABLE


def test_sf_decode_batch_mixing_short_and_long_tables(oracle, shafa):
    """One launch whose blocks have Lmax 12 and Lmax 15/16 (all complete): the long-code kernels serve both."""
    import torch
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream()
    blocks = [oracle.gen_bytes(5, 500000, shafa.zipf_table(1.2)), oracle.gen_bytes(6, 400001, shafa.zipf_table(2.0)),
              oracle.gen_bytes(7, 300000, shafa.zipf_table(1.2)), oracle.gen_bytes(8, 1 << 20, shafa.zipf_table(1.6))]
    tabs = [oracle.sf_build(oracle.hist256(b)) for b in blocks]
    assert max(t.lens().max() for t in tabs) > 13 and min(t.lens().max() for t in tabs) <= 13
    encs = []
    for b, t in zip(blocks, tabs):
        rc, e = oracle.sf_encode(b, t)
        assert rc == 0
        encs.append(e)
    cap = ((max(len(e) for e in encs) + 255) // 256) * 256
    ocap = ((max(len(b) for b in blocks) + 255) // 256) * 256
    nb = len(blocks)
    h_in = np.zeros(nb * cap, dtype=np.uint8)
    for i, e in enumerate(encs):
        h_in[i * cap:i * cap + len(e)] = e
    d_in = torch.from_numpy(h_in).to(dev)
    d_out = torch.zeros(nb * ocap, dtype=torch.uint8, device=dev)
    bt = shafa.Batch(nb, ocap)
    bt.sf_decode(st, d_in, [i * cap for i in range(nb)], [len(e) for e in encs], [to_shafa_table(shafa, t) for t in tabs],
                 [len(b) for b in blocks], d_out, [i * ocap for i in range(nb)])
    bt.finish(st, nb)
    out = d_out.cpu().numpy()
    for i, b in enumerate(blocks):
        assert out[i * ocap:i * ocap + len(b)].tobytes() == b.tobytes(), f"block {i}: {first_diff(out[i * ocap:i * ocap + len(b)], b)}"


def _random_complete_table(oracle, rng, nsyms, skew):
    """Shannon-Fano table of a random histogram: `skew` stretches the counts geometrically (deeper trees)."""
    freq = np.zeros(256, dtype=np.uint64)
    syms = rng.permutation(256)[:nsyms]
    w = rng.random(nsyms) ** skew
    freq[syms] = np.maximum(1, (w / w.max() * 2.0 ** 40).astype(np.uint64))
    return oracle.sf_build(freq), syms.astype(np.uint8), w / w.sum()


def test_sf_roundtrip_random_tables_and_sizes(oracle, shafa):
    """Property test over random complete tables (Lmax from 2 to > 16, i.e. every decode path) and ragged sizes:
    the HIP encoder equals the oracle's bytes and the HIP decoder returns the data; sizes straddle chunk (32 B),
    tile (8 KiB) and multi-tile-per-workgroup boundaries of the encoded stream."""
    rng = np.random.default_rng(20261003)
    seen = set()
    for trial in range(40):
        nsyms = int(rng.integers(2, 257))
        skew = float(rng.choice([0.5, 1.0, 3.0, 8.0, 20.0]))
        otab, syms, prob = _random_complete_table(oracle, rng, nsyms, skew)
        lmax = int(otab.lens().max())
        seen.add("<=13" if lmax <= 13 else "14-16" if lmax <= 16 else ">16")
        n = int(rng.choice([1, 5, 31, 257, 4097, 8191, 8193, 40001, 65536 * 4 + 3, 1 << 20]))
        data = syms[rng.choice(nsyms, size=n, p=prob)]
        if n >= nsyms and trial % 3 == 0:
            data[:nsyms] = syms                                    # every symbol, including the rarest codes
        rc, want = oracle.sf_encode(data, otab)
        assert rc == 0
        t = to_shafa_table(shafa, otab)
        got = shafa.sf_encode(data, t)
        assert got.tobytes() == want.tobytes(), f"trial {trial} encode lmax={lmax} n={n}: {first_diff(got, want)}"
        if nsyms >= 2:
            back = shafa.sf_decode(want, t, n)
            assert back.tobytes() == data.tobytes(), f"trial {trial} decode lmax={lmax} n={n} nsyms={nsyms}: {first_diff(back, data)}"
            if want.size > 2:                                       # a truncated stream is an error on every path
                rc2, _ = shafa.sf_decode(want[: want.size - 2], t, n, raw_rc=True)
                assert rc2 == shafa.FILE_UNRECOGNIZABLE, f"trial {trial} lmax={lmax} n={n}: truncated stream accepted"
    assert seen == {"<=13", "14-16", ">16"}, seen


def test_sf_decode_codes_of_17_to_32_bits_fast_path(oracle, shafa):
    """Complete codes with 16 < Lmax <= 32 (a long tail of rare bytes, as in real files at -b M): 32-entry maps, the
    fast DP with the high-exit bit ring, the automaton and the three-code write pass with the 32-bit resolver."""
    rng = np.random.default_rng(7)
    zt = shafa.zipf_table(1.2)
    for n, nrare in ((300000, 40), ((1 << 21) + 77, 56)):
        data = oracle.gen_bytes(400 + n, n, np.where(zt >= 200, zt % 200, zt).astype(np.uint8))
        for k in range(nrare):                                       # 1 .. a few hundred occurrences each
            pos = rng.integers(0, n, size=1 + (k * k) // 8)
            data[pos] = 200 + k
        otab = oracle.sf_build(oracle.hist256(data))
        lmax = int(otab.lens().max())
        assert 16 < lmax <= 32, lmax
        got, enc = decode_roundtrip(oracle, shafa, data, otab)
        assert got.tobytes() == data.tobytes(), f"n={n} lmax={lmax}: {first_diff(got, data)}"
        # dense in the rarest symbols: long codes next to each other and across chunk / tile boundaries
        rare = np.nonzero(otab.lens() > 16)[0].astype(np.uint8)
        mix = data[:200000].copy()
        mix[::2] = rare[np.arange(mix[::2].size) % rare.size]
        got, enc = decode_roundtrip(oracle, shafa, mix, otab)
        assert got.tobytes() == mix.tobytes(), f"dense rare n={n} lmax={lmax}: {first_diff(got, mix)}"
        rc, _ = shafa.sf_decode(enc[: enc.size // 3], to_shafa_table(shafa, otab), len(mix), raw_rc=True)
        assert rc == shafa.FILE_UNRECOGNIZABLE


DEFAULT_OPTIONS = {"sf_encode_one_pass_min_blocks": 0, "sf_encode_lanes": 0, "sf_encode_window_bits": 0,
                   "sf_decode_speculate": 1, "sf_decode_path": 0, "rle_encode_general": 0, "rle_encode_one_pass": 0}


@pytest.mark.parametrize("options", [
    {"sf_encode_one_pass_min_blocks": 1 << 30},                            # count / scan / pack kernels (sfe3_*) for every launch
    {"sf_encode_one_pass_min_blocks": 1},                                  # the one-pass kernel even for one block
    {"sf_encode_one_pass_min_blocks": 1, "sf_encode_lanes": 256},          # ... 256-lane workgroups, 8 KiB tiles
    {"sf_encode_one_pass_min_blocks": 1, "sf_encode_lanes": 512},          # ... 512-lane workgroups, 16 KiB tiles
    {"sf_encode_one_pass_min_blocks": 1, "sf_encode_window_bits": 4},      # ... windows too small: flagged, encoded again (256 lanes)
    {"sf_decode_speculate": 0},                                            # exact DP kernels (sfd_sync16 / sfd_countfsm)
    {"sf_decode_speculate": 2},                                            # speculative entries whatever the code
    {"sf_decode_path": 1},                                                 # tables treated as incomplete: the byte-map kernels
    {"sf_decode_path": 2},                                                 # generic byte-map kernels (sfd_sync / sfd_tiles)
    {"rle_encode_general": 1},                                             # per-element general RLE tile code for every tile
    {"rle_encode_one_pass": 1},                                            # RLE + histogram of its output in one chained pass (rle4_kernel)
    {"rle_encode_one_pass": 1, "rle_encode_general": 1},                   # ... with the general tile code for every unit
], ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()))
def test_alternative_kernel_paths_stay_bit_exact(oracle, shafa, options):
    """Every fall-back / alternative kernel path that the library can be told to take (shafa_hip_set_option) must produce
    the oracle's bytes on the same inputs as the default path."""
    import golden.make_golden as mg
    shafa.lib().shafa_hip_init(0)
    for k, v in options.items():
        shafa.set_option(k, v)
    try:
        zt = shafa.zipf_table(1.2)
        for n in (5000, 300001):
            data = mg.runs_stream(900 + n, n, zt)
            want = oracle.rle_encode(data)
            got, freq = shafa.rle_encode(data, want_freq=True)
            assert got.tobytes() == want.tobytes(), f"{options} rle n={n}: {first_diff(got, want)}"
            assert (freq == oracle.hist256(want)).all(), f"{options} rle histogram n={n}"
            assert shafa.rle_decode(want).tobytes() == data.tobytes(), f"{options} rle_decode n={n}"
            for src in (want, oracle.gen_bytes(n, n, shafa.zipf_table(2.0))):        # Lmax <= 13 and 14..16
                otab = oracle.sf_build(oracle.hist256(src))
                rc, enc = oracle.sf_encode(src, otab)
                t = to_shafa_table(shafa, otab)
                got_enc = shafa.sf_encode(src, t)
                assert got_enc.tobytes() == enc.tobytes(), f"{options} sf_encode n={n}: {first_diff(got_enc, enc)}"
                back = shafa.sf_decode(enc, t, len(src))
                assert back.tobytes() == src.tobytes(), f"{options} sf_decode n={n}: {first_diff(back, src)}"
    finally:
        for k, v in DEFAULT_OPTIONS.items():
            shafa.set_option(k, v)
