"""Pin the oracle (oracle/shafa_oracle.c) against the files the REFERENCE wrote (tests/golden/).

The reference has no tests of its own; tests/golden/make_golden.py ran the compiled
reference (oracle/_ref/shafa) on deterministic inputs and stored what it produced.  Every oracle
function is checked here against those bytes, so that the GPU parity tests can trust the oracle.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib
from oracle_lib import parse_blocks_text, parse_shaf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case(name):
    d = os.path.join(GOLD, name)
    with open(os.path.join(d, "manifest.json")) as f:
        man = json.load(f)

    def rd(fn):
        with open(os.path.join(d, fn), "rb") as f:
            return f.read()
    return man, rd


def split_blocks(data, bs):
    return [data[i:i + bs] for i in range(0, len(data), bs)]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


# ----------------------------------------------------------------------------- Module F
@pytest.mark.parametrize("name,fn,bs", [
    ("runs_default", "x", 65536), ("edges_forced_rle", "e", 65536),
    ("uniform_forced_both", "v", 65536), ("runs_force_freq", "w", 65536),
    ("tiny_1024", "a", 65536), ("cli_errors", "z", 65536),
])
def test_rle_encode_and_freq_match_reference(oracle, name, fn, bs):
    man, rd = case(name)
    data = rd(fn)
    rle_ref = rd(fn + ".rle")
    mode, fblocks = parse_blocks_text(rd(fn + ".rle.freq"))
    assert mode == "R"
    blocks = split_blocks(data, bs)
    assert len(blocks) == len(fblocks)
    pos = 0
    for blk, (size, ftext) in zip(blocks, fblocks):
        a = np.frombuffer(blk, dtype=np.uint8)
        r1 = oracle.rle_encode(a)
        r2 = oracle.rle_encode(a, elementwise=True)
        assert r1.tobytes() == r2.tobytes()
        assert r1.size == size
        assert r1.tobytes() == rle_ref[pos:pos + size]
        pos += size
        assert oracle.freq_write_block(oracle.hist256(r1)) == ftext
        rc, f = oracle.freq_parse_block(ftext)
        assert rc == 0 and (f == oracle.hist256(r1)).all()
    assert pos == len(rle_ref)


def test_rle_decision_rule(oracle):
    # accepted cases: the reference produced .rle ; rejected: only .freq (f.c:250-258)
    man, rd = case("runs_default")
    d = rd("x")[:65536]
    assert oracle.rle_accept(65536, oracle.rle_encode(d).size)
    man, rd = case("uniform_no_rle")
    assert "u.rle" not in man["files"] and "u.freq" in man["files"]
    d = rd("u")[:65536]
    assert not oracle.rle_accept(65536, oracle.rle_encode(d).size)
    assert oracle.rle_accept(65536, oracle.rle_encode(d).size, force=True)
    # threshold: (float) ratio < 0.05 rejects; exactly 5 % is accepted
    assert oracle.rle_accept(1000, 950) and not oracle.rle_accept(1000, 951)


@pytest.mark.parametrize("name", ["cfg0_K_runs", "cfg0_K_uniform"])
def test_config0_640KiB_block(oracle, name):
    """BASELINE config[0]: single 640 KiB block, -b K, Module F (input regenerated from its seed)."""
    import golden.make_golden as mg
    man, rd = case(name)
    if name == "cfg0_K_runs":
        data = mg.runs_stream(7, 655360, mg.zipf_table(1.2))
        assert sha(data) == man["files"]["k"]["sha256"]
        rle = oracle.rle_encode(data)
        assert rle.size == man["files"]["k.rle"]["size"] and sha(rle) == man["files"]["k.rle"]["sha256"]
        assert oracle.rle_accept(655360, rle.size)
        mode, fb = parse_blocks_text(rd("k.rle.freq"))
        assert mode == "R" and fb[0][0] == rle.size
        assert oracle.freq_write_block(oracle.hist256(rle)) == fb[0][1]
    else:
        data = mg.gen_bytes(8, 655360)
        assert sha(data) == man["files"]["k"]["sha256"]
        rle = oracle.rle_encode(data)
        assert not oracle.rle_accept(655360, rle.size)
        mode, fb = parse_blocks_text(rd("k.freq"))
        assert mode == "N" and fb[0][0] == 655360
        assert oracle.freq_write_block(oracle.hist256(data)) == fb[0][1]


def test_generator_matches_numpy_model(oracle):
    import golden.make_golden as mg
    zt = mg.zipf_table(1.2)
    for seed, n, first in [(8, 4099, 0), (1234, 1000, 777)]:
        assert (oracle.gen_bytes(seed, n, first=first) == mg.gen_bytes(seed, n, first=first)).all()
        assert (oracle.gen_bytes(seed, n, zt, first) == mg.gen_bytes(seed, n, zt, first)).all()


def test_force_freq_writes_both(oracle):
    man, rd = case("runs_force_freq")
    data = rd("w")
    mode, fb = parse_blocks_text(rd("w.freq"))
    assert mode == "N"
    for blk, (size, ftext) in zip(split_blocks(data, 65536), fb):
        assert size == len(blk)
        assert oracle.freq_write_block(oracle.hist256(blk)) == ftext


# ----------------------------------------------------------------------------- Module T
@pytest.mark.parametrize("name,stem", [
    ("runs_default", "x.rle"), ("edges_forced_rle", "e.rle"), ("uniform_no_rle", "u"),
    ("textlike_m", "t"), ("tiny_1024", "a.rle"), ("t_handmade", "h"),
])
def test_sf_codes_match_reference(oracle, name, stem):
    man, rd = case(name)
    fmode, fblocks = parse_blocks_text(rd(stem + ".freq"))
    cmode, cblocks = parse_blocks_text(rd(stem + ".cod"))
    assert fmode == cmode and len(fblocks) == len(cblocks)
    for (fsize, ftext), (csize, ctext) in zip(fblocks, cblocks):
        assert fsize == csize
        rc, freq = oracle.freq_parse_block(ftext)
        assert rc == 0
        tab = oracle.sf_build(freq)
        assert oracle.cod_write_block(tab) == ctext
        rc, tab2 = oracle.cod_parse_block(ctext)
        assert rc == 0
        assert bytes(tab2.len) == bytes(tab.len) and bytes(tab2.bits) == bytes(tab.bits)


def test_handmade_code_shapes(oracle):
    man, rd = case("t_handmade")
    _, cblocks = parse_blocks_text(rd("h.cod"))
    assert cblocks[0][1] == b";" * 255                      # single symbol: every code empty
    rc, ties = oracle.cod_parse_block(cblocks[1][1])
    assert rc == 0 and set(ties.lens()) == {8}              # 256 equal frequencies: 8-bit codes
    rc, fib = oracle.cod_parse_block(cblocks[2][1])
    assert rc == 0 and fib.lens().max() >= 25                # Fibonacci: a deep tree


# ----------------------------------------------------------------------------- Module C
@pytest.mark.parametrize("name,stem,bs", [
    ("runs_default", "x.rle", None), ("edges_forced_rle", "e.rle", None),
    ("uniform_no_rle", "u", 65536), ("textlike_m", "t", 8388608), ("tiny_1024", "a.rle", None),
])
def test_sf_encode_matches_reference(oracle, name, stem, bs):
    man, rd = case(name)
    data = rd(stem)
    _, cblocks = parse_blocks_text(rd(stem + ".cod"))
    payloads = parse_shaf(rd(stem + ".shaf"))
    assert len(payloads) == len(cblocks)
    pos = 0
    for (size, ctext), payload in zip(cblocks, payloads):
        blk = data[pos:pos + size]
        pos += size
        rc, tab = oracle.cod_parse_block(ctext)
        assert rc == 0
        rc, enc = oracle.sf_encode(blk, tab)
        assert rc == 0
        assert enc.tobytes() == payload
    assert pos == len(data)


# ----------------------------------------------------------------------------- Module D
@pytest.mark.parametrize("name,stem", [
    ("runs_default", "x.rle"), ("edges_forced_rle", "e.rle"), ("uniform_no_rle", "u"),
    ("textlike_m", "t"),
])
def test_sf_decode_matches_reference(oracle, name, stem):
    man, rd = case(name)
    _, cblocks = parse_blocks_text(rd(stem + ".cod"))
    payloads = parse_shaf(rd(stem + ".shaf"))
    dec = b""
    for (size, ctext), payload in zip(cblocks, payloads):
        rc, tab = oracle.cod_parse_block(ctext)
        rc, out = oracle.sf_decode(payload, tab, size)
        assert rc == 0
        dec += out.tobytes()
    # the reference's own SF-only decode gave back exactly the encoder's input file
    assert dec == rd(stem)
    key = {"x.rle": "decoded__sf_only", "u": "decoded__sf", "t": "decoded__sf"}.get(stem)
    if key:
        assert sha(dec) == man["files"][key]["sha256"]


@pytest.mark.parametrize("name,fn,key", [
    ("runs_default", "x", "decoded__rle_only"), ("edges_forced_rle", "e", "decoded__sf_rle"),
])
def test_rle_decode_matches_reference(oracle, name, fn, key):
    man, rd = case(name)
    rle = rd(fn + ".rle")
    _, fblocks = parse_blocks_text(rd(fn + ".rle.freq"))
    pos, dec = 0, b""
    for size, _ in fblocks:
        rc, out = oracle.rle_decode(rle[pos:pos + size])
        assert rc == 0
        dec += out.tobytes()
        pos += size
    assert sha(dec) == man["files"][key]["sha256"]
    assert dec == rd(fn)


# ----------------------------------------------------------------------------- defined edge behaviour
def test_edge_semantics(oracle):
    # {0,s,0} behaves as one literal s (d.c:179-184); triple cut by the block end is an error here
    rc, out = oracle.rle_decode(bytes([0, 65, 0, 66, 0, 67, 3]))
    assert rc == 0 and out.tobytes() == b"AB" + b"CCC"
    assert oracle.rle_decode(bytes([65, 0, 66]))[0] == 4
    assert oracle.rle_decode(bytes([65, 0]))[0] == 4
    # zero byte is always escaped, even alone (f.c:40)
    assert oracle.rle_encode(bytes([0, 5, 0, 0])).tobytes() == bytes([0, 0, 1, 5, 0, 0, 2])
    # empty input
    assert oracle.rle_encode(b"").size == 0
    rc, out = oracle.rle_decode(b"")
    assert rc == 0 and out.size == 0
    # single-symbol block: every code empty, 0 bytes out; decode refuses
    f = np.zeros(256, dtype=np.uint64)
    f[7] = 100
    tab = oracle.sf_build(f)
    assert tab.lens().max() == 0
    rc, enc = oracle.sf_encode(bytes([7] * 100), tab)
    assert rc == 0 and enc.size == 0
    assert oracle.sf_decode(b"", tab, 100)[0] == 4
    # symbol without a code in a non-empty table
    f[9] = 50
    tab = oracle.sf_build(f)
    assert oracle.sf_encode(bytes([7, 9, 11]), tab)[0] == 4
    # RLE decode output limit 64 MiB + 1 KiB (d.c:165-168)
    n_tr = (oracle_lib.RLE_DECODE_MAX // 255) + 1
    big = np.tile(np.array([0, 1, 255], dtype=np.uint8), n_tr)
    assert oracle.rle_decode(big, cap=oracle_lib.RLE_DECODE_MAX)[0] == 4


def test_roundtrip_property_random_tables(oracle):
    rng = np.random.RandomState(5)
    for trial in range(8):
        nsym = rng.randint(2, 257)
        w = rng.rand(nsym) ** rng.randint(1, 8)
        data = rng.choice(nsym, size=rng.randint(1, 5000), p=w / w.sum()).astype(np.uint8)
        tab = oracle.sf_build(oracle.hist256(data))
        rc, enc = oracle.sf_encode(data, tab)
        assert rc == 0
        bits = int((tab.lens()[data].astype(np.int64)).sum())
        assert enc.size == (bits + 7) // 8
        rc, dec = oracle.sf_decode(enc, tab, data.size)
        assert rc == 0 and (dec == data).all()
        rle = oracle.rle_encode(data)
        rc, back = oracle.rle_decode(rle)
        assert rc == 0 and (back == data).all()


def test_structured_mixed_file_pins_the_oracle(oracle):
    """BASELINE config[2] stand-in with real-file structure (binary section in block 0, dictionary text after it,
    3 x 64 MiB at -b M): the oracle's RLE sizes, RLE-byte histograms, block-0 RLE verdict and Shannon-Fano tables equal
    what the reference binary wrote (.rle.freq / .rle.cod stored; .rle / .rle.shaf pinned by SHA-256 in the manifest and
    replayed on the GPU box)."""
    import golden.make_golden as mg
    man, rd = case("full_mixed_M")
    data = mg.make_input(man["generators"]["s"])
    assert sha(data.tobytes()) == man["files"]["s"]["sha256"], "generator drifted from the fixture"
    fmode, fblocks = parse_blocks_text(rd("s.rle.freq"))
    cmode, cblocks = parse_blocks_text(rd("s.rle.cod"))
    assert fmode == "R" and cmode == "R" and len(fblocks) == 3 == len(cblocks)
    bs = 64 << 20
    h = hashlib.sha256()
    for b, ((fsize, ftext), (csize, ctext)) in enumerate(zip(fblocks, cblocks)):
        r = oracle.rle_encode(data[b * bs:(b + 1) * bs])
        assert r.size == fsize == csize, f"block {b}: RLE size {r.size} vs the reference's {fsize}"
        h.update(r.tobytes())
        freq = oracle.hist256(r)
        assert oracle.freq_write_block(freq) == ftext, f"block {b}: .rle.freq"
        assert oracle.cod_write_block(oracle.sf_build(freq)) == ctext, f"block {b}: .rle.cod"
        if b == 0:      # f.c:250-258: block 0's gain (>= 5 %) switches RLE on for the text blocks too
            assert (bs - r.size) / bs >= 0.05
        else:           # ... which would have declined on their own
            assert (bs - r.size) / bs < 0.05
    assert h.hexdigest() == man["files"]["s.rle"]["sha256"], ".rle differs from the reference's"
