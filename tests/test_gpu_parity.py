"""GPU parity tests: the HIP path (through the C-ABI of libshafa_hip.so) against the oracle and
against the reference-generated golden files, bit-exact.  Run with `-m gpu` on an MI355X box.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib
from oracle_lib import parse_blocks_text, parse_shaf

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rd(case, fn):
    with open(os.path.join(GOLD, case, fn), "rb") as f:
        return f.read()


def first_diff(a, b):
    a, b = np.frombuffer(bytes(a), dtype=np.uint8), np.frombuffer(bytes(b), dtype=np.uint8)
    if a.size != b.size:
        m = min(a.size, b.size)
        d = np.nonzero(a[:m] != b[:m])[0]
        return f"sizes {a.size} vs {b.size}; first diff at {d[0] if d.size else m}"
    d = np.nonzero(a != b)[0]
    return "equal" if not d.size else f"{d.size} bytes differ, first at {d[0]}: {a[d[0]]} vs {b[d[0]]}"


def to_shafa_table(shafa, otab):
    t = shafa.CodeTable()
    import ctypes as C
    C.memmove(C.byref(t), C.byref(otab), C.sizeof(t))
    return t


def streams(oracle, shafa):
    zt = shafa.zipf_table(1.2)
    out = {
        "uniform": lambda n, seed=1: oracle.gen_bytes(seed, n),
        "zipf": lambda n, seed=2: oracle.gen_bytes(seed, n, zt),
        "same": lambda n, seed=0: np.full(n, 0x5A, dtype=np.uint8),
        "zeros": lambda n, seed=0: np.zeros(n, dtype=np.uint8),
        "two": lambda n, seed=3: (oracle.gen_bytes(seed, n) & 1).astype(np.uint8) * 200 + 3,
    }
    return out


SIZES = [0, 1, 15, 16, 17, 255, 4095, 4096, 4097, 16383, 16384, 16385, 65536, 262144 + 5, 1048576 + 77]


# ----------------------------------------------------------------------------- K1 hist256
@pytest.mark.parametrize("kind", ["uniform", "zipf", "same", "zeros"])
def test_hist256_matches_oracle(oracle, shafa, kind):
    gen = streams(oracle, shafa)[kind]
    for n in SIZES:
        data = gen(n)
        got, want = shafa.hist256(data), oracle.hist256(data)
        assert (got == want).all(), f"{kind} n={n}: bins differ at {np.nonzero(got != want)[0][:8]}"


# ----------------------------------------------------------------------------- K3 sf_encode
def encode_case(oracle, shafa, data, freq_for_table=None):
    freq = oracle.hist256(data) if freq_for_table is None else freq_for_table
    otab = oracle.sf_build(freq)
    rc, want = oracle.sf_encode(data, otab)
    assert rc == 0
    got = shafa.sf_encode(data, to_shafa_table(shafa, otab))
    return got, want, otab


@pytest.mark.parametrize("kind", ["uniform", "zipf", "two"])
def test_sf_encode_matches_oracle_sizes(oracle, shafa, kind):
    gen = streams(oracle, shafa)[kind]
    for n in SIZES[1:]:
        data = gen(n)
        if kind == "uniform" and n < 2:
            continue
        got, want, otab = encode_case(oracle, shafa, data)
        assert got.tobytes() == want.tobytes(), f"{kind} n={n} lmax={otab.lens().max()}: {first_diff(got, want)}"


def long_code_case(oracle, n, nsyms, ratio, seed):
    """Histogram with geometrically decaying counts -> deep Shannon-Fano tree; data uses every symbol."""
    freq = np.zeros(256, dtype=np.uint64)
    for i in range(nsyms):
        freq[(i * 7) % 256] = max(1, int(2.0 ** 50 * ratio ** i))
    otab = oracle.sf_build(freq)
    syms = np.array([(i * 7) % 256 for i in range(nsyms)], dtype=np.uint8)
    # mostly frequent symbols, rare ones sprinkled in
    r = oracle.gen_bytes(seed, n).astype(np.int64)
    idx = np.minimum((r * r * nsyms) // (255 * 255 + 1), nsyms - 1)
    data = syms[idx]
    data[:nsyms] = syms
    return otab, data


def test_sf_encode_long_codes_and_generic(oracle, shafa):
    # G=2 path: code lengths 17..32
    otab, data = long_code_case(oracle, 200000, 30, 0.5, 9)
    assert 16 < otab.lens().max() <= 32, otab.lens().max()
    rc, want = oracle.sf_encode(data, otab)
    got = shafa.sf_encode(data, to_shafa_table(shafa, otab))
    assert rc == 0 and got.tobytes() == want.tobytes(), first_diff(got, want)
    # generic path: codes longer than 32 bits
    otab, data = long_code_case(oracle, 50000, 60, 0.5, 10)
    assert otab.lens().max() > 32, otab.lens().max()
    rc, want = oracle.sf_encode(data, otab)
    got = shafa.sf_encode(data, to_shafa_table(shafa, otab))
    assert rc == 0 and got.tobytes() == want.tobytes(), first_diff(got, want)


def test_sf_encode_expanding_tile_multi_round(oracle, shafa):
    """Only rare symbols (12..16-bit codes): a 16 KiB tile expands to > 24 KiB, more than one LDS window."""
    otab, _ = long_code_case(oracle, 1000, 17, 0.5, 3)
    lens = otab.lens()
    assert lens.max() == 16
    rare = np.nonzero(lens >= 12)[0].astype(np.uint8)
    assert rare.size >= 4
    data = rare[oracle.gen_bytes(4, 70000) % rare.size]
    rc, want = oracle.sf_encode(data, otab)
    got = shafa.sf_encode(data, to_shafa_table(shafa, otab))
    assert rc == 0 and got.tobytes() == want.tobytes(), first_diff(got, want)


def test_sf_encode_edge_semantics(oracle, shafa):
    # single-symbol block: every code empty -> 0 bytes (reference c.c:156)
    f = np.zeros(256, dtype=np.uint64)
    f[7] = 100
    tab = to_shafa_table(shafa, oracle.sf_build(f))
    rc, out = shafa.sf_encode(np.full(100, 7, dtype=np.uint8), tab, raw_rc=True)
    assert rc == 0 and out.size == 0
    # a symbol without a code in a non-empty table
    f[9] = 50
    tab = to_shafa_table(shafa, oracle.sf_build(f))
    rc, _ = shafa.sf_encode(np.array([7, 9, 11] * 100, dtype=np.uint8), tab, raw_rc=True)
    assert rc == shafa.FILE_UNRECOGNIZABLE
    # capacity too small
    data = oracle.gen_bytes(5, 5000)
    otab = oracle.sf_build(oracle.hist256(data))
    rc, _ = shafa.sf_encode(data, to_shafa_table(shafa, otab), cap=1000, raw_rc=True)
    assert rc == shafa.LACK_OF_MEMORY
    # empty block
    rc, out = shafa.sf_encode(np.zeros(0, dtype=np.uint8), to_shafa_table(shafa, otab), raw_rc=True)
    assert rc == 0 and out.size == 0


@pytest.mark.parametrize("case,stem", [("runs_default", "x.rle"), ("edges_forced_rle", "e.rle"),
                                       ("uniform_no_rle", "u"), ("textlike_m", "t"), ("tiny_1024", "a.rle")])
def test_sf_encode_matches_reference_files(shafa, case, stem):
    """HIP encode of the reference's own inputs with the reference's own .cod == its .shaf payloads."""
    data = rd(case, stem)
    _, cblocks = parse_blocks_text(rd(case, stem + ".cod"))
    payloads = parse_shaf(rd(case, stem + ".shaf"))
    pos = 0
    for i, ((size, ctext), payload) in enumerate(zip(cblocks, payloads)):
        rc, tab = shafa.cod_parse(ctext)
        assert rc == 0
        got = shafa.sf_encode(data[pos:pos + size], tab)
        pos += size
        assert got.tobytes() == payload, f"{case} block {i}: {first_diff(got, payload)}"


# ----------------------------------------------------------------------------- batch API + generator
def test_batch_encode_and_generator(oracle, shafa):
    import torch
    dev = torch.device("cuda:0")
    zt = shafa.zipf_table(1.2)
    sizes = [1 << 20, (1 << 20) + 4096, 300000, 16, 700001]
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + ((s + 15) // 16) * 16)
    total = offs[-1]
    d_in = torch.empty(total, dtype=torch.uint8, device=dev)
    d_map = torch.from_numpy(zt).to(dev)
    st = torch.cuda.Stream()
    shafa.gen_bytes(st, 77, 0, d_in, total, d_map)
    st.synchronize()
    host = d_in.cpu().numpy()
    assert (host == oracle.gen_bytes(77, total, zt)).all(), "HIP generator differs from oracle generator"
    d_u = torch.empty(4099, dtype=torch.uint8, device=dev)
    shafa.gen_bytes(st, 5, 1600, d_u, 4099)
    st.synchronize()
    assert (d_u.cpu().numpy() == oracle.gen_bytes(5, 4099, first=1600)).all()

    nb = len(sizes)
    bt = shafa.Batch(nb, 1 << 21)
    tables, want = [], []
    for b in range(nb):
        blk = host[offs[b]:offs[b] + sizes[b]]
        otab = oracle.sf_build(oracle.hist256(blk))
        tables.append(to_shafa_table(shafa, otab))
        rc, enc = oracle.sf_encode(blk, otab)
        assert rc == 0
        want.append(enc)
    # ragged blocks at padded offsets, one launch
    cap = [((s * 2 + 64 + 15) // 16) * 16 for s in sizes]
    out_off = [0]
    for c in cap:
        out_off.append(out_off[-1] + c)
    d_out = torch.empty(out_off[-1], dtype=torch.uint8, device=dev)
    d_out_n = torch.zeros(nb, dtype=torch.int64, device=dev)
    bt.sf_encode(st, d_in, offs[:-1], sizes, tables, d_out, out_off[:-1], cap, d_out_n)
    rc, errs = bt.finish(st, nb)
    assert rc == 0
    ns = d_out_n.cpu().numpy()
    for b in range(nb):
        got = d_out[out_off[b]:out_off[b] + int(ns[b])].cpu().numpy()
        assert got.tobytes() == want[b].tobytes(), f"block {b}: {first_diff(got, want[b])}"
    d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
    bt.hist256(st, d_in, offs[:-1], sizes, d_freq)
    bt.finish(st, nb)
    fr = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
    for b in range(nb):
        assert (fr[b] == oracle.hist256(host[offs[b]:offs[b] + sizes[b]])).all(), f"batch hist block {b}"

    # one launch, many equal-size blocks (the bench shape): 6 x 1 MiB
    nb2, bs = 6, 1 << 20
    d_in2 = torch.empty(nb2 * bs, dtype=torch.uint8, device=dev)
    shafa.gen_bytes(st, 123, 0, d_in2, nb2 * bs, d_map)
    bt2 = shafa.Batch(nb2, bs)
    off2 = [b * bs for b in range(nb2)]
    d_freq2 = torch.zeros(nb2 * 256, dtype=torch.int64, device=dev)
    bt2.hist256(st, d_in2, off2, [bs] * nb2, d_freq2)
    bt2.finish(st, nb2)
    host2 = d_in2.cpu().numpy()
    fr = d_freq2.cpu().numpy().astype(np.uint64).reshape(nb2, 256)
    tabs2 = []
    for b in range(nb2):
        assert (fr[b] == oracle.hist256(host2[b * bs:(b + 1) * bs])).all()
        tabs2.append(shafa.sf_build_codes(fr[b]))
    d_out2 = torch.empty(nb2 * bs, dtype=torch.uint8, device=dev)
    d_n2 = torch.zeros(nb2, dtype=torch.int64, device=dev)
    for rep in range(3):   # repeated launches reuse the workspace / descriptors
        bt2.sf_encode(st, d_in2, off2, [bs] * nb2, tabs2, d_out2, off2, [bs] * nb2, d_n2)
    rc, errs = bt2.finish(st, nb2)
    assert rc == 0 and not any(errs)
    ns2 = d_n2.cpu().numpy()
    for b in range(nb2):
        rc, enc = oracle.sf_encode(host2[b * bs:(b + 1) * bs], oracle.sf_build(fr[b]))
        got = d_out2[off2[b]:off2[b] + int(ns2[b])].cpu().numpy()
        assert got.tobytes() == enc.tobytes(), f"batch block {b}: {first_diff(got, enc)}"
