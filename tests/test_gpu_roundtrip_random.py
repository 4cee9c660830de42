"""Randomised round trips through the whole block codec on the GPU (Module F -> C -> D of the reference: f.c:29-55,
c.c:52-237, d.c:116-197 + 466-551): distributions from almost uniform to steep, runs, sizes from a few bytes to several
tiles, launches that mix them.  RLE encode -> histogram -> codes (host Module T) -> SF encode -> SF decode -> RLE decode
must return the input; every stage is also compared with the oracle on the same bytes.  The speculative decoder runs in
its three modes."""
import numpy as np
import pytest

from test_gpu_parity import first_diff, to_shafa_table

pytestmark = pytest.mark.gpu


def make_block(rng, n):
    kind = rng.integers(0, 6)
    if kind == 0:                                   # Zipf with a random exponent over a random alphabet
        s, m = rng.uniform(0.3, 2.5), int(rng.integers(2, 257))
        p = np.arange(1, m + 1, dtype=np.float64) ** (-s)
        x = rng.choice(m, size=n, p=p / p.sum()).astype(np.uint8)
        return rng.permutation(256).astype(np.uint8)[x]
    if kind == 1:                                   # runs of geometric length
        out = np.empty(n, dtype=np.uint8)
        pos, mean = 0, rng.choice([1.3, 3.0, 40.0, 700.0])
        while pos < n:
            L = int(rng.geometric(1.0 / mean))
            out[pos:pos + L] = rng.integers(0, 256 if rng.random() < 0.7 else 4)
            pos += L
        return out
    if kind == 2:
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 3:                                   # zeros with sparse other bytes
        x = np.zeros(n, dtype=np.uint8)
        idx = rng.integers(0, n, max(1, n // int(rng.integers(3, 300))))
        x[idx] = rng.integers(1, 256, idx.size)
        return x
    if kind == 4:                                   # two alternating stretches of different statistics
        a, b = make_block(rng, n // 2 + 1), make_block(rng, n // 2 + 1)
        return np.concatenate([a, b])[:n]
    return np.full(n, rng.integers(0, 256), dtype=np.uint8)


@pytest.mark.parametrize("seed", range(16))
def test_random_round_trips(oracle, shafa, seed):
    rng = np.random.default_rng(1000 + seed)
    sizes = [int(rng.choice([1, 7, 300, 4096, 8191, 8193, 40000, 262144 + 5, 1500000])) for _ in range(7)]
    blocks = [make_block(rng, n) for n in sizes]
    bad = []
    for mode in (1, 2, 0):
        shafa.set_option("sf_decode_speculate", mode)
        for i, b in enumerate(blocks):
            rle, freq = shafa.rle_encode(b, want_freq=True)
            want_rle = oracle.rle_encode(b)
            if rle.tobytes() != want_rle.tobytes():
                bad.append(f"seed {seed} block {i} n={b.size}: rle_encode {first_diff(rle, want_rle)}")
                continue
            otab = oracle.sf_build(oracle.hist256(want_rle))
            if otab.lens().max() == 0:               # one symbol: no code, the reference rejects it
                continue
            tab = to_shafa_table(shafa, otab)
            enc = shafa.sf_encode(rle, tab)
            rc, want_enc = oracle.sf_encode(want_rle, otab)
            if rc != 0 or enc.tobytes() != want_enc.tobytes():
                bad.append(f"seed {seed} block {i} n={b.size}: sf_encode {first_diff(enc, want_enc)}")
                continue
            dec = shafa.sf_decode(enc, tab, rle.size)
            if dec.tobytes() != rle.tobytes():
                bad.append(f"seed {seed} mode {mode} block {i} n={b.size} lmax={otab.lens().max()}: sf_decode {first_diff(dec, rle)}")
                continue
            back = shafa.rle_decode(dec)
            if back.tobytes() != b.tobytes():
                bad.append(f"seed {seed} block {i} n={b.size}: rle_decode {first_diff(back, b)}")
    shafa.set_option("sf_decode_speculate", 1)
    assert not bad, "\n".join(bad[:12])


def test_rle_encode_many_blocks_per_launch(oracle, shafa):
    """A launch of many multi-tile blocks, repeated: every block's RLE bytes and fused histogram against the oracle
    (a missing barrier in the first pass once showed only here, in one block of 32, one launch in three)."""
    import torch
    import golden.make_golden as mg
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    nb, bs = 32, 4 << 20
    zt = shafa.zipf_table(1.2)
    blocks = [mg.runs_stream(70 + (b % 4), bs, zt) for b in range(4)]
    want = [oracle.rle_encode(x) for x in blocks]
    wfreq = [oracle.hist256(w) for w in want]
    d_in = torch.cat([torch.from_numpy(blocks[b % 4]) for b in range(nb)]).to(dev)
    rcap = 2 * bs + 64
    d_rle = torch.empty(nb * rcap, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(nb, dtype=torch.int64, device=dev)
    d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
    bt = shafa.Batch(nb, rcap)
    bad = []
    for it in range(6):
        torch.cuda.synchronize()
        bt.rle_encode(st, d_in, [b * bs for b in range(nb)], [bs] * nb, d_rle, [b * rcap for b in range(nb)], [rcap] * nb, d_n, d_freq)
        bt.finish(st, nb)
        ns = d_n.cpu().numpy()
        fr = d_freq.cpu().numpy().reshape(nb, 256)
        out = d_rle.cpu().numpy()
        for b in range(nb):
            w = want[b % 4]
            got = out[b * rcap:b * rcap + int(ns[b])]
            if got.size != w.size or got.tobytes() != w.tobytes():
                bad.append(f"launch {it} block {b}: {first_diff(got, w)}")
            elif not (fr[b] == wfreq[b % 4]).all():
                bad.append(f"launch {it} block {b}: histogram of the RLE bytes differs")
    bt.close()
    assert not bad, "\n".join(bad[:10])


def test_hundreds_of_small_blocks_per_launch(oracle, shafa):
    """600 blocks of 40-70 KB in one launch of each batch entry point (the reference's default block size is 64 KiB,
    file.h): per-launch records of more than 16 KB travel through the upload kernel (api.hip batch_upload,
    batch_params_commit), not the runtime's copy.  hist256_tiles, rle_encode, sf_encode (with and without tile histograms),
    sf_decode, rle_decode — every block against the oracle."""
    import ctypes as C
    import torch
    import golden.make_golden as mg
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    nb = 600
    zt = shafa.zipf_table(1.2)
    rng = np.random.default_rng(5)
    sizes = [int(x) for x in rng.integers(40000, 70000, nb)]
    base = [mg.runs_stream(300 + i, 70000, zt) for i in range(6)]
    blocks = [base[b % 6][:sizes[b]] for b in range(nb)]
    off, pos = [], 0
    for n in sizes:
        off.append(pos)
        pos += (n + 15) // 16 * 16
    host = np.zeros(pos + 16, dtype=np.uint8)
    for o, b in zip(off, blocks):
        host[o:o + b.size] = b
    d_in = torch.from_numpy(host).to(dev)
    bt = shafa.Batch(nb, 160000)
    bad = []
    # Module F: RLE bytes, their histogram and their tile histograms
    rcap = 2 * 70000 + 64
    roff = [b * rcap for b in range(nb)]
    thb = shafa.tile_hist_bytes(rcap) + 16
    thoff = [b * thb for b in range(nb)]
    d_rle = torch.empty(nb * rcap, dtype=torch.uint8, device=dev)
    d_rn = torch.zeros(nb, dtype=torch.int64, device=dev)
    d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
    d_th = torch.zeros(nb * thb, dtype=torch.uint8, device=dev)
    bt.rle_encode_tiles(st, d_in, off, sizes, d_rle, roff, [rcap] * nb, d_rn, d_freq, d_th, thoff)
    bt.finish(st, nb)
    rn = [int(x) for x in d_rn.cpu().numpy()]
    fr = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
    rle = d_rle.cpu().numpy()
    want_rle = {}
    for b in range(nb):
        key = (b % 6, sizes[b])
        if key not in want_rle:
            want_rle[key] = oracle.rle_encode(blocks[b])
        w = want_rle[key]
        got = rle[roff[b]:roff[b] + rn[b]]
        if got.size != w.size or got.tobytes() != w.tobytes():
            bad.append(f"rle_encode block {b}: {first_diff(got, w)}")
        elif not (fr[b] == oracle.hist256(w)).all():
            bad.append(f"rle_encode block {b}: histogram differs")
    assert not bad, "\n".join(bad[:10])
    # Module T on the host, Module C with the tile histograms and without: the same bytes as the oracle's
    tabs, want_enc = [], []
    for b in range(nb):
        ot = oracle.sf_build(fr[b])
        t = shafa.CodeTable()
        C.memmove(C.byref(t), C.byref(ot), C.sizeof(t))
        tabs.append(t)
        rc, e = oracle.sf_encode(rle[roff[b]:roff[b] + rn[b]], ot)
        assert rc == 0
        want_enc.append(e)
    ecap = 80000
    eoff = [b * ecap for b in range(nb)]
    d_n = torch.zeros(nb, dtype=torch.int64, device=dev)
    for with_tiles in (True, False):
        d_enc = torch.zeros(nb * ecap, dtype=torch.uint8, device=dev)
        if with_tiles:
            bt.sf_encode_tiles(st, d_rle, roff, rn, tabs, d_th, thoff, d_enc, eoff, [ecap] * nb, d_n)
        else:
            bt.sf_encode(st, d_rle, roff, rn, tabs, d_enc, eoff, [ecap] * nb, d_n)
        bt.finish(st, nb)
        en = d_n.cpu().numpy()
        enc = d_enc.cpu().numpy()
        for b in range(nb):
            got = enc[eoff[b]:eoff[b] + int(en[b])]
            if got.size != want_enc[b].size or got.tobytes() != want_enc[b].tobytes():
                bad.append(f"sf_encode (tile histograms: {with_tiles}) block {b}: {first_diff(got, want_enc[b])}")
        assert not bad, "\n".join(bad[:10])
    # Module D: SF decode, then RLE decode, back to the input
    d_sym = torch.zeros(nb * rcap, dtype=torch.uint8, device=dev)
    bt.sf_decode(st, d_enc, eoff, [w.size for w in want_enc], tabs, rn, d_sym, roff)
    bt.finish(st, nb)
    assert torch.equal(d_sym, d_rle) or all(
        torch.equal(d_sym[roff[b]:roff[b] + rn[b]], d_rle[roff[b]:roff[b] + rn[b]]) for b in range(nb)), "sf_decode differs"
    dcap = 70000 + 1040
    doff = [b * 71168 for b in range(nb)]
    d_dec = torch.zeros(nb * 71168, dtype=torch.uint8, device=dev)
    d_dn = torch.zeros(nb, dtype=torch.int64, device=dev)
    bt.rle_decode(st, d_sym, roff, rn, d_dec, doff, [dcap] * nb, d_dn)
    bt.finish(st, nb)
    dn = d_dn.cpu().numpy()
    dec = d_dec.cpu().numpy()
    for b in range(nb):
        got = dec[doff[b]:doff[b] + int(dn[b])]
        if got.size != sizes[b] or got.tobytes() != blocks[b].tobytes():
            bad.append(f"rle_decode block {b}: {first_diff(got, blocks[b])}")
    bt.close()
    assert not bad, "\n".join(bad[:10])
