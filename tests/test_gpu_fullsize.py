"""Full-size parity (-m gpu): the 8 MiB / 64 MiB block cases of tests/golden/full_* through the LAYER-2 batch
entry points (device buffers, many blocks per launch), assembled into the reference's on-disk formats and
compared with the SHA-256 of the files the reference binary wrote (tests/golden/make_golden.py ran it); the
same sessions go through the CLI in tests/test_cli.py.  Every stage is also compared with the oracle on the
same block.  Reference: f.c:231-356 (F block loop), t.c:246-445, c.c:360-421, d.c:694-764."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BLOCK = {"m": 8 << 20, "M": 64 << 20}


def _sha(b):
    return hashlib.sha256(b).hexdigest()


def _manifest(case):
    with open(os.path.join(GOLD, case, "manifest.json")) as f:
        return json.load(f)


def _blocks(n, bs):
    """fsize block split (file.c:52-117): ceil(n / bs) blocks, the last one holds what is left — however little (the 1 KiB
    rule of f.c:220,366 is about the FILE: a last block of one byte is a block, tests/golden/edge_tail_1)."""
    nb = n // bs
    rem = n - nb * bs
    return [bs] * nb + ([rem] if rem else [])


def _al(x, a=256):
    return (x + a - 1) // a * a


class Session:
    """One reference session (F -> T -> C, then D) on the layer-2 batch API, device resident."""

    def __init__(self, shafa, case):
        import torch
        import golden.make_golden as mg
        self.torch, self.shafa, self.case = torch, shafa, case
        self.man = _manifest(case)
        (self.fn, gen), = self.man["generators"].items()
        argv0 = self.man["cmds"][0]["argv"]
        self.bs = BLOCK[argv0[argv0.index("-b") + 1]]
        self.force_rle = "-c" in argv0 and argv0[argv0.index("-c") + 1] == "r"
        self.data = mg.make_input(gen)
        assert _sha(self.data.tobytes()) == self.man["files"][self.fn]["sha256"]
        self.sizes = _blocks(self.data.size, self.bs)
        self.dev = torch.device("cuda", 0)
        self.st = torch.cuda.Stream(device=self.dev)
        shafa.lib().shafa_hip_init(0)
        self.nb = len(self.sizes)
        self.bt = shafa.Batch(self.nb, 2 * max(self.sizes) + 64)

    def expect(self, name):
        return self.man["files"][name]

    def close(self):
        self.bt.close()


def _freq_file(shafa, mode, sizes, freqs):
    out = b"@" + mode + b"@" + str(len(sizes)).encode()
    for n, f in zip(sizes, freqs):
        out += b"@" + str(n).encode() + b"@" + shafa.freq_format(f)
    return out + b"@0"


def _cod_file(shafa, mode, sizes, tables):
    out = b"@" + mode + b"@" + str(len(sizes)).encode()
    for n, t in zip(sizes, tables):
        out += b"@" + str(n).encode() + b"@" + shafa.cod_format(t)
    return out + b"@0"


@pytest.mark.parametrize("case", ["full_uniform_m", "full_zipf_M", "full_zipfmod_M_forced_rle", "full_single_run_M",
                                  "full_alt01_M", "full_longtail_M", "full_mixed_M"])
def test_layer2_batches_reproduce_reference_files(case, shafa, oracle):
    S = Session(shafa, case)
    torch, bt, st, dev, nb = S.torch, S.bt, S.st, S.dev, S.nb
    try:
        d_in = torch.from_numpy(S.data).to(dev)
        off, pos = [], 0
        for n in S.sizes:
            off.append(pos)
            pos += n
        assert all(o % 16 == 0 for o in off)

        # ---- Module F: RLE of every block + histogram of the RLE bytes (f.c:248,310); block 0 decides (f.c:250-258)
        rcap = _al(2 * max(S.sizes) + 3)
        roff = [b * rcap for b in range(nb)]
        d_rle = torch.empty(nb * rcap, dtype=torch.uint8, device=dev)
        d_rle_n = torch.zeros(nb, dtype=torch.int64, device=dev)
        d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()      # torch's fills run on its own stream: finish them before ours starts
        bt.rle_encode(st, d_in, off, S.sizes, d_rle, roff, [rcap] * nb, d_rle_n, d_freq)
        bt.finish(st, nb)
        rle_n = [int(x) for x in d_rle_n.cpu().numpy()]
        use_rle = shafa.host().shafa_rle_worthwhile(S.sizes[0], rle_n[0], S.force_rle)
        stem = S.fn + (".rle" if use_rle else "")
        assert (stem + ".shaf") in S.man["files"], f"{case}: RLE decision differs from the reference's"
        if use_rle:
            rle_bytes = b"".join(d_rle[roff[b]:roff[b] + rle_n[b]].cpu().numpy().tobytes() for b in range(nb))
            assert len(rle_bytes) == S.expect(stem)["size"]
            assert _sha(rle_bytes) == S.expect(stem)["sha256"], f"{case}: .rle differs from the reference's"
            # oracle on block 0 (bit-exact)
            want0 = oracle.rle_encode(S.data[:S.sizes[0]])
            assert rle_bytes[:rle_n[0]] == want0.tobytes()
            del rle_bytes
            src, soff, ssz = d_rle, roff, rle_n
        else:
            bt.hist256(st, d_in, off, S.sizes, d_freq)
            bt.finish(st, nb)
            src, soff, ssz = d_in, off, S.sizes
        freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
        ffile = _freq_file(shafa, b"R" if use_rle else b"N", ssz, freq)
        assert _sha(ffile) == S.expect(stem + ".freq")["sha256"], f"{case}: .freq differs"

        # ---- Module T (host) and Module C (c.c:360-421)
        tables = [shafa.sf_build_codes(freq[b]) for b in range(nb)]
        cfile = _cod_file(shafa, b"R" if use_rle else b"N", ssz, tables)
        assert _sha(cfile) == S.expect(stem + ".cod")["sha256"], f"{case}: .cod differs"
        lens = np.stack([t.lens() for t in tables]).astype(np.uint64)
        enc_bytes = [int(x) for x in ((freq * lens).sum(axis=1) + 7) // 8]
        cap = _al(max(enc_bytes) + 4096)
        eoff = [b * cap for b in range(nb)]
        d_enc = torch.empty(nb * cap, dtype=torch.uint8, device=dev)
        d_enc_n = torch.zeros(nb, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        bt.sf_encode(st, src, soff, ssz, tables, d_enc, eoff, [cap] * nb, d_enc_n)
        bt.finish(st, nb)
        assert [int(x) for x in d_enc_n.cpu().numpy()] == enc_bytes
        shaf = b"@" + str(nb).encode()
        for b in range(nb):
            shaf += b"@" + str(enc_bytes[b]).encode() + b"@" + d_enc[eoff[b]:eoff[b] + enc_bytes[b]].cpu().numpy().tobytes()
        assert len(shaf) == S.expect(stem + ".shaf")["size"]
        assert _sha(shaf) == S.expect(stem + ".shaf")["sha256"], f"{case}: .shaf differs from the reference's"
        # oracle on the last block (ragged where the case has one)
        lb = nb - 1
        import ctypes as C
        otab = oracle.sf_build(freq[lb])
        assert bytes(otab.len) == bytes(tables[lb].len) and bytes(otab.bits) == bytes(tables[lb].bits)
        blk = src[soff[lb]:soff[lb] + ssz[lb]].cpu().numpy()
        rc, want = oracle.sf_encode(blk, otab, cap=enc_bytes[lb] + 16)
        assert rc == 0 and want.tobytes() == d_enc[eoff[lb]:eoff[lb] + enc_bytes[lb]].cpu().numpy().tobytes()
        del shaf, blk, want

        # ---- Module D: SF decode (+ RLE decode), d.c:694-764
        d_sym = torch.empty(nb * rcap if use_rle else pos, dtype=torch.uint8, device=dev)
        bt.sf_decode(st, d_enc, eoff, enc_bytes, tables, ssz, d_sym, soff)
        bt.finish(st, nb)
        for b in range(nb):
            assert torch.equal(d_sym[soff[b]:soff[b] + ssz[b]], src[soff[b]:soff[b] + ssz[b]]), f"{case}: SF decode, block {b}"
        if use_rle:
            dcap = _al(max(S.sizes) + 1024)
            doff = [b * dcap for b in range(nb)]
            d_dec = torch.empty(nb * dcap, dtype=torch.uint8, device=dev)
            d_dec_n = torch.zeros(nb, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            bt.rle_decode(st, d_sym, soff, ssz, d_dec, doff, [dcap] * nb, d_dec_n)
            bt.finish(st, nb)
            assert [int(x) for x in d_dec_n.cpu().numpy()] == S.sizes
            for b in range(nb):
                assert torch.equal(d_dec[doff[b]:doff[b] + S.sizes[b]], d_in[off[b]:off[b] + S.sizes[b]]), \
                    f"{case}: RLE decode, block {b}"
    finally:
        S.close()


def test_single_run_block_is_263172_triples_plus_remainder(shafa, oracle):
    """f.c:38-52 cap/remainder logic on a 64 MiB single-byte block (SURVEY.md §7 hard case), layer 1."""
    n = 64 << 20
    data = np.full(n, 0x41, dtype=np.uint8)
    got = shafa.rle_encode(data)
    assert got.size == 3 * 263172 + 3
    t = got.reshape(-1, 3)
    assert (t[:-1] == np.array([0, 0x41, 255], dtype=np.uint8)).all() and t[-1].tolist() == [0, 0x41, 4]
    assert got.tobytes() == oracle.rle_encode(data).tobytes()
    back = shafa.rle_decode(got)
    assert back.size == n and (back == 0x41).all()
