"""GPU parity of the Shannon-Fano decoder's two newer passes (sf_decode.hip; reference d.c:171-300 decode_shafa):
  * sfd_scan: speculative chunk entries, verified exactly, with a per-block fall-back to the exact kernels;
  * sfd_wstage: the symbol pass staged through an LDS image of the output.
Every launch is decoded with "sf_decode_speculate" 0 (exact kernels only), 1 (host heuristic) and 2 (speculate
whatever the code: exercises the device-side fall-back); the three must give the block's original bytes, and
nothing may be written outside a block's n_symbols."""
import numpy as np
import pytest

from test_gpu_parity import first_diff, to_shafa_table

pytestmark = pytest.mark.gpu


@pytest.fixture()
def modes(shafa):
    shafa.lib().shafa_hip_init(0)
    yield (0, 1, 2)
    shafa.set_option("sf_decode_speculate", 1)


def decode_batch(shafa, oracle, blocks, tables, mode, out_shift=0):
    """Encode with the oracle, decode the launch on the GPU; returns a list of problems (empty = parity)."""
    import torch
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    enc = []
    for b, t in zip(blocks, tables):
        rc, e = oracle.sf_encode(b, t)
        assert rc == 0
        enc.append(e)
    off, pos = [], 0
    for e in enc:
        off.append(pos)
        pos += (e.size + 15) // 16 * 16
    host = np.zeros(max(pos, 16), dtype=np.uint8)
    for o, e in zip(off, enc):
        host[o:o + e.size] = e
    d_in = torch.from_numpy(host).to(dev)
    ooff, opos = [], out_shift
    for b in blocks:
        ooff.append(opos)
        opos += (b.size + 15) // 16 * 16 + 48               # block outputs start 16-byte aligned (shafa_hip.h)
    d_out = torch.full((opos + 64,), 0xEE, dtype=torch.uint8, device=dev)
    bt = shafa.Batch(len(blocks), max(max(e.size for e in enc), 16))
    stabs = [to_shafa_table(shafa, t) for t in tables]
    torch.cuda.synchronize()
    shafa.set_option("sf_decode_speculate", mode)
    bt.sf_decode(st, d_in, off, [e.size for e in enc], stabs, [b.size for b in blocks], d_out, ooff)
    rc, errs = bt.finish(st, len(blocks), raise_on_error=False)
    out = d_out.cpu().numpy()
    bt.close()
    bad = []
    for i, b in enumerate(blocks):
        want_rc = oracle.sf_decode(enc[i], tables[i], b.size)[0] if tables[i].lens().max() == 0 else 0
        if errs[i] != want_rc:                              # a one-symbol block has no code: the reference rejects it too
            bad.append(f"mode {mode} block {i}: error {errs[i]}, the oracle says {want_rc}")
        if errs[i] or want_rc:
            continue
        got = out[ooff[i]:ooff[i] + b.size]
        if got.tobytes() != b.tobytes():
            bad.append(f"mode {mode} block {i} (n={b.size}, lmax={tables[i].lens().max()}): {first_diff(got, b)}")
        end = ooff[i] + b.size
        nxt = ooff[i + 1] if i + 1 < len(blocks) else end + 48
        assert nxt - end >= 48
        if not (out[end:nxt] == 0xEE).all():
            bad.append(f"mode {mode} block {i}: wrote past its n_symbols")
        if i == 0 and out_shift and not (out[:out_shift] == 0xEE).all():
            bad.append(f"mode {mode}: wrote in front of the first block")
    return bad


def check(shafa, oracle, blocks, modes, **kw):
    tables = [oracle.sf_build(oracle.hist256(b)) for b in blocks]
    bad = []
    for m in modes:
        bad += decode_batch(shafa, oracle, blocks, tables, m, **kw)
    assert not bad, "\n".join(bad[:12])


def zipfmod(oracle, seed, n):
    import golden.make_golden as mg
    return oracle.gen_bytes(seed, n, mg.zipf_mod256_table(1.2))


def skewed(seed, n, p0, nsym=7):
    """byte 0 with probability p0, the rest spread over nsym - 1 other bytes: codes of 1 .. few bits."""
    rng = np.random.default_rng(seed)
    x = rng.integers(1, nsym, size=n, dtype=np.uint8)
    x[rng.random(n) < p0] = 0
    return x


SIZES = [1, 2, 17, 255, 4096, 8191, 8192, 8193, 10085, 32768 + 3, 65536, 262144 + 5, 1048576 + 77, 3 * 1048576 + 1]


def test_spec_sizes_zipfmod(oracle, shafa, modes):
    # ragged sizes: the stream ends inside a strip, a chunk, a tile, a region
    check(shafa, oracle, [zipfmod(oracle, 500 + i, n) for i, n in enumerate(SIZES)], modes)


def test_spec_fallback_uniform_and_fixed_length(oracle, shafa, modes):
    # uniform bytes (8/9-bit codes) and a fixed-length code never re-synchronise: mode 2 speculates, fails its
    # verification on the device and the blocks take the exact kernels
    uni = [oracle.gen_bytes(900 + i, n) for i, n in enumerate([70000, 1048576 + 3])]
    rng = np.random.default_rng(5)
    flat = np.tile(np.arange(256, dtype=np.uint8), 2048)          # equal counts: every code has 8 bits
    rng.shuffle(flat)
    check(shafa, oracle, uni + [flat], modes)


def test_spec_mixed_launch(oracle, shafa, modes):
    # one launch, different tables: some blocks speculate, some do not, one has a single symbol, one two symbols
    import golden.make_golden as mg
    zt = shafa.zipf_table(1.2)
    blocks = [zipfmod(oracle, 31, 300000), oracle.gen_bytes(32, 200000), oracle.gen_bytes(33, 150001, zt),
              np.full(70000, 9, dtype=np.uint8), skewed(34, 250000, 0.5, nsym=2), zipfmod(oracle, 35, 65536 * 3)]
    check(shafa, oracle, blocks, modes, out_shift=16)


def test_wstage_dense_tiles(oracle, shafa, modes):
    # codes of 1-2 bits: a tile holds several times more symbols than bytes, the output image goes in rounds
    blocks = [skewed(40, 2 * 1048576 + 9, 0.93), skewed(41, 1048576, 0.80, nsym=4), skewed(42, 300001, 0.99, nsym=3)]
    check(shafa, oracle, blocks, modes, out_shift=48)


def test_wstage_dense_next_to_sparse(oracle, shafa, modes):
    # the image is sized for the launch's densest block; a block that is dense only in places (a long stretch of
    # the most frequent byte inside Zipf data) overflows it there and nowhere else
    a = zipfmod(oracle, 50, 1048576)
    a[300000:500000] = np.bincount(a).argmax()
    check(shafa, oracle, [a, oracle.gen_bytes(51, 500000)], modes)


def test_spec_many_blocks(oracle, shafa, modes):
    # a launch of many small blocks: tiles per workgroup shrink, regions end after one tile
    blocks = [zipfmod(oracle, 600 + i, 8192 * (1 + i % 5) + 13 * i) for i in range(40)]
    check(shafa, oracle, blocks, modes)


def rare_tail(oracle, seed, n, ncommon, nrare, div):
    """Zipf data over `ncommon` bytes plus `nrare` bytes that occur 1 + k^3 / div times: the long tail a real file has at
    a large block size (the recipe of tools/longtail_time.py)."""
    import golden.make_golden as mg
    zt = mg.zipf_mod256_table(1.2)
    t = np.where(zt >= ncommon, zt % ncommon, zt).astype(np.uint8)
    a = oracle.gen_bytes(seed, n, t).copy()
    rng = np.random.default_rng(seed)
    for k, s in enumerate(range(ncommon, ncommon + nrare)):
        a[rng.integers(0, n, size=1 + (k * k * k) // div)] = s
    return a


def test_spec_long_codes(oracle, shafa, modes):
    # codes of 14-16 bits (one launch) and of 17-32 bits (another): the long code is an escape inside the walk; mode 2
    # forces the speculation, mode 0 the exact kernels, both must agree with the oracle
    mid = [rare_tail(oracle, 70, 100000, 150, 12, 60), rare_tail(oracle, 70, 200000, 150, 10, 60)]
    for b in mid:
        lm = oracle.sf_build(oracle.hist256(b)).lens().max()
        assert 13 < lm <= 16, lm
    check(shafa, oracle, mid, modes)
    deep = [rare_tail(oracle, 70, 1048579, 200, 56, 600), rare_tail(oracle, 70, 3145728, 200, 56, 200),
            rare_tail(oracle, 70, 300000, 200, 40, 2000)]
    for b in deep:
        lm = oracle.sf_build(oracle.hist256(b)).lens().max()
        assert 16 < lm <= 32, lm
    check(shafa, oracle, deep, modes, out_shift=32)


def test_many_blocks_with_malformed_tables_among_them(oracle, shafa, modes):
    """A launch of 48 blocks with 48 different tables (the launcher prepares them on helper threads, sf_decode.hip) of
    which two are not prefix-free: those two blocks — and only those — report FILE_UNRECOGNIZABLE, the others decode."""
    import torch
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    blocks = [zipfmod(oracle, 900 + i, 8192 * (1 + i % 4) + 7 * i) for i in range(48)]
    tables = [oracle.sf_build(oracle.hist256(b)) for b in blocks]
    enc = [oracle.sf_encode(b, t)[1] for b, t in zip(blocks, tables)]
    stabs = [to_shafa_table(shafa, t) for t in tables]
    broken = (5, 33)
    for i in broken:                                        # symbol b takes symbol a's code: a duplicate, not prefix-free
        used = [s for s in range(256) if stabs[i].len[s]]
        a, b = used[0], used[1]
        stabs[i].len[b] = stabs[i].len[a]
        for q in range(32):
            stabs[i].bits[b][q] = stabs[i].bits[a][q]
    off, pos = [], 0
    for e in enc:
        off.append(pos)
        pos += (e.size + 15) // 16 * 16
    host = np.zeros(pos, dtype=np.uint8)
    for o, e in zip(off, enc):
        host[o:o + e.size] = e
    d_in = torch.from_numpy(host).to(dev)
    ooff, opos = [], 0
    for b in blocks:
        ooff.append(opos)
        opos += (b.size + 15) // 16 * 16 + 48
    d_out = torch.full((opos + 64,), 0xEE, dtype=torch.uint8, device=dev)
    for mode in modes:
        shafa.set_option("sf_decode_speculate", mode)
        d_out.fill_(0xEE)
        bt = shafa.Batch(len(blocks), max(e.size for e in enc))
        torch.cuda.synchronize()
        bt.sf_decode(st, d_in, off, [e.size for e in enc], stabs, [b.size for b in blocks], d_out, ooff)
        rc, errs = bt.finish(st, len(blocks), raise_on_error=False)
        out = d_out.cpu().numpy()
        bt.close()
        for i, b in enumerate(blocks):
            if i in broken:
                assert errs[i] == shafa.FILE_UNRECOGNIZABLE, (mode, i, errs[i])
                assert (out[ooff[i]:ooff[i] + b.size] == 0xEE).all(), f"mode {mode}: block {i} has no valid table and was written"
            else:
                assert errs[i] == 0, (mode, i, errs[i])
                assert out[ooff[i]:ooff[i] + b.size].tobytes() == b.tobytes(), f"mode {mode} block {i}: {first_diff(out[ooff[i]:ooff[i] + b.size], b)}"
