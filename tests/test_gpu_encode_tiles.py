"""GPU parity of the tile-histogram path of Module F -> Module C (include/shafa_hip.h "Tile histograms"):
shafa_hipd_hist256_tiles (make_freq, reference f.c:63-79, plus the histogram of every 32 KiB tile) and
shafa_hipd_sf_encode_tiles (compress_to_buffer + binary_coding, c.c:52-237, as a one-shot grid whose tiles know their
output offsets before the launch: sf_encode6.hip).  Everything is compared bit for bit with the oracle, block by block."""
import numpy as np
import pytest

from test_gpu_encode_onepass import _long_tail_blocks, zipf_blocks
from test_gpu_parity import first_diff, long_code_case, to_shafa_table

pytestmark = pytest.mark.gpu

TILE = 32768


def np_tile_hist(b):
    nt = (b.size + TILE - 1) // TILE
    out = np.zeros((nt, 256), dtype=np.uint16)
    for t in range(nt):
        out[t] = np.bincount(b[t * TILE:(t + 1) * TILE], minlength=256).astype(np.uint16)
    return out


def layout(sizes, align=16, pad=0):
    off, pos = [], 0
    for n in sizes:
        off.append(pos)
        pos += (n + align - 1) // align * align + pad
    return off, pos


def run_tiles(shafa, oracle, blocks, tables, caps=None, expect_err=None, corrupt_hist_of=None, garbage_hist_of=None):
    """hist256_tiles -> (check sidecar and histogram) -> sf_encode_tiles -> compare with the oracle."""
    import torch
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    nb = len(blocks)
    sizes = [b.size for b in blocks]
    off, pos = layout(sizes)
    host = np.zeros(max(pos, 16), dtype=np.uint8)
    for o, b in zip(off, blocks):
        host[o:o + b.size] = b
    d_in = torch.from_numpy(host).to(dev)
    toff, tpos = layout([shafa.tile_hist_bytes(n) for n in sizes], pad=16)
    d_th = torch.full((tpos + 16,), 0xAB, dtype=torch.uint8, device=dev)
    d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
    bt = shafa.Batch(nb, max(max(sizes), 16))
    torch.cuda.synchronize()
    bt.hist256_tiles(st, d_in, off, sizes, d_freq, d_th, toff)
    bt.finish(st, nb)
    th = d_th.cpu().numpy()
    freq = d_freq.cpu().numpy().reshape(nb, 256)
    bad = []
    for i, b in enumerate(blocks):
        want = np_tile_hist(b)
        got = th[toff[i]:toff[i] + want.size * 2].view(np.uint16).reshape(want.shape)
        if not np.array_equal(got, want):
            t = int(np.nonzero((got != want).any(axis=1))[0][0])
            bad.append(f"block {i} (n={b.size}): tile histogram {t} of {want.shape[0]} differs")
        if not np.array_equal(freq[i].astype(np.uint64), oracle.hist256(b)):
            bad.append(f"block {i}: block histogram differs from make_freq")
        end = toff[i] + want.size * 2
        if not (th[end:end + 16] == 0xAB).all():
            bad.append(f"block {i}: wrote past its tile histograms")
    assert not bad, "\n".join(bad[:10])
    if corrupt_hist_of is not None:                     # move one count of a tile from one symbol to another
        i = corrupt_hist_of
        v = d_th[toff[i]:toff[i] + 512].cpu().numpy().view(np.uint16).copy()
        lens = tables[i].lens()
        a = int(np.argmax(v))                            # a symbol that occurs ...
        cand = [s for s in range(256) if lens[s] and lens[s] != lens[a]]
        v[a] -= 1
        v[cand[0]] += 1                                  # ... counted as one with another code length
        d_th[toff[i]:toff[i] + 512] = torch.from_numpy(v.view(np.uint8)).to(dev)

    for i, mode in (garbage_hist_of or {}).items():     # the whole sidecar of a block is something else
        nbytes = shafa.tile_hist_bytes(sizes[i])
        if mode == "ff":
            d_th[toff[i]:toff[i] + nbytes] = 0xFF
        else:
            rng = np.random.default_rng(99 + i)
            d_th[toff[i]:toff[i] + nbytes] = torch.from_numpy(rng.integers(0, 256, nbytes, dtype=np.uint8)).to(dev)

    want = [oracle.sf_encode(b, t) for b, t in zip(blocks, tables)]
    if caps is None:
        caps = [((w[1].size + 15) // 16 + 1) * 16 for w in want]
    ooff, opos = layout(caps, pad=64)
    d_out = torch.full((opos + 64,), 0xEE, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(nb, dtype=torch.int64, device=dev)
    stabs = [to_shafa_table(shafa, t) for t in tables]
    torch.cuda.synchronize()
    bt.sf_encode_tiles(st, d_in, off, sizes, stabs, d_th, toff, d_out, ooff, caps, d_n)
    rc, errs = bt.finish(st, nb, raise_on_error=False)
    out = d_out.cpu().numpy()
    got_n = d_n.cpu().numpy()
    for i in range(nb):
        exp = expect_err.get(i, 0) if expect_err else 0
        end = ooff[i] + (caps[i] + 15) // 16 * 16
        if not (out[end:end + 64] == 0xEE).all():
            bad.append(f"block {i}: wrote past its output region")
        if errs[i] != exp:
            bad.append(f"block {i}: error {errs[i]} expected {exp}")
            continue
        if exp:
            continue
        got = out[ooff[i]:ooff[i] + int(got_n[i])]
        if int(got_n[i]) != want[i][1].size or got.tobytes() != want[i][1].tobytes():
            bad.append(f"block {i} (n={blocks[i].size}, lmax={tables[i].lens().max()}): size {got_n[i]} vs {want[i][1].size}; "
                       f"{first_diff(got, want[i][1])}")
    bt.close()
    assert not bad, "\n".join(bad[:10])


SIZES = [1, 15, 16, 17, 33, 255, 4097, 8191, 8192, 8193, 32767, 32768, 32769, 65536, 65536 + 5, 98304, 100000, 131072,
         262144 + 5, 300000, 524288, 1048576 + 77, 2 * 1048576, 3 * 1048576 + 8191, 12345, 77777, 32768 * 5, 32768 * 7 + 1,
         40, 9000, 70000, 650000, (1 << 20) + 32768]


def test_tiles_match_oracle_ragged_sizes(shafa, oracle):
    shafa.lib().shafa_hip_init(0)
    blocks, tables = zipf_blocks(shafa, oracle, SIZES, seed0=6100)
    run_tiles(shafa, oracle, blocks, tables)


def test_tiles_one_block_alone_and_many_small_ones(shafa, oracle):
    """No launch is too small (no chain to keep short) or too wide: one 4 MiB block, then 700 blocks around the tile size."""
    shafa.lib().shafa_hip_init(0)
    blocks, tables = zipf_blocks(shafa, oracle, [4 << 20], seed0=6200)
    run_tiles(shafa, oracle, blocks, tables)
    sizes = [32768 * (1 + i % 3) + (i * 37) % 32768 for i in range(700)]
    sizes[7], sizes[8], sizes[9] = 32768, 5, 65536
    blocks, tables = zipf_blocks(shafa, oracle, sizes, seed0=6300)
    run_tiles(shafa, oracle, blocks, tables)


@pytest.mark.parametrize("kind", ["uniform", "two", "few", "lmax16", "lmax13", "lmax12"])
def test_tiles_code_length_classes(shafa, oracle, kind):
    """NW = 3 (Lmax <= 8), 4 (<= 12), 5 (<= 15) and the Lmax == 16 variant of sfe6_kernel; blocks made of their longest
    codes only fill the LDS buffers to their worst case."""
    shafa.lib().shafa_hip_init(0)
    blocks, tables = [], []
    if kind == "uniform":          # 8/9-bit codes
        for i, n in enumerate([70000, 32768, 500000, 33]):
            blocks.append(oracle.gen_bytes(900 + i, n))
            tables.append(oracle.sf_build(oracle.hist256(oracle.gen_bytes(900 + i, 1 << 20))))
    elif kind == "two":            # 1-bit codes: the shortest strings (a sub-tile is 64 pieces)
        for i, n in enumerate([100000, 32767, 65536 + 3, 32768 * 4]):
            b = (oracle.gen_bytes(910 + i, n) & 1).astype(np.uint8) * 200 + 3
            blocks.append(b)
            tables.append(oracle.sf_build(oracle.hist256(b)))
    elif kind == "few":            # RLE-like: 5 symbols
        for i, n in enumerate([300000, 12345]):
            b = (oracle.gen_bytes(920 + i, n) % 5).astype(np.uint8) * 50
            blocks.append(b)
            tables.append(oracle.sf_build(oracle.hist256(b)))
    elif kind == "lmax16":
        otab, _ = long_code_case(oracle, 1000, 17, 0.5, 3)
        assert otab.lens().max() == 16
        syms = np.nonzero(otab.lens())[0].astype(np.uint8)
        rare = np.nonzero(otab.lens() >= 12)[0].astype(np.uint8)
        blocks = [syms[oracle.gen_bytes(4, 200000) % syms.size], rare[oracle.gen_bytes(5, 70000) % rare.size],
                  np.full(100000, np.nonzero(otab.lens() == 16)[0][0], dtype=np.uint8)]      # only 16-bit codes
        tables = [otab] * 3
    elif kind == "lmax12":
        otab, data = long_code_case(oracle, 400000, 13, 0.5, 8)
        assert 10 < otab.lens().max() <= 12, otab.lens().max()
        rare = np.nonzero(otab.lens() >= 10)[0].astype(np.uint8)
        blocks = [data, data[:32769], rare[oracle.gen_bytes(6, 100000) % rare.size]]
        tables = [otab] * 3
    else:
        otab, data = long_code_case(oracle, 400000, 14, 0.5, 6)
        assert 12 < otab.lens().max() <= 15, otab.lens().max()
        blocks, tables = [data, data[:65537]], [otab, otab]
    run_tiles(shafa, oracle, blocks, tables)


def test_tiles_mixed_with_longer_codes_in_one_launch(shafa, oracle):
    """Blocks of 17..32-bit codes and of longer hand-made ones have no tile path: in the same launch they take the kernels
    they take without tile histograms; the others are not disturbed."""
    shafa.lib().shafa_hip_init(0)
    blocks, tables = zipf_blocks(shafa, oracle, [200000, 65536, 70000], seed0=6400)
    lb, lt = _long_tail_blocks(shafa, oracle, [300000, 250000], 9300)
    assert all(int(t.lens().max()) > 16 for t in lt)
    otab, data = long_code_case(oracle, 60000, 40, 0.5, 11)
    assert otab.lens().max() > 32
    blocks = [blocks[0], lb[0], blocks[1], data, lb[1], blocks[2]]
    tables = [tables[0], lt[0], tables[1], otab, lt[1], tables[2]]
    run_tiles(shafa, oracle, blocks, tables)


def test_tiles_error_semantics(shafa, oracle):
    """A data symbol without a code -> _FILE_UNRECOGNIZABLE for that block only; a too small output region ->
    _LACK_OF_MEMORY for that block only, nothing written past it; tile histograms that are not the block's own ->
    _OUTSIDE_MODULE for that block only, nothing written outside its region."""
    shafa.lib().shafa_hip_init(0)
    blocks, tables = zipf_blocks(shafa, oracle, [150000] * 7, seed0=6500)
    f = oracle.hist256(blocks[2])
    blocks[2] = blocks[2].copy()
    blocks[2][12345] = 250
    f[250] = 0
    tables[2] = oracle.sf_build(f)
    assert tables[2].lens()[250] == 0
    want_sizes = [oracle.sf_encode(b, t)[1].size for b, t in zip(blocks, tables)]
    caps = [((w + 15) // 16 + 1) * 16 for w in want_sizes]
    caps[4] = (want_sizes[4] // 2) // 16 * 16
    run_tiles(shafa, oracle, blocks, tables, caps=caps,
              expect_err={2: shafa.FILE_UNRECOGNIZABLE, 4: shafa.LACK_OF_MEMORY, 5: shafa.OUTSIDE_MODULE}, corrupt_hist_of=5)


def test_tiles_garbage_sidecar_is_refused(shafa, oracle):
    """Tile histograms that are not histograms of the block at all (all 0xFFFF, random bytes: an uninitialised or foreign
    sidecar, 40 tiles so that the offsets scan over more than one wave's worth) -> _OUTSIDE_MODULE for that block only,
    nothing written outside its region; its neighbours encode as usual."""
    shafa.lib().shafa_hip_init(0)
    blocks, tables = zipf_blocks(shafa, oracle, [32768 * 40 + 123, 150000, 32768 * 40, 150000, 70000], seed0=6600)
    run_tiles(shafa, oracle, blocks, tables, garbage_hist_of={0: "ff", 2: "random", 4: "random"},
              expect_err={0: shafa.OUTSIDE_MODULE, 2: shafa.OUTSIDE_MODULE, 4: shafa.OUTSIDE_MODULE})


@pytest.mark.parametrize("one_pass", [0, 1])
def test_tiles_of_rle_output_feed_the_encoder(shafa, oracle, one_pass):
    """(one_pass: the two-pass RLE kernels and a histogram pass, or the chained rle4_kernel that counts its own output.)
    F -> T -> C on the device: shafa_hipd_rle_encode_tiles (block_compression + make_freq of the RLE bytes, f.c:248,310)
    leaves the tile histograms of the RLE bytes, whose sizes only the device knows at that point; Module T on the host;
    shafa_hipd_sf_encode_tiles encodes the RLE bytes.  Compared with the oracle's F -> T -> C per block."""
    import torch
    shafa.lib().shafa_hip_init(0)
    synth = __import__("pkgload").load_submodule("synth")
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    sizes = [300000, 65536, 32768 * 9 + 11, 1 << 20, 40000]
    blocks = [synth.runs_stream(50 + i, n, shafa.zipf_table(1.2)) for i, n in enumerate(sizes)]
    nb = len(blocks)
    off, pos = layout(sizes)
    host = np.zeros(pos, dtype=np.uint8)
    for o, b in zip(off, blocks):
        host[o:o + b.size] = b
    d_in = torch.from_numpy(host).to(dev)
    rcap = [2 * n + 16 for n in sizes]
    roff, rpos = layout(rcap)
    toff, tpos = layout([shafa.tile_hist_bytes(c) for c in rcap])
    d_rle = torch.empty(rpos, dtype=torch.uint8, device=dev)
    d_rle_n = torch.zeros(nb, dtype=torch.int64, device=dev)
    d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
    d_th = torch.full((tpos,), 0x5A, dtype=torch.uint8, device=dev)          # garbage: the sidecar is written, not added to
    bt = shafa.Batch(nb, max(rcap))
    torch.cuda.synchronize()
    shafa.set_option("rle_encode_one_pass", one_pass)
    try:
        bt.rle_encode_tiles(st, d_in, off, sizes, d_rle, roff, rcap, d_rle_n, d_freq, d_th, toff)
        bt.finish(st, nb)
    finally:
        shafa.set_option("rle_encode_one_pass", 0)
    rle_n = [int(x) for x in d_rle_n.cpu().numpy()]
    freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
    rle = d_rle.cpu().numpy()
    th = d_th.cpu().numpy()
    stabs, want = [], []
    for i, b in enumerate(blocks):
        orle = oracle.rle_encode(b)
        assert rle_n[i] == orle.size and rle[roff[i]:roff[i] + rle_n[i]].tobytes() == orle.tobytes()
        assert np.array_equal(freq[i], oracle.hist256(orle))
        w = np_tile_hist(orle)
        assert np.array_equal(th[toff[i]:toff[i] + w.size * 2].view(np.uint16).reshape(w.shape), w), f"block {i}"
        otab = oracle.sf_build(freq[i])
        stabs.append(to_shafa_table(shafa, otab))
        want.append(oracle.sf_encode(orle, otab)[1])
    caps = [((w.size + 15) // 16 + 1) * 16 for w in want]
    ooff, opos = layout(caps)
    d_out = torch.empty(opos, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(nb, dtype=torch.int64, device=dev)
    bt.sf_encode_tiles(st, d_rle, roff, rle_n, stabs, d_th, toff, d_out, ooff, caps, d_n)
    bt.finish(st, nb)
    out = d_out.cpu().numpy()
    for i in range(nb):
        assert int(d_n[i]) == want[i].size and out[ooff[i]:ooff[i] + want[i].size].tobytes() == want[i].tobytes(), f"block {i}"
    bt.close()


def test_tiles_at_full_block_size_equal_the_chained_encoder(shafa, oracle):
    """8 x 64 MiB (-b M) of the headline stream: the tile path and the chained one-pass encoder (itself pinned against the
    reference's .shaf files in test_gpu_fullsize.py) must write the same bytes; encoded sizes = sum(freq * len) / 8."""
    import torch
    shafa.lib().shafa_hip_init(0)
    synth = __import__("pkgload").load_submodule("synth")
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    nb, bs = 8, 64 << 20
    d_in = torch.empty(nb * bs, dtype=torch.uint8, device=dev)
    d_map = torch.from_numpy(synth.zipf_mod256_table(1.2)).to(dev)
    shafa.gen_bytes(st, 424242, 0, d_in, nb * bs, d_map)
    off, sizes = [b * bs for b in range(nb)], [bs] * nb
    thb = shafa.tile_hist_bytes(bs)
    toff = [b * thb for b in range(nb)]
    d_th = torch.zeros(nb * thb, dtype=torch.uint8, device=dev)
    d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
    bt = shafa.Batch(nb, bs)
    torch.cuda.synchronize()
    bt.hist256_tiles(st, d_in, off, sizes, d_freq, d_th, toff)
    bt.finish(st, nb)
    freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
    tables = [shafa.sf_build_codes(freq[b]) for b in range(nb)]
    enc = [int((freq[b] * tables[b].lens().astype(np.uint64)).sum() + 7) // 8 for b in range(nb)]
    cap = (max(enc) + 4096 + 255) // 256 * 256
    ooff = [b * cap for b in range(nb)]
    d_a = torch.zeros(nb * cap, dtype=torch.uint8, device=dev)
    d_b = torch.zeros(nb * cap, dtype=torch.uint8, device=dev)
    d_na = torch.zeros(nb, dtype=torch.int64, device=dev)
    d_nb = torch.zeros(nb, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    bt.sf_encode_tiles(st, d_in, off, sizes, tables, d_th, toff, d_a, ooff, [cap] * nb, d_na)
    bt.sf_encode(st, d_in, off, sizes, tables, d_b, ooff, [cap] * nb, d_nb)
    bt.finish(st, nb)
    assert [int(x) for x in d_na.cpu().numpy()] == enc and [int(x) for x in d_nb.cpu().numpy()] == enc
    for b in range(nb):
        assert torch.equal(d_a[ooff[b]:ooff[b] + enc[b]], d_b[ooff[b]:ooff[b] + enc[b]]), f"block {b} differs"
    # and the oracle on the head of block 0 (first 2 MiB with the block's table: a prefix of the stream)
    head = d_in[:2 << 20].cpu().numpy()
    otab = oracle.sf_build(freq[0])
    rc, want = oracle.sf_encode(head, otab)
    got = d_a[:want.size - 1].cpu().numpy()
    assert got.tobytes() == want[:want.size - 1].tobytes()
    bt.close()
