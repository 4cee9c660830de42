"""GPU parity of the one-pass Shannon-Fano encoder (sf_encode4.hip; compress_to_buffer + binary_coding,
reference c.c:52-237): launches with many blocks, bit-exact against the oracle block by block.  The launch
threshold is lowered / raised through shafa_hip_set_option so that both encoders see the same inputs."""
import numpy as np
import pytest

from test_gpu_parity import first_diff, long_code_case, to_shafa_table

pytestmark = pytest.mark.gpu


@pytest.fixture
def one_pass(shafa):
    """The chained one-pass kernel (sfe5_kernel) whatever the number of blocks in the launch."""
    shafa.lib().shafa_hip_init(0)
    shafa.set_option("sf_encode_one_pass_min_blocks", 1)
    yield
    shafa.set_option("sf_encode_one_pass_min_blocks", 0)


def run_batch(shafa, oracle, blocks, tables, caps=None, expect_err=None):
    """blocks: list of uint8 arrays; tables: oracle tables.  Returns per-block encoded bytes (or checks errors)."""
    import torch
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    nb = len(blocks)
    off, pos = [], 0
    for b in blocks:
        off.append(pos)
        pos += (b.size + 15) // 16 * 16
    host = np.zeros(max(pos, 16), dtype=np.uint8)
    for o, b in zip(off, blocks):
        host[o:o + b.size] = b
    d_in = torch.from_numpy(host).to(dev)
    want = []
    for b, t in zip(blocks, tables):
        rc, enc = oracle.sf_encode(b, t)
        want.append((rc, enc))
    if caps is None:
        caps = [((w[1].size + 15) // 16 + 1) * 16 for w in want]
    ooff, opos = [], 0
    for c in caps:
        ooff.append(opos)
        opos += (c + 15) // 16 * 16 + 64
    d_out = torch.full((opos + 64,), 0xEE, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(nb, dtype=torch.int64, device=dev)
    bt = shafa.Batch(nb, max(max(b.size for b in blocks), 16))
    stabs = [to_shafa_table(shafa, t) for t in tables]
    torch.cuda.synchronize()          # torch's fills run on its own stream: finish them before ours starts
    bt.sf_encode(st, d_in, off, [b.size for b in blocks], stabs, d_out, ooff, caps, d_n)
    rc, errs = bt.finish(st, nb, raise_on_error=False)
    out = d_out.cpu().numpy()
    sizes = d_n.cpu().numpy()
    bad = []
    for i in range(nb):
        exp = expect_err.get(i, 0) if expect_err else 0
        if errs[i] != exp:
            bad.append(f"block {i}: error {errs[i]} expected {exp}")
            continue
        if exp:
            continue
        got = out[ooff[i]:ooff[i] + int(sizes[i])]
        if int(sizes[i]) != want[i][1].size or got.tobytes() != want[i][1].tobytes():
            bad.append(f"block {i} (n={blocks[i].size}, lmax={tables[i].lens().max()}): size {sizes[i]} vs {want[i][1].size}; {first_diff(got, want[i][1])}")
        # nothing written past the block's region (guard bytes stay 0xEE)
        end = ooff[i] + (caps[i] + 15) // 16 * 16
        if not (out[end:end + 64] == 0xEE).all():
            bad.append(f"block {i}: wrote past its output region")
    bt.close()
    assert not bad, "\n".join(bad[:10])
    return out, ooff, sizes


def zipf_blocks(shafa, oracle, sizes, seed0=100):
    zt = shafa.zipf_table(1.2)
    blocks = [oracle.gen_bytes(seed0 + i, n, zt) for i, n in enumerate(sizes)]
    tables = [oracle.sf_build(oracle.hist256(b)) for b in blocks]
    return blocks, tables


SIZES = [1, 2, 15, 16, 17, 31, 33, 255, 4095, 4096, 4097, 8191, 8192, 8193, 16383, 16384, 16385, 24576 + 1, 65536, 65536 + 5,
         100000, 131072, 262144 + 5, 300000, 524288, 1048576 + 77, 2 * 1048576, 3 * 1048576 + 8191, 12345, 77777,
         8192 * 5, 8192 * 7 + 1, 40, 41, 9000, 70000, 200000, 650000, 1 << 20, (1 << 20) + 8192]


def test_one_pass_matches_oracle_ragged_sizes(shafa, oracle, one_pass):
    blocks, tables = zipf_blocks(shafa, oracle, SIZES)
    run_batch(shafa, oracle, blocks, tables)


def test_one_pass_blocks_of_whole_8_kib_tiles(shafa, oracle, one_pass):
    """Every block is a multiple of 8 KiB but not of 32 KiB: no remainder for the 8 KiB tiles of the 256-lane kernel,
    a remainder of one to three quarters of a tile for the 32 KiB tiles of the 1024-lane one (the tail kernel must be
    launched by the form that runs, not by the 8 KiB rule)."""
    sizes = [8192 * (4 * i + 1 + i % 3) for i in range(12)] + [8192, 16384, 24576, 8192 * 127]
    assert all(n % 8192 == 0 and n % 32768 for n in sizes)
    blocks, tables = zipf_blocks(shafa, oracle, sizes, seed0=4000)
    run_batch(shafa, oracle, blocks, tables)


def test_one_pass_and_three_kernel_agree_on_a_big_launch(shafa, oracle):
    """128 blocks of 0.5 MiB + ragged tails through the default dispatch (one pass: the default threshold is 6 blocks for codes of <= 12 bits)."""
    sizes = [(1 << 19) + 13 * i for i in range(128)]
    blocks, tables = zipf_blocks(shafa, oracle, sizes, seed0=500)
    shafa.lib().shafa_hip_init(0)
    run_batch(shafa, oracle, blocks, tables)
    shafa.set_option("sf_encode_one_pass_min_blocks", 1 << 20)       # force count / scan / pack
    try:
        run_batch(shafa, oracle, blocks, tables)
    finally:
        shafa.set_option("sf_encode_one_pass_min_blocks", 0)


@pytest.mark.parametrize("kind", ["uniform", "two", "few", "lmax16", "lmax13", "lmax12", "single_long_chain"])
def test_one_pass_code_length_classes(shafa, oracle, one_pass, kind):
    """NW = 3 (Lmax <= 8), 4 (<= 12), 5 (<= 15) and the Lmax == 16 variant; one block alone = the longest chain."""
    blocks, tables = [], []
    if kind == "uniform":          # 8/9-bit codes
        for i, n in enumerate([70000, 8192, 500000, 33]):
            b = oracle.gen_bytes(900 + i, n)
            blocks.append(b)
            tables.append(oracle.sf_build(oracle.hist256(oracle.gen_bytes(900 + i, 1 << 20))))
    elif kind == "two":            # 1-bit codes: 8 symbols = 8 bits per oct
        for i, n in enumerate([100000, 8191, 65536 + 3]):
            b = (oracle.gen_bytes(910 + i, n) & 1).astype(np.uint8) * 200 + 3
            blocks.append(b)
            tables.append(oracle.sf_build(oracle.hist256(b)))
    elif kind == "few":            # RLE-like: 5 symbols, Lmax <= 8
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
                  np.full(30000, np.nonzero(otab.lens() == 16)[0][0], dtype=np.uint8)]      # only 16-bit codes: 64-bit groups
        tables = [otab] * 3
    elif kind == "lmax12":         # the widest codes of the 1024-lane form: three 48 KiB windows in one CU's LDS
        otab, data = long_code_case(oracle, 400000, 13, 0.5, 8)
        assert 10 < otab.lens().max() <= 12, otab.lens().max()
        rare = np.nonzero(otab.lens() >= 10)[0].astype(np.uint8)
        blocks = [data, data[:16385], rare[oracle.gen_bytes(6, 100000) % rare.size]]     # the last one: only long codes
        tables = [otab] * 3
    elif kind == "lmax13":
        otab, data = long_code_case(oracle, 400000, 14, 0.5, 6)
        assert 12 < otab.lens().max() <= 15, otab.lens().max()
        blocks, tables = [data, data[:8193]], [otab, otab]
    else:                          # one 4 MiB block: 512 tiles on one chain, every workgroup of the grid on it
        zt = shafa.zipf_table(1.2)
        b = oracle.gen_bytes(77, 4 << 20, zt)
        blocks, tables = [b], [oracle.sf_build(oracle.hist256(b))]
    run_batch(shafa, oracle, blocks, tables)


def test_one_pass_error_semantics(shafa, oracle, one_pass):
    """A data symbol without a code -> _FILE_UNRECOGNIZABLE for that block only; a too small output region ->
    _LACK_OF_MEMORY for that block only, nothing written past it."""
    blocks, tables = zipf_blocks(shafa, oracle, [50000] * 6, seed0=700)
    # block 2: table built without symbol 250, which the data contains
    f = oracle.hist256(blocks[2])
    blocks[2] = blocks[2].copy()
    blocks[2][12345] = 250
    f[250] = 0
    tables[2] = oracle.sf_build(f)
    assert tables[2].lens()[250] == 0
    caps = None
    want_sizes = [oracle.sf_encode(b, t)[1].size for b, t in zip(blocks, tables)]
    caps = [((w + 15) // 16 + 1) * 16 for w in want_sizes]
    caps[4] = (want_sizes[4] // 2) // 16 * 16                      # half of what block 4 needs
    run_batch(shafa, oracle, blocks, tables, caps=caps, expect_err={2: shafa.FILE_UNRECOGNIZABLE, 4: shafa.LACK_OF_MEMORY})


def test_one_pass_more_blocks_than_workgroups(shafa, oracle):
    """1300 small blocks in one launch: more blocks than the persistent grid has workgroups (1024), so workgroups move on
    to a second block (table reload, pipeline drain and restart); sizes around the tile size, default dispatch."""
    sizes = [8192 * (1 + i % 4) + (i * 37) % 8192 for i in range(1300)]
    sizes[7] = 8192
    sizes[8] = 5
    blocks, tables = zipf_blocks(shafa, oracle, sizes, seed0=3000)
    shafa.lib().shafa_hip_init(0)
    run_batch(shafa, oracle, blocks, tables)


@pytest.mark.parametrize("window_bits", [3, 6, 7, 9])
def test_one_pass_windows_smaller_than_the_worst_case_encode_again(shafa, oracle, window_bits):
    """The 1024-lane form sizes its LDS windows for at most 12 bits per symbol (13..16-bit codes) and encodes a block whose
    tile does not fit again with the 256-lane form.  `sf_encode_window_bits` shrinks the windows so that ordinary data
    takes that path: 3 = every tile of every block overflows, 6 / 7 = some tiles of some blocks (Zipf(1.2): 6.5 bits per
    symbol on average), 9 = none.  Ragged sizes: the tail kernel of the second pass must touch flagged blocks only."""
    shafa.lib().shafa_hip_init(0)
    sizes = [(1 << 18) + 977 * i for i in range(10)] + [32768, 65536 + 5, 8192 * 9, 40000, 1 << 20]
    blocks, tables = zipf_blocks(shafa, oracle, sizes, seed0=8100)
    # a block whose first half is its rare symbols (long codes) and whose second half its frequent ones: with 7-bit windows
    # the tiles of the first half overflow, those of the second do not
    lens = tables[0].lens()
    rare = np.nonzero(lens >= 8)[0].astype(np.uint8)
    freq = np.nonzero((lens > 0) & (lens <= 5))[0].astype(np.uint8)
    half = 1 << 18
    mixed = np.concatenate([rare[oracle.gen_bytes(1, half) % rare.size], freq[oracle.gen_bytes(2, half) % freq.size]])
    blocks.append(mixed)
    tables.append(tables[0])
    shafa.set_option("sf_encode_one_pass_min_blocks", 1)
    shafa.set_option("sf_encode_window_bits", window_bits)
    try:
        run_batch(shafa, oracle, blocks, tables)
    finally:
        shafa.set_option("sf_encode_window_bits", 0)
        shafa.set_option("sf_encode_one_pass_min_blocks", 0)


def test_one_pass_wide_form_for_codes_of_13_to_16_bits(shafa, oracle):
    """Default dispatch, >= 6 blocks of 13..16-bit codes: the 1024-lane form with 12-bit windows plus the (idle) second
    pass; one block is made of its rarest symbols only (16-bit codes back to back: its tiles do not fit, it is encoded
    again)."""
    shafa.lib().shafa_hip_init(0)
    otab, data = long_code_case(oracle, 600000, 17, 0.5, 3)
    assert otab.lens().max() == 16
    rare = np.nonzero(otab.lens() >= 14)[0].astype(np.uint8)
    blocks = [data, data[:300001], data[5:250000], rare[oracle.gen_bytes(9, 200000) % rare.size], data[:32768], data[:70000],
              data[100:500000], data[:8191]]
    run_batch(shafa, oracle, blocks, [otab] * len(blocks))


def _long_tail_blocks(shafa, oracle, sizes, seed0):
    """Zipf data over 200 common symbols plus 56 rare ones (1 .. a few hundred occurrences): Lmax 17..32, what a real file
    gives at -b M (tools/longtail_time.py)."""
    zt = shafa.zipf_table(1.2)
    zt = np.where(zt >= 200, zt % 200, zt).astype(np.uint8)
    rng = np.random.default_rng(seed0)
    blocks, tables = [], []
    for i, n in enumerate(sizes):
        b = oracle.gen_bytes(seed0 + i, n, zt).copy()
        for k in range(56):
            cnt = 1 + (k * k) // 9
            b[rng.integers(0, n, size=min(cnt, n))] = 200 + k
        blocks.append(b)
        tables.append(oracle.sf_build(oracle.hist256(b)))
    return blocks, tables


@pytest.mark.parametrize("window_bits", [0, 5, 7])
def test_one_pass_codes_of_17_to_32_bits(shafa, oracle, window_bits):
    """The quad form of the one-pass encoder (codes of up to 32 bits: four symbols per unit, windows sized for 12 bits per
    symbol, flagged blocks encoded again by its 256-lane form): ragged sizes, blocks made of their longest codes only (these
    do not fit the windows), a missing code and a short output region in one block among many; with smaller windows the
    second pass also runs for ordinary blocks."""
    shafa.lib().shafa_hip_init(0)
    sizes = [300000 + 1013 * i for i in range(7)] + [32768, 32768 * 3 + 5, 8192 * 5, 70001, (1 << 20) + 77, 33, 4097]
    blocks, tables = _long_tail_blocks(shafa, oracle, sizes, 9100)
    lm = [int(t.lens().max()) for t in tables]
    assert all(l > 16 for l in lm[:7]) and max(lm) <= 32, lm
    lens = tables[0].lens()
    rare = np.nonzero(lens >= 17)[0].astype(np.uint8)
    assert rare.size >= 2
    blocks.append(rare[oracle.gen_bytes(3, 150000) % rare.size])          # only codes of 17+ bits: no tile fits a 12-bit window
    tables.append(tables[0])
    # same-length 32-bit codes next to each other need the 64-bit shift edge cases of the quad: the deepest table we can make
    otab, data = long_code_case(oracle, 120000, 33, 0.5, 11)
    assert otab.lens().max() == 32
    deep = np.nonzero(otab.lens() == 32)[0].astype(np.uint8)
    blocks += [data, deep[oracle.gen_bytes(4, 50000) % deep.size]]
    tables += [otab, otab]
    shafa.set_option("sf_encode_one_pass_min_blocks", 1)
    shafa.set_option("sf_encode_window_bits", window_bits)
    try:
        run_batch(shafa, oracle, blocks, tables)
        # errors stay with their block
        f = oracle.hist256(blocks[2])
        bad = blocks[2].copy()
        bad[777] = 199 if f[199] == 0 else 198
        f2 = f.copy()
        f2[bad[777]] = 0
        t_bad = oracle.sf_build(f2)
        if t_bad.lens()[bad[777]] == 0 and int(t_bad.lens().max()) > 16:
            want = [oracle.sf_encode(b, t)[1].size for b, t in zip(blocks[:6], tables[:6])]
            caps = [((w + 15) // 16 + 1) * 16 for w in want]
            caps[4] = (want[4] // 2) // 16 * 16
            bl, tb = list(blocks[:6]), list(tables[:6])
            bl[2], tb[2] = bad, t_bad
            run_batch(shafa, oracle, bl, tb, caps=caps, expect_err={2: shafa.FILE_UNRECOGNIZABLE, 4: shafa.LACK_OF_MEMORY})
    finally:
        shafa.set_option("sf_encode_window_bits", 0)
        shafa.set_option("sf_encode_one_pass_min_blocks", 0)
