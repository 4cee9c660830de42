"""Module T's core on the device (shafa_hipd_sf_build_codes, csrc/sf_tables.hip; reference t.c:74-210) against the host's
shafa_sf_build_codes (host/sfcodes.c, itself checked against the reference's .cod files in test_abi_cpu.py) and against the
oracle's: the same lengths and the same bits for every symbol, whatever the histogram."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def histograms():
    rng = np.random.default_rng(11)
    hs = []
    hs.append(rng.integers(0, 1 << 20, 256))                            # random counts
    hs.append(rng.integers(0, 4, 256))                                  # many ties and zeros
    hs.append(np.full(256, 7))                                          # all equal: 8-bit codes
    z = np.zeros(256, dtype=np.uint64); z[65] = 1000; hs.append(z)      # one symbol: no codes
    z = np.zeros(256, dtype=np.uint64); z[0] = 5; z[255] = 5; hs.append(z)          # two symbols, a tie
    hs.append(np.zeros(256, dtype=np.uint64))                           # empty
    fib = [1, 1]
    while len(fib) < 88:
        fib.append(fib[-1] + fib[-2])
    z = np.zeros(256, dtype=np.uint64); z[100:188] = np.array(fib, dtype=np.uint64)[::-1]; hs.append(z)   # a tree 87 deep
    z = np.array([1 << min(i, 55) for i in range(256)], dtype=np.uint64); hs.append(z)   # geometric, then a long plateau
    z = np.zeros(256, dtype=np.uint64); z[:4] = [1 << 62, 1 << 61, 1 << 61, 1 << 60]; hs.append(z)        # sums near 2^63.6
    ranks = np.arange(1, 257, dtype=np.float64)
    hs.append(np.floor(1e9 / ranks ** 1.2))                             # Zipf counts
    for k in range(12):                                                 # what a block's own histogram looks like
        p = rng.dirichlet(np.full(256, 0.05 + 0.3 * k))
        hs.append(rng.multinomial(1 << 23, p))
    return np.stack([np.asarray(h, dtype=np.uint64) for h in hs])


def test_device_tables_equal_the_hosts(oracle, shafa):
    import torch
    freq = histograms()
    nb = freq.shape[0]
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    bt = shafa.Batch(nb, 1 << 20)
    d_freq = torch.from_numpy(freq.view(np.int64)).to(dev)
    tsz = C.sizeof(shafa.CodeTable)
    d_tab = torch.full((nb * tsz,), 0xEE, dtype=torch.uint8, device=dev)
    bt.sf_build_codes(st, nb, d_freq, d_tab)
    rc, errs = bt.finish(st, nb, raise_on_error=False)
    assert rc == 0 and not any(errs), (rc, errs)
    got = d_tab.cpu().numpy().reshape(nb, tsz)
    want = shafa.sf_build_codes_batch(freq)
    for b in range(nb):
        w = np.frombuffer(bytes(want[b]), dtype=np.uint8)
        assert got[b, :256].tobytes() == w[:256].tobytes(), f"histogram {b}: code lengths differ"
        assert got[b].tobytes() == w.tobytes(), f"histogram {b}: code bits differ"
        if int(freq[b].sum()) < (1 << 31):                               # (the reference's sums are `int`, t.c:133)
            ot = oracle.sf_build(freq[b])
            assert bytes(ot.len) == got[b, :256].tobytes(), f"histogram {b}: lengths differ from the oracle's"


def test_counts_whose_sum_passes_64_bits_are_refused(shafa):
    import torch
    freq = np.zeros((2, 256), dtype=np.uint64)
    freq[0, :3] = [1 << 63, 1 << 63, 5]                                  # the host's case (128-bit sums there)
    freq[1, :3] = [9, 8, 7]
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    bt = shafa.Batch(2, 1 << 20)
    d_freq = torch.from_numpy(freq.view(np.int64)).to(dev)
    tsz = C.sizeof(shafa.CodeTable)
    d_tab = torch.full((2 * tsz,), 0xEE, dtype=torch.uint8, device=dev)
    bt.sf_build_codes(st, 2, d_freq, d_tab)
    rc, errs = bt.finish(st, 2, raise_on_error=False)
    assert errs[0] == shafa.OUTSIDE_MODULE and errs[1] == 0
    got = d_tab.cpu().numpy().reshape(2, tsz)
    assert not got[0].any()                                             # an empty table, not garbage
    w = np.frombuffer(bytes(shafa.sf_build_codes(freq[1])), dtype=np.uint8)
    assert got[1].tobytes() == w.tobytes()
