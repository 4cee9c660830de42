"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/shafa_hip.h declares (no compute calls — there is no GPU here), and the C host's formats and
Module T agree with the reference-generated golden files."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle_lib import parse_blocks_text

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def declared_symbols(header):
    with open(header) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(shafa_(?:hipd?|pipe)_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(shafa):
    syms = declared_symbols(os.path.join(ROOT, "include", "shafa_hip.h"))
    assert len(syms) >= 24, syms
    L = ctypes.CDLL(shafa.LIB_PATH)
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, f"declared in include/shafa_hip.h but not exported: {missing}"
    assert L.shafa_hip_abi_version() == 8
    assert ctypes.sizeof(shafa.CodeTable) == 256 + 256 * 32


def test_library_exports_nothing_but_the_declared_symbols(shafa):
    """exported ⊆ declared: no experiment hooks or undeclared entry points in the product library (dynamic symbol
    table of the .so; internal C++ helpers are allowed only with hidden/mangled names, i.e. not `shafa_*` C names)."""
    import subprocess
    declared = set(declared_symbols(os.path.join(ROOT, "include", "shafa_hip.h")))
    out = subprocess.run(["nm", "-D", "--defined-only", shafa.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set()
    for line in out.splitlines():
        parts = line.split()
        if len(parts) >= 3 and parts[1] in "TtWw" and parts[2].startswith("shafa_"):
            exported.add(parts[2])
    extra = sorted(exported - declared)
    assert not extra, f"exported by libshafa_hip.so but not declared in include/shafa_hip.h: {extra}"
    assert shafa.lib().shafa_hip_set_option(b"no_such_option", 1) == shafa.OUTSIDE_MODULE
    assert shafa.lib().shafa_hip_set_option(b"sf_encode_one_pass_min_blocks", 0) == shafa.SUCCESS
    for v in (0, 2, 1):
        assert shafa.lib().shafa_hip_set_option(b"sf_decode_speculate", v) == shafa.SUCCESS
    for name, good, bad in ((b"sf_encode_lanes", (256, 512, 0), 100),
                            (b"sf_encode_window_bits", (4, 16, 0), 17), (b"sf_decode_path", (1, 2, 0), 3), (b"rle_encode_general", (1, 0), None),
                            (b"rle_encode_one_pass", (0, 1), None)):
        for v in good:
            assert shafa.lib().shafa_hip_set_option(name, v) == shafa.SUCCESS, (name, v)
        if bad is not None:
            assert shafa.lib().shafa_hip_set_option(name, bad) == shafa.OUTSIDE_MODULE, (name, bad)


def test_no_gpu_is_reported_not_faked(shafa):
    """Without a GPU the product path must fail loudly (no CPU fallback)."""
    L = shafa.lib()
    if L.shafa_hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    assert L.shafa_hip_init(0) == shafa.DEVICE_ERROR
    with pytest.raises(shafa.ShafaError) as e:
        shafa.hist256(np.zeros(100, dtype=np.uint8))
    assert e.value.code == shafa.DEVICE_ERROR


def rd(case, fn):
    with open(os.path.join(GOLD, case, fn), "rb") as f:
        return f.read()


@pytest.mark.parametrize("case,stem", [("runs_default", "x.rle"), ("edges_forced_rle", "e.rle"),
                                       ("uniform_no_rle", "u"), ("textlike_m", "t"), ("t_handmade", "h")])
def test_host_module_t_and_formats_match_reference_files(shafa, case, stem):
    fmode, fblocks = parse_blocks_text(rd(case, stem + ".freq"))
    cmode, cblocks = parse_blocks_text(rd(case, stem + ".cod"))
    for (fsize, ftext), (csize, ctext) in zip(fblocks, cblocks):
        rc, freq = shafa.freq_parse(ftext)
        assert rc == 0
        assert shafa.freq_format(freq) == ftext
        tab = shafa.sf_build_codes(freq)
        assert shafa.cod_format(tab) == ctext
        rc, tab2 = shafa.cod_parse(ctext)
        assert rc == 0 and bytes(tab2.bits) == bytes(tab.bits) and bytes(tab2.len) == bytes(tab.len)


def test_host_module_t_batch_equals_block_by_block(shafa):
    """shafa_sf_build_codes_batch (the launch's blocks over threads) = shafa_sf_build_codes per block, whatever the count."""
    rng = np.random.default_rng(5)
    for n in (1, 15, 16, 130):
        freq = rng.integers(0, 1 << 20, size=(n, 256), dtype=np.uint64)
        freq[0, 1:] = 0                                   # one symbol: no codes
        if n > 2:
            freq[1] = 0                                   # empty histogram
            freq[2] = 1 << np.minimum(np.arange(256, dtype=np.uint64), 62)     # geometric: codes of up to 255 bits
        arr = shafa.sf_build_codes_batch(freq)
        assert len(arr) == n
        for b in range(n):
            one = shafa.sf_build_codes(freq[b])
            assert bytes(arr[b].len) == bytes(one.len) and bytes(arr[b].bits) == bytes(one.bits), (n, b)


def test_host_parsers_reject_malformed(shafa):
    assert shafa.freq_parse(b";" * 255)[0] == shafa.FILE_UNRECOGNIZABLE          # field 0 must be a number
    assert shafa.freq_parse(b"1" + b";" * 254)[0] == shafa.FILE_UNRECOGNIZABLE   # 255 fields
    assert shafa.freq_parse(b"1" + b";" * 256)[0] == shafa.FILE_UNRECOGNIZABLE   # 257 fields
    assert shafa.freq_parse(b"1" + b";" * 255)[0] == 0
    assert shafa.cod_parse(b"0;1" + b";" * 254)[0] == 0
    assert shafa.cod_parse(b"0;2" + b";" * 254)[0] == shafa.FILE_UNRECOGNIZABLE
    assert shafa.cod_parse(b"0;1" + b";" * 252)[0] == shafa.FILE_UNRECOGNIZABLE
    assert shafa.cod_parse(b"0" * 256 + b";" * 255)[0] == shafa.FILE_UNRECOGNIZABLE  # > 255 bits


def test_block_split_and_rle_rule(shafa):
    H = shafa.host()
    bs, last = ctypes.c_uint64(65536), ctypes.c_uint64(0)
    assert H.shafa_block_count(212345, ctypes.byref(bs), ctypes.byref(last)) == 4 and last.value == 212345 - 3 * 65536
    assert H.shafa_block_count(131072, ctypes.byref(bs), ctypes.byref(last)) == 2 and last.value == 65536
    bs = ctypes.c_uint64(100)
    H.shafa_block_count(5000, ctypes.byref(bs), ctypes.byref(last))
    assert bs.value == 512                                                     # file.c:61-65 clamp
    assert H.shafa_rle_worthwhile(1000, 950, False) and not H.shafa_rle_worthwhile(1000, 951, False)
    assert H.shafa_rle_worthwhile(1000, 2000, True)
