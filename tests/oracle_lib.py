"""ctypes binding of oracle/libshafa_oracle.so — the CPU restatement used as the parity checker.

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libshafa_oracle.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "shafa")

RLE_DECODE_MAX = 67108864 + 1024


class CodeTable(C.Structure):
    _fields_ = [("len", C.c_uint8 * 256), ("bits", (C.c_uint8 * 32) * 256)]

    def lens(self):
        return np.ctypeslib.as_array(self.len).copy()

    def code_str(self, s):
        l = self.len[s]
        return "".join("1" if (self.bits[s][b >> 3] >> (7 - (b & 7))) & 1 else "0" for b in range(l))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        u8p, u64p, szp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)
        tp = C.POINTER(CodeTable)
        lib.orc_hist256.argtypes = [u8p, C.c_size_t, u64p]
        lib.orc_hist256.restype = None
        lib.orc_rle_encode.argtypes = [u8p, C.c_size_t, u8p]
        lib.orc_rle_encode.restype = C.c_size_t
        lib.orc_rle_encode_elementwise.argtypes = [u8p, C.c_size_t, u8p]
        lib.orc_rle_encode_elementwise.restype = C.c_size_t
        lib.orc_rle_accept.argtypes = [C.c_size_t, C.c_size_t, C.c_int]
        lib.orc_rle_accept.restype = C.c_int
        lib.orc_freq_write_block.argtypes = [u64p, C.c_char_p]
        lib.orc_freq_write_block.restype = C.c_size_t
        lib.orc_freq_parse_block.argtypes = [C.c_char_p, u64p]
        lib.orc_freq_parse_block.restype = C.c_int
        lib.orc_sf_build.argtypes = [u64p, tp]
        lib.orc_sf_build.restype = None
        lib.orc_cod_write_block.argtypes = [tp, C.c_char_p]
        lib.orc_cod_write_block.restype = C.c_size_t
        lib.orc_cod_parse_block.argtypes = [C.c_char_p, tp]
        lib.orc_cod_parse_block.restype = C.c_int
        lib.orc_sf_encode.argtypes = [u8p, C.c_size_t, tp, u8p, C.c_size_t, szp]
        lib.orc_sf_encode.restype = C.c_int
        lib.orc_sf_decode.argtypes = [u8p, C.c_size_t, tp, u8p, C.c_size_t]
        lib.orc_sf_decode.restype = C.c_int
        lib.orc_rle_decode.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, szp]
        lib.orc_rle_decode.restype = C.c_int
        lib.orc_gen_bytes.argtypes = [C.c_uint64, C.c_uint64, u8p, u8p, C.c_size_t]
        lib.orc_gen_bytes.restype = None

    # -- thin numpy wrappers -------------------------------------------------
    @staticmethod
    def _arr(data):
        a = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else data
        return np.ascontiguousarray(a, dtype=np.uint8)

    def hist256(self, data):
        a = self._arr(data)
        f = np.zeros(256, dtype=np.uint64)
        self.lib.orc_hist256(_u8p(a), a.size, f.ctypes.data_as(C.POINTER(C.c_uint64)))
        return f

    def rle_encode(self, data, elementwise=False):
        a = self._arr(data)
        out = np.empty(2 * a.size + 3, dtype=np.uint8)
        fn = self.lib.orc_rle_encode_elementwise if elementwise else self.lib.orc_rle_encode
        n = fn(_u8p(a), a.size, _u8p(out))
        assert n != C.c_size_t(-1).value
        return out[:n].copy()

    def rle_accept(self, n0, rle0, force=False):
        return bool(self.lib.orc_rle_accept(n0, rle0, int(force)))

    def freq_write_block(self, freq):
        f = np.ascontiguousarray(freq, dtype=np.uint64)
        buf = C.create_string_buffer(256 * 21 + 1)
        n = self.lib.orc_freq_write_block(f.ctypes.data_as(C.POINTER(C.c_uint64)), buf)
        return buf.raw[:n]

    def freq_parse_block(self, text):
        f = np.zeros(256, dtype=np.uint64)
        rc = self.lib.orc_freq_parse_block(text, f.ctypes.data_as(C.POINTER(C.c_uint64)))
        return rc, f

    def sf_build(self, freq):
        f = np.ascontiguousarray(freq, dtype=np.uint64)
        t = CodeTable()
        self.lib.orc_sf_build(f.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(t))
        return t

    def cod_write_block(self, tab):
        buf = C.create_string_buffer(33152 + 8)
        n = self.lib.orc_cod_write_block(C.byref(tab), buf)
        return buf.raw[:n]

    def cod_parse_block(self, text):
        t = CodeTable()
        rc = self.lib.orc_cod_parse_block(text, C.byref(t))
        return rc, t

    def sf_encode(self, data, tab, cap=None):
        a = self._arr(data)
        cap = cap if cap is not None else a.size * 32 + 8
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n = C.c_size_t(0)
        rc = self.lib.orc_sf_encode(_u8p(a), a.size, C.byref(tab), _u8p(out), cap, C.byref(n))
        return rc, out[:n.value].copy()

    def sf_decode(self, data, tab, n_symbols):
        a = self._arr(data)
        out = np.empty(max(n_symbols, 1), dtype=np.uint8)
        rc = self.lib.orc_sf_decode(_u8p(a), a.size, C.byref(tab), _u8p(out), n_symbols)
        return rc, out[:n_symbols].copy()

    def rle_decode(self, data, cap=None):
        a = self._arr(data)
        cap = cap if cap is not None else min(a.size * 255 + 8, RLE_DECODE_MAX)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n = C.c_size_t(0)
        rc = self.lib.orc_rle_decode(_u8p(a), a.size, _u8p(out), cap, C.byref(n))
        return rc, out[:n.value].copy()

    def gen_bytes(self, seed, n, table=None, first=0):
        out = np.empty(n, dtype=np.uint8)
        tp = _u8p(np.ascontiguousarray(table, dtype=np.uint8)) if table is not None else None
        self.lib.orc_gen_bytes(seed, first, tp, _u8p(out), n)
        return out


def build():
    """Compile the restatement (and, when /root/reference is present, the reference binary)."""
    subprocess.run(["make", "-C", ORACLE_DIR, "--no-print-directory"], check=True,
                   stdout=subprocess.DEVNULL)


_cached = None


def load():
    global _cached
    if _cached is None and os.environ.get("SHAFA_ORACLE_LIB"):      # tools/san/run_san.sh: the build under ASan + UBSan
        _cached = Oracle(C.CDLL(os.environ["SHAFA_ORACLE_LIB"]))
    if _cached is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(
                os.path.join(ORACLE_DIR, "shafa_oracle.c")):
            build()
        _cached = Oracle(C.CDLL(LIB))
    return _cached


# ---------------------------------------------------------------- on-disk format helpers (tests)
def parse_blocks_text(data):
    """Split '@<R|N>@<n>' + n * '@<size>@<payload>' + '@0' (.freq / .cod) -> (mode, [(size, payload)])."""
    assert data[:1] == b"@"
    mode = data[1:2].decode()
    parts = data.split(b"@")
    # ['', mode, n, size0, payload0, size1, payload1, ..., '0']
    n = int(parts[2])
    blocks = [(int(parts[3 + 2 * i]), parts[4 + 2 * i]) for i in range(n)]
    assert parts[3 + 2 * n] == b"0"
    return mode, blocks


def parse_shaf(data):
    """'@<n>' + n * ('@<size>@' + size raw bytes) -> [payload bytes]."""
    assert data[:1] == b"@"
    p = 1
    q = data.index(b"@", p)
    n = int(data[p:q])
    out = []
    p = q
    for _ in range(n):
        assert data[p:p + 1] == b"@"
        q = data.index(b"@", p + 1)
        size = int(data[p + 1:q])
        out.append(data[q + 1:q + 1 + size])
        p = q + 1 + size
    assert p == len(data)
    return out
