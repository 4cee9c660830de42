"""shafa-cd_amd — MI355X-native implementation of Shafa's block-codec hot path (Modules F, C, D).

This Python layer is plumbing only: it binds the C-ABI of ``libshafa_hip.so`` (include/shafa_hip.h)
with ctypes so that tests and bench.py can drive the HIP kernels with device memory owned by
PyTorch.  The product is the shared library (hand-written HIP for gfx950) and the C host in
``host/`` that mirrors the reference's module entry points.

The directory name contains a hyphen, so import it by path (tests/pkgload.py) — it registers
itself as ``shafa_cd_amd``.

There is NO CPU fallback: if the HIP library is missing, importing ``hip`` members raises.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.path.join(PKG_DIR, "libshafa_hip.so")
HOST_LIB_PATH = os.path.join(PKG_DIR, "libshafa_host.so")
CLI_PATH = os.path.join(PKG_DIR, "bin", "shafa")

# _modules_error values (reference utils/errors.h:5-16) + device error
SUCCESS, OUTSIDE_MODULE, LACK_OF_MEMORY, FILE_INACCESSIBLE, FILE_UNRECOGNIZABLE = 0, 1, 2, 3, 4
FILE_STREAM_FAILED, FILE_TOO_SMALL, THREAD_CREATION_FAILED, THREAD_TERMINATION_FAILED = 5, 6, 7, 8
DEVICE_ERROR = 9
RLE_DECODE_MAX = 67108864 + 1024

BLOCK_SIZES = {"K": 655360, "m": 8388608, "M": 67108864, None: 65536}   # shafa.c:97-104,304-305


class CodeTable(C.Structure):
    """Binary form of one .cod block: len[s] bits, bits[s] MSB-first (include/shafa_hip.h)."""
    _fields_ = [("len", C.c_uint8 * 256), ("bits", (C.c_uint8 * 32) * 256)]

    def lens(self):
        return np.ctypeslib.as_array(self.len).copy()

    @classmethod
    def from_strings(cls, codes):
        """codes: 256 strings of '0'/'1' (the fields of a .cod block)."""
        t = cls()
        assert len(codes) == 256
        for s, code in enumerate(codes):
            t.len[s] = len(code)
            for i, ch in enumerate(code):
                if ch == "1":
                    t.bits[s][i >> 3] |= 0x80 >> (i & 7)
        return t


class PipeResult(C.Structure):
    """shafa_pipe_result (include/shafa_hip.h, layer 3)."""
    _fields_ = [("out", C.c_void_p), ("out_n", C.c_size_t), ("mid_n", C.c_size_t),
                ("freq", C.c_uint64 * 256), ("freq_in", C.c_uint64 * 256)]


class PipeBlock(C.Structure):
    """shafa_pipe_block (include/shafa_hip.h, layer 3 groups)."""
    _fields_ = [("in_off", C.c_size_t), ("in_n", C.c_size_t), ("table", C.POINTER(CodeTable)), ("n_symbols", C.c_size_t),
                ("out_cap", C.c_size_t)]


PIPE_GROUP_MAX = 256
OP_HIST, OP_RLE_ENCODE, OP_SF_ENCODE, OP_SF_DECODE, OP_RLE_DECODE, OP_SF_RLE_DECODE, OP_FTC = 1, 2, 3, 4, 5, 6, 7
PIPE_INPUT_HIST, PIPE_FTC_RLE, PIPE_FTC_PLAIN = 1, 2, 4


class ShafaError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__(f"shafa_hip error {code} {what}")


_lib = None


def lib():
    """Load libshafa_hip.so (built by __graft_entry__.build()).  Fails loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    u8p, u64p, szp, vp = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_size_t), C.c_void_p
    tp = C.POINTER(CodeTable)
    L.shafa_hip_abi_version.restype = C.c_int
    L.shafa_hip_device_count.restype = C.c_int
    L.shafa_hip_init.argtypes = [C.c_int]
    L.shafa_hip_last_error.restype = C.c_char_p
    L.shafa_hip_set_option.argtypes = [C.c_char_p, C.c_long]
    L.shafa_hip_set_option.restype = C.c_int
    L.shafa_hip_hist256.argtypes = [u8p, C.c_size_t, u64p]
    L.shafa_hip_rle_encode.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, szp, u64p]
    L.shafa_hip_sf_encode.argtypes = [u8p, C.c_size_t, tp, u8p, C.c_size_t, szp]
    L.shafa_hip_sf_decode.argtypes = [u8p, C.c_size_t, tp, u8p, C.c_size_t]
    L.shafa_hip_rle_decode.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, szp]
    L.shafa_hipd_batch_create.argtypes = [C.c_int, C.c_size_t, C.POINTER(vp)]
    L.shafa_hipd_batch_destroy.argtypes = [vp]
    L.shafa_hipd_batch_destroy.restype = None
    L.shafa_hipd_hist256.argtypes = [vp, vp, C.c_int, u8p, u64p, u64p, vp]
    L.shafa_hipd_rle_encode.argtypes = [vp, vp, C.c_int, u8p, u64p, u64p, u8p, u64p, u64p, vp, vp]
    L.shafa_hipd_sf_encode.argtypes = [vp, vp, C.c_int, u8p, u64p, u64p, tp, u8p, u64p, u64p, vp]
    L.shafa_hipd_sf_decode.argtypes = [vp, vp, C.c_int, u8p, u64p, u64p, tp, u64p, u8p, u64p]
    L.shafa_hipd_rle_decode.argtypes = [vp, vp, C.c_int, u8p, u64p, u64p, u8p, u64p, u64p, vp]
    L.shafa_hipd_sf_build_codes.argtypes = [vp, vp, C.c_int, vp, vp]
    L.shafa_hip_tile_hist_bytes.argtypes = [C.c_size_t]
    L.shafa_hip_tile_hist_bytes.restype = C.c_size_t
    L.shafa_hipd_hist256_tiles.argtypes = [vp, vp, C.c_int, u8p, u64p, u64p, vp, u8p, u64p]
    L.shafa_hipd_rle_encode_tiles.argtypes = [vp, vp, C.c_int, u8p, u64p, u64p, u8p, u64p, u64p, vp, vp, u8p, u64p]
    L.shafa_hipd_sf_encode_tiles.argtypes = [vp, vp, C.c_int, u8p, u64p, u64p, tp, u8p, u64p, u8p, u64p, u64p, vp]
    L.shafa_hipd_finish.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.shafa_hipd_gen_bytes.argtypes = [vp, C.c_uint64, C.c_uint64, u8p, u8p, C.c_size_t]
    L.shafa_pipe_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.shafa_pipe_destroy.argtypes = [vp]
    L.shafa_pipe_destroy.restype = None
    L.shafa_pipe_slots.argtypes = [vp]
    L.shafa_pipe_slot_device.argtypes = [vp, C.c_int]
    L.shafa_pipe_slot_device.restype = C.c_int
    L.shafa_hip_init_devices.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.shafa_hip_init_devices.restype = C.c_int
    L.shafa_hip_devices.restype = C.c_int
    L.shafa_pipe_in.argtypes = [vp, C.c_int, C.c_size_t]
    L.shafa_pipe_in.restype = C.c_void_p
    L.shafa_pipe_submit.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, tp, C.c_size_t, C.c_size_t, C.c_int]
    L.shafa_pipe_wait.argtypes = [vp, C.c_int, C.POINTER(PipeResult)]
    L.shafa_pipe_submit_group.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(PipeBlock), C.c_int]
    L.shafa_pipe_wait_group.argtypes = [vp, C.c_int, C.c_int, C.POINTER(PipeResult), C.POINTER(C.c_int)]
    L.shafa_pipe_ftc_encode.argtypes = [vp, C.c_int, C.c_int, tp, C.c_size_t]
    for name in ("shafa_pipe_create", "shafa_pipe_slots", "shafa_pipe_submit", "shafa_pipe_wait", "shafa_pipe_submit_group",
                 "shafa_pipe_wait_group", "shafa_pipe_ftc_encode"):
        getattr(L, name).restype = C.c_int
    for name in ("shafa_hip_init", "shafa_hip_hist256", "shafa_hip_rle_encode", "shafa_hip_sf_encode",
                 "shafa_hip_sf_decode", "shafa_hip_rle_decode", "shafa_hipd_batch_create",
                 "shafa_hipd_hist256", "shafa_hipd_rle_encode", "shafa_hipd_sf_encode",
                 "shafa_hipd_sf_decode", "shafa_hipd_rle_decode", "shafa_hipd_finish",
                 "shafa_hipd_gen_bytes", "shafa_hipd_hist256_tiles", "shafa_hipd_rle_encode_tiles",
                 "shafa_hipd_sf_encode_tiles"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


def init_devices(devices=None):
    """Select the GPUs of the layer-3 pipeline (None: every visible device); include/shafa_hip.h: shafa_hip_init_devices."""
    if devices:
        arr = (C.c_int * len(devices))(*devices)
        _check(lib().shafa_hip_init_devices(arr, len(devices)), "init_devices")
    else:
        _check(lib().shafa_hip_init_devices(None, 0), "init_devices")
    return lib().shafa_hip_devices()


def set_option(name, value):
    """Tuning knobs of the library (include/shafa_hip.h: shafa_hip_set_option)."""
    _check(lib().shafa_hip_set_option(name.encode(), int(value)), "set_option " + name)


def _check(rc, what=""):
    if rc != SUCCESS:
        msg = lib().shafa_hip_last_error().decode() if rc == DEVICE_ERROR else ""
        raise ShafaError(rc, f"{what} {msg}")


def _np_u8(data):
    a = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else data
    return np.ascontiguousarray(a, dtype=np.uint8)


# ------------------------------------------------------------------ layer 1: host buffers, one block
def hist256(data):
    a = _np_u8(data)
    f = np.zeros(256, dtype=np.uint64)
    _check(lib().shafa_hip_hist256(a.ctypes.data, a.size, f.ctypes.data_as(C.POINTER(C.c_uint64))), "hist256")
    return f


def rle_encode(data, want_freq=False):
    a = _np_u8(data)
    out = np.empty(2 * a.size + 3, dtype=np.uint8)
    n = C.c_size_t(0)
    f = np.zeros(256, dtype=np.uint64) if want_freq else None
    fp = f.ctypes.data_as(C.POINTER(C.c_uint64)) if want_freq else None
    _check(lib().shafa_hip_rle_encode(a.ctypes.data, a.size, out.ctypes.data, out.size, C.byref(n), fp), "rle_encode")
    return (out[:n.value].copy(), f) if want_freq else out[:n.value].copy()


def sf_encode(data, table, cap=None, raw_rc=False):
    a = _np_u8(data)
    cap = cap if cap is not None else a.size * 32 + 16
    out = np.empty(max(cap, 1), dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().shafa_hip_sf_encode(a.ctypes.data, a.size, C.byref(table), out.ctypes.data, cap, C.byref(n))
    if raw_rc:
        return rc, out[:n.value].copy()
    _check(rc, "sf_encode")
    return out[:n.value].copy()


def sf_decode(data, table, n_symbols, raw_rc=False):
    a = _np_u8(data)
    out = np.empty(max(n_symbols, 1), dtype=np.uint8)
    rc = lib().shafa_hip_sf_decode(a.ctypes.data, a.size, C.byref(table), out.ctypes.data, n_symbols)
    if raw_rc:
        return rc, out[:n_symbols].copy()
    _check(rc, "sf_decode")
    return out[:n_symbols].copy()


def rle_decode(data, cap=None, raw_rc=False):
    a = _np_u8(data)
    cap = cap if cap is not None else min(a.size * 255 + 16, RLE_DECODE_MAX)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().shafa_hip_rle_decode(a.ctypes.data, a.size, out.ctypes.data, cap, C.byref(n))
    if raw_rc:
        return rc, out[:n.value].copy()
    _check(rc, "rle_decode")
    return out[:n.value].copy()


# ------------------------------------------------------------------ layer 2: device buffers, batches
def _u64arr(v):
    return np.ascontiguousarray(v, dtype=np.uint64)


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


class Batch:
    """Reusable batch context (shafa_hipd_batch).  Tensors are torch uint8/int64 CUDA tensors; the
    stream is a torch.cuda.Stream (or None for the library's own stream)."""

    def __init__(self, max_blocks, max_block_bytes):
        self.h = C.c_void_p()
        self.max_blocks = max_blocks
        _check(lib().shafa_hipd_batch_create(max_blocks, max_block_bytes, C.byref(self.h)), "batch_create")

    def close(self):
        if self.h:
            lib().shafa_hipd_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _st(stream):
        """hipStream_t of a torch stream.  The tensors handed to a launch were typically produced (allocated,
        filled, copied) by torch on ITS current stream: order this launch after that work."""
        if stream is None:
            return None
        import torch
        cur = torch.cuda.current_stream(stream.device)
        if cur.cuda_stream != stream.cuda_stream:
            stream.wait_stream(cur)
        return C.c_void_p(stream.cuda_stream)

    @staticmethod
    def _tables(tables):
        arr = (CodeTable * len(tables))()
        for i, t in enumerate(tables):
            C.memmove(C.byref(arr[i]), C.byref(t), C.sizeof(CodeTable))
        return arr

    # Every *_off / *_n / *_cap argument is a host sequence with one entry per block.
    def hist256(self, stream, d_in, in_off, in_n, d_freq):
        io, il = _u64arr(in_off), _u64arr(in_n)
        _check(lib().shafa_hipd_hist256(self.h, self._st(stream), len(io), d_in.data_ptr(), _p64(io), _p64(il),
                                        d_freq.data_ptr()), "hipd_hist256")

    def sf_build_codes(self, stream, nblocks, d_freq, d_tables):
        """Module T on the device: d_freq nblocks x 256 u64 -> d_tables nblocks x sizeof(CodeTable) bytes (device memory)."""
        _check(lib().shafa_hipd_sf_build_codes(self.h, self._st(stream), nblocks, d_freq.data_ptr(), d_tables.data_ptr()),
               "hipd_sf_build_codes")

    def rle_encode(self, stream, d_in, in_off, in_n, d_out, out_off, out_cap, d_out_n, d_freq=None):
        io, il, oo, oc = _u64arr(in_off), _u64arr(in_n), _u64arr(out_off), _u64arr(out_cap)
        _check(lib().shafa_hipd_rle_encode(self.h, self._st(stream), len(io), d_in.data_ptr(), _p64(io), _p64(il),
                                           d_out.data_ptr(), _p64(oo), _p64(oc), d_out_n.data_ptr(),
                                           d_freq.data_ptr() if d_freq is not None else None), "hipd_rle_encode")

    def sf_encode(self, stream, d_in, in_off, in_n, tables, d_out, out_off, out_cap, d_out_n):
        io, il, oo, oc = _u64arr(in_off), _u64arr(in_n), _u64arr(out_off), _u64arr(out_cap)
        tarr = tables if isinstance(tables, C.Array) else self._tables(tables)
        _check(lib().shafa_hipd_sf_encode(self.h, self._st(stream), len(io), d_in.data_ptr(), _p64(io), _p64(il),
                                          tarr, d_out.data_ptr(), _p64(oo), _p64(oc), d_out_n.data_ptr()),
               "hipd_sf_encode")

    # ---- with tile histograms (include/shafa_hip.h: "Tile histograms"): block b's at d_thist + thist_off[b] ----
    def hist256_tiles(self, stream, d_in, in_off, in_n, d_freq, d_thist, thist_off):
        io, il, to = _u64arr(in_off), _u64arr(in_n), _u64arr(thist_off)
        _check(lib().shafa_hipd_hist256_tiles(self.h, self._st(stream), len(io), d_in.data_ptr(), _p64(io), _p64(il),
                                              d_freq.data_ptr(), d_thist.data_ptr(), _p64(to)), "hipd_hist256_tiles")

    def rle_encode_tiles(self, stream, d_in, in_off, in_n, d_out, out_off, out_cap, d_out_n, d_freq, d_thist, thist_off):
        io, il, oo, oc, to = _u64arr(in_off), _u64arr(in_n), _u64arr(out_off), _u64arr(out_cap), _u64arr(thist_off)
        _check(lib().shafa_hipd_rle_encode_tiles(self.h, self._st(stream), len(io), d_in.data_ptr(), _p64(io), _p64(il),
                                                 d_out.data_ptr(), _p64(oo), _p64(oc), d_out_n.data_ptr(), d_freq.data_ptr(),
                                                 d_thist.data_ptr(), _p64(to)), "hipd_rle_encode_tiles")

    def sf_encode_tiles(self, stream, d_in, in_off, in_n, tables, d_thist, thist_off, d_out, out_off, out_cap, d_out_n):
        io, il, oo, oc, to = _u64arr(in_off), _u64arr(in_n), _u64arr(out_off), _u64arr(out_cap), _u64arr(thist_off)
        tarr = tables if isinstance(tables, C.Array) else self._tables(tables)
        _check(lib().shafa_hipd_sf_encode_tiles(self.h, self._st(stream), len(io), d_in.data_ptr(), _p64(io), _p64(il),
                                                tarr, d_thist.data_ptr(), _p64(to), d_out.data_ptr(), _p64(oo), _p64(oc),
                                                d_out_n.data_ptr()), "hipd_sf_encode_tiles")

    def sf_decode(self, stream, d_in, in_off, in_n, tables, n_symbols, d_out, out_off):
        io, il, oo, ns = _u64arr(in_off), _u64arr(in_n), _u64arr(out_off), _u64arr(n_symbols)
        tarr = tables if isinstance(tables, C.Array) else self._tables(tables)
        _check(lib().shafa_hipd_sf_decode(self.h, self._st(stream), len(io), d_in.data_ptr(), _p64(io), _p64(il),
                                          tarr, _p64(ns), d_out.data_ptr(), _p64(oo)), "hipd_sf_decode")

    def rle_decode(self, stream, d_in, in_off, in_n, d_out, out_off, out_cap, d_out_n):
        io, il, oo, oc = _u64arr(in_off), _u64arr(in_n), _u64arr(out_off), _u64arr(out_cap)
        _check(lib().shafa_hipd_rle_decode(self.h, self._st(stream), len(io), d_in.data_ptr(), _p64(io), _p64(il),
                                           d_out.data_ptr(), _p64(oo), _p64(oc), d_out_n.data_ptr()), "hipd_rle_decode")

    def finish(self, stream, nblocks, raise_on_error=True):
        errs = (C.c_int * max(nblocks, 1))()
        rc = lib().shafa_hipd_finish(self.h, self._st(stream), nblocks, errs)
        if raise_on_error:
            _check(rc, "hipd_finish")
        return rc, list(errs)[:nblocks]


class Pipe:
    """Layer 3: bounded in-order block pipeline over host buffers (shafa_pipe_*)."""

    def __init__(self, n_slots):
        self._h = C.c_void_p()
        _check(lib().shafa_pipe_create(n_slots, C.byref(self._h)), "pipe_create")
        self.n_slots = lib().shafa_pipe_slots(self._h)
        self.devices = [lib().shafa_pipe_slot_device(self._h, i) for i in range(self.n_slots)]

    def close(self):
        if self._h:
            lib().shafa_pipe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, slot, op, data, table=None, n_symbols=0, out_cap=0, flags=0):
        a = _np_u8(data)
        p = lib().shafa_pipe_in(self._h, slot, a.size)
        if not p:
            raise ShafaError(LACK_OF_MEMORY, "pipe_in (slot busy?)")
        C.memmove(p, a.ctypes.data, a.size)
        _check(lib().shafa_pipe_submit(self._h, slot, op, a.size, C.byref(table) if table is not None else None,
                                       n_symbols, out_cap, flags), "pipe_submit")

    def wait(self, slot, raw_rc=False):
        """-> (rc, result bytes, PipeResult)"""
        r = PipeResult()
        rc = lib().shafa_pipe_wait(self._h, slot, C.byref(r))
        if rc and not raw_rc:
            _check(rc, "pipe_wait")
        out = C.string_at(r.out, r.out_n) if rc == 0 and r.out_n else b""
        return rc, out, r


    def ftc_encode(self, slot, use_rle, table, out_cap):
        """stage two of SHAFA_OP_FTC: Module C from the bytes stage one left on the device (shafa_pipe_ftc_encode)"""
        _check(lib().shafa_pipe_ftc_encode(self._h, slot, 1 if use_rle else 0, C.byref(table), out_cap), "pipe_ftc_encode")

    def submit_group(self, slot, op, datas, tables=None, n_symbols=None, out_caps=None, flags=0):
        """several blocks in one slot: inputs laid out at 16-byte aligned offsets (shafa_pipe_submit_group)"""
        arrs = [_np_u8(d) for d in datas]
        offs, pos = [], 0
        for a in arrs:
            offs.append(pos)
            pos += (a.size + 15) // 16 * 16
        p = lib().shafa_pipe_in(self._h, slot, max(pos, 16))
        if not p:
            raise ShafaError(LACK_OF_MEMORY, "pipe_in (slot busy?)")
        blocks = (PipeBlock * len(arrs))()
        self._keep = tables                                   # the tables must outlive the call
        for i, a in enumerate(arrs):
            if a.size:
                C.memmove(p + offs[i], a.ctypes.data, a.size)
            blocks[i].in_off, blocks[i].in_n = offs[i], a.size
            blocks[i].table = C.pointer(tables[i]) if tables is not None else None
            blocks[i].n_symbols = n_symbols[i] if n_symbols is not None else 0
            blocks[i].out_cap = out_caps[i] if out_caps is not None else 0
        _check(lib().shafa_pipe_submit_group(self._h, slot, op, len(arrs), blocks, flags), "pipe_submit_group")
        return len(arrs)

    def wait_group(self, slot, n):
        """-> (rc of the call, [block rc], [result bytes], [PipeResult])"""
        res = (PipeResult * n)()
        brc = (C.c_int * n)()
        rc = lib().shafa_pipe_wait_group(self._h, slot, n, res, brc)
        outs = [C.string_at(res[i].out, res[i].out_n) if rc == 0 and brc[i] == 0 and res[i].out_n else b"" for i in range(n)]
        return rc, list(brc), outs, res


TILE_BYTES = 32768            # SHAFA_TILE_BYTES: the tile of the tile histograms


def tile_hist_bytes(n):
    """bytes of the tile histograms of a block of n bytes (256 x u16 per 32 KiB tile)."""
    return int(lib().shafa_hip_tile_hist_bytes(int(n)))


def gen_bytes(stream, seed, first_index, d_out, n, d_map=None):
    st = C.c_void_p(stream.cuda_stream) if stream is not None else None
    _check(lib().shafa_hipd_gen_bytes(st, seed, first_index, d_map.data_ptr() if d_map is not None else None,
                                      d_out.data_ptr(), n), "hipd_gen_bytes")


def zipf_table(s=1.2, nsym=256):
    """2^16-entry inverse CDF of Zipf(s) over nsym symbols: the byte map of the synthetic streams
    (SURVEY.md §8(d) config 4).  Same construction as tests/golden/make_golden.py."""
    w = np.arange(1, nsym + 1, dtype=np.float64) ** (-s)
    cdf = np.cumsum(w) / np.sum(w)
    edges = np.minimum(np.floor(cdf * 65536.0 + 0.5).astype(np.int64), 65536)
    edges[-1] = 65536
    table = np.zeros(65536, dtype=np.uint8)
    lo = 0
    for k in range(nsym):
        table[lo:edges[k]] = k
        lo = max(lo, edges[k])
    return table


# ------------------------------------------------------------------ C host library (formats, Module T, drivers)
_host = None


def host():
    """Load libshafa_host.so: the C host's formats, Module T and module drivers (host/shafa_host.h)."""
    global _host
    if _host is not None:
        return _host
    lib()   # libshafa_host.so links against libshafa_hip.so
    path = os.environ.get("SHAFA_HOST_LIB") or HOST_LIB_PATH        # (the sanitizer build of tools/san/run_san.sh)
    if not os.path.exists(path):
        raise ImportError(f"{path} not built: run __graft_entry__.build()")
    H = C.CDLL(path)
    u64p, tp = C.POINTER(C.c_uint64), C.POINTER(CodeTable)
    H.shafa_sf_build_codes.argtypes = [u64p, tp]
    H.shafa_sf_build_codes.restype = None
    H.shafa_sf_build_codes_batch.argtypes = [u64p, C.c_int, tp]
    H.shafa_sf_build_codes_batch.restype = None
    H.shafa_freq_format.argtypes = [u64p, C.c_char_p]
    H.shafa_freq_format.restype = C.c_size_t
    H.shafa_freq_parse.argtypes = [C.c_char_p, u64p]
    H.shafa_cod_format.argtypes = [tp, C.c_char_p]
    H.shafa_cod_format.restype = C.c_size_t
    H.shafa_cod_parse.argtypes = [C.c_char_p, tp]
    H.shafa_rle_worthwhile.argtypes = [C.c_uint64, C.c_uint64, C.c_bool]
    H.shafa_rle_worthwhile.restype = C.c_bool
    H.shafa_block_count.argtypes = [C.c_uint64, u64p, u64p]
    H.shafa_block_count.restype = C.c_uint64
    _host = H
    return H


def sf_build_codes(freq):
    """Module T core (reference t.c:74-210) on one 256-bin histogram -> CodeTable."""
    f = np.ascontiguousarray(freq, dtype=np.uint64)
    t = CodeTable()
    host().shafa_sf_build_codes(f.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(t))
    return t


def sf_build_codes_batch(freq):
    """Module T on n histograms (array n x 256) -> ctypes array of n CodeTable (what Batch.sf_encode / sf_decode take)."""
    f = np.ascontiguousarray(freq, dtype=np.uint64).reshape(-1, 256)
    arr = (CodeTable * f.shape[0])()
    host().shafa_sf_build_codes_batch(f.ctypes.data_as(C.POINTER(C.c_uint64)), f.shape[0], arr)
    return arr


def freq_format(freq):
    f = np.ascontiguousarray(freq, dtype=np.uint64)
    buf = C.create_string_buffer(256 * 21 + 1)
    n = host().shafa_freq_format(f.ctypes.data_as(C.POINTER(C.c_uint64)), buf)
    return buf.raw[:n]


def freq_parse(text):
    f = np.zeros(256, dtype=np.uint64)
    rc = host().shafa_freq_parse(text, f.ctypes.data_as(C.POINTER(C.c_uint64)))
    return rc, f


def cod_format(table):
    buf = C.create_string_buffer(33152 + 8)
    n = host().shafa_cod_format(C.byref(table), buf)
    return buf.raw[:n]


def cod_parse(text):
    t = CodeTable()
    rc = host().shafa_cod_parse(text, C.byref(t))
    return rc, t
