#!/usr/bin/env python3
"""host time of one shafa_hipd_sf_encode call (128 blocks) against its GPU time: Python wrapper vs the C launcher"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import pkgload
pkg = pkgload.load()
dev = torch.device("cuda", 0); pkg.lib().shafa_hip_init(0); st = torch.cuda.Stream(device=dev)
nb, bs = 128, int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 8 << 20
d_in = torch.randint(0, 256, (nb * bs,), dtype=torch.uint8, device=dev)
bt = pkg.Batch(nb, bs)
off = [b * bs for b in range(nb)]; n = [bs] * nb
d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
bt.hist256(st, d_in, off, n, d_freq); bt.finish(st, nb)
freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
tables = bt._tables([pkg.sf_build_codes(freq[b]) for b in range(nb)])
cap = bs + bs // 8
eoff = [b * cap for b in range(nb)]
d_enc = torch.empty(nb * cap, dtype=torch.uint8, device=dev); d_n = torch.zeros(nb, dtype=torch.int64, device=dev)
for _ in range(3):
    bt.sf_encode(st, d_in, off, n, tables, d_enc, eoff, [cap] * nb, d_n)
bt.finish(st, nb)
t0 = time.perf_counter()
for _ in range(20):
    bt.sf_encode(st, d_in, off, n, tables, d_enc, eoff, [cap] * nb, d_n)
t1 = time.perf_counter()
bt.finish(st, nb)
t2 = time.perf_counter()
print(f"python wrapper + C launcher: {(t1 - t0) / 20 * 1e3:.3f} ms per call (host, asynchronous); drain {(t2 - t1) * 1e3:.2f} ms")
io, il, oo, oc = (np.ascontiguousarray(x, dtype=np.uint64) for x in (off, n, eoff, [cap] * nb))
p64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint64))
L = pkg.lib(); sth = C.c_void_p(st.cuda_stream)
t0 = time.perf_counter()
for _ in range(20):
    L.shafa_hipd_sf_encode(bt.h, sth, nb, d_in.data_ptr(), p64(io), p64(il), tables, d_enc.data_ptr(), p64(oo), p64(oc), d_n.data_ptr())
t1 = time.perf_counter()
bt.finish(st, nb)
print(f"C launcher alone: {(t1 - t0) / 20 * 1e3:.3f} ms per call")
