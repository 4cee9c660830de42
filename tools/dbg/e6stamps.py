#!/usr/bin/env python3
"""Phase times and real residency of sfe6_kernel's workgroups (build with -DE6_STAMPS first: tools/dbg/mkvar.sh
sf_encode6.hip STAMPS:-DE6_STAMPS; cp _ab/STAMPS.so shafa-cd_amd/libshafa_hip.so).  One encode of <blocks> x 64 MiB, then
the stamps of the first 65536 workgroups: start, inputs arrived, end (100 MHz ticks), XCC / CU they ran on."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
import pkgload
pkg = pkgload.load()
synth = pkgload.load_submodule("synth")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bs = 64 << 20
dev = torch.device("cuda", 0)
pkg.lib().shafa_hip_init(0)
st = torch.cuda.Stream(device=dev)
d_in = torch.empty(nb * bs, dtype=torch.uint8, device=dev)
d_map = torch.from_numpy(synth.zipf_mod256_table(1.2)).to(dev)
pkg.gen_bytes(st, 20260101, 0, d_in, nb * bs, d_map)
off, n = [b * bs for b in range(nb)], [bs] * nb
thb = pkg.tile_hist_bytes(bs)
toff = [b * thb for b in range(nb)]
d_th = torch.zeros(nb * thb, dtype=torch.uint8, device=dev)
d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
bt = pkg.Batch(nb, bs)
torch.cuda.synchronize()
bt.hist256_tiles(st, d_in, off, n, d_freq, d_th, toff)
bt.finish(st, nb)
freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
tables = bt._tables([pkg.sf_build_codes(freq[b]) for b in range(nb)])
cap = ((bs + 4096 + 255) // 256) * 256
ooff = [b * cap for b in range(nb)]
d_out = torch.empty(nb * cap, dtype=torch.uint8, device=dev)
d_n = torch.zeros(nb, dtype=torch.int64, device=dev)
for _ in range(3):
    bt.sf_encode_tiles(st, d_in, off, n, tables, d_th, toff, d_out, ooff, [cap] * nb, d_n)
bt.finish(st, nb)
N = 65536
buf = (C.c_ulonglong * (N * 4))()
L = pkg.lib()
L.shafa_e6_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.shafa_e6_read_stamps(buf, N * 4) == 0
a = np.ctypeslib.as_array(buf).reshape(N, 4).astype(np.int64)
t0, t1, t2 = a[:, 0], a[:, 1], a[:, 2]
ok = t2 > 0
a, t0, t1, t2 = a[ok], t0[ok], t1[ok], t2[ok]
hw = a[:, 3]
xcc = (hw >> 32) & 0xF
hwid = hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF
se = (hwid >> 13) & 0x7
key = xcc * 256 + se * 16 + cu
print(f"{len(t0)} workgroups stamped; span {(t2.max() - t0.min()) / 100:.1f} us")
print(f"load phase  (start -> inputs arrived): mean {(t1 - t0).mean() / 100:.2f} us  p10 {np.percentile(t1 - t0, 10) / 100:.2f}  p90 {np.percentile(t1 - t0, 90) / 100:.2f}")
print(f"work phase  (inputs -> end):           mean {(t2 - t1).mean() / 100:.2f} us  p10 {np.percentile(t2 - t1, 10) / 100:.2f}  p90 {np.percentile(t2 - t1, 90) / 100:.2f}")
print(f"distinct (xcc, se, cu): {len(np.unique(key))}")
# residency on one CU over time: workgroups alive, sampled
for k in np.unique(key)[:4]:
    m = key == k
    s, e = t0[m], t2[m]
    ts = np.linspace(s.min(), e.max(), 400)
    alive = [(int(((s <= x) & (e > x)).sum())) for x in ts]
    loading = [(int(((s <= x) & (t1[m] > x)).sum())) for x in ts]
    # gaps between one workgroup's end and the next start on the same CU
    print(f"cu {int(k):5d}: {m.sum()} WGs, alive mean {np.mean(alive):.2f} max {max(alive)}, in load phase mean {np.mean(loading):.2f}, "
          f"throughput {m.sum() / ((e.max() - s.min()) / 100):.3f} WG/us")
