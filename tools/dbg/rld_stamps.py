#!/usr/bin/env python3
"""Where a tile's time goes in rle_decode_kernel (build with -DRLD_STAMPS first: tools/dbg/mkvar.sh rle_decode.hip
STAMPS:-DRLD_STAMPS; cp _ab/STAMPS.so shafa-cd_amd/libshafa_hip.so).  One K2 + K5 pass over bench.py's pipeline blocks
(`runs` or `mixed`), then the eight phase stamps of the first 65536 workgroups (100 MHz ticks).
usage: tools/dbg/rld_stamps.py [runs|mixed] [blocks]"""
import ctypes as C
import os
import sys
import numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, R)
import torch
import pkgload
import bench
pkg = pkgload.load()
synth = pkgload.load_submodule("synth")
kind = sys.argv[1] if len(sys.argv) > 1 else "runs"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
bs = 64 << 20
dev = torch.device("cuda", 0)
pkg.lib().shafa_hip_init(0)
st = torch.cuda.Stream(device=dev)
d_in = bench.pipeline_blocks(kind, synth, torch, dev, bs, nb)
off, n = [b * bs for b in range(nb)], [bs] * nb
rcap = 2 * bs + 64
roff = [b * rcap for b in range(nb)]
d_rle = torch.empty(nb * rcap, dtype=torch.uint8, device=dev)
d_rle_n = torch.zeros(nb, dtype=torch.int64, device=dev)
bt = pkg.Batch(nb, rcap)
torch.cuda.synchronize()
bt.rle_encode(st, d_in, off, n, d_rle, roff, [rcap] * nb, d_rle_n, None)
bt.finish(st, nb)
rle_n = [int(x) for x in d_rle_n.cpu().numpy()]
dcap = bs + 2048
doff = [b * dcap for b in range(nb)]
d_dec = torch.empty(nb * dcap, dtype=torch.uint8, device=dev)
d_dec_n = torch.zeros(nb, dtype=torch.int64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    e0.record(st)
    bt.rle_decode(st, d_rle, roff, rle_n, d_dec, doff, [bs + 1024] * nb, d_dec_n)
    e1.record(st)
    bt.finish(st, nb)
print(f"{kind}: {nb} x 64 MiB, rle_n/n {sum(rle_n) / (nb * bs):.3f}; rle_decode {e0.elapsed_time(e1):.3f} ms (stamped build)")
if len(sys.argv) > 3 and sys.argv[3] == "nostamps":      # tools/dbg/rld_abl.sh: the time only
    sys.exit(0)
N = 65536
buf = (C.c_ulonglong * (N * 8))()
L = pkg.lib()
L.shafa_rld_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.shafa_rld_read_stamps(buf, N * 8) == 0
a = np.ctypeslib.as_array(buf).reshape(N, 8).astype(np.int64)
a = a[(a[:, 7] > 0) & (a[:, 0] > 0)]
a = a[len(a) // 4:]                      # past the ramp-up of the grid
names = ["ticket + table -> barrier", "tile loaded (vmcnt 0)", "lane maps -> barrier", "state look-back -> barrier",
         "lengths, scan -> barrier", "image built, O arrived", "stores issued and retired"]
print(f"{len(a)} workgroups; whole tile mean {(a[:, 7] - a[:, 0]).mean() / 100:.2f} us  p10 {np.percentile(a[:, 7] - a[:, 0], 10) / 100:.2f}  p90 {np.percentile(a[:, 7] - a[:, 0], 90) / 100:.2f}")
for i, nm in enumerate(names):
    d = a[:, i + 1] - a[:, i]
    print(f"  {nm:34s} mean {d.mean() / 100:6.2f} us   p10 {np.percentile(d, 10) / 100:6.2f}   p50 {np.percentile(d, 50) / 100:6.2f}   p90 {np.percentile(d, 90) / 100:6.2f}")
span = (a[:, 7].max() - a[:, 0].min()) / 100
print(f"span of these workgroups {span:.1f} us -> {len(a) / span:.1f} tiles per us")
