#!/usr/bin/env python3
"""How much of the one-pass encoder's time is LDS bank conflicts of its table look-ups?  Same code lengths (6 bits for
32 hot symbols, 9 for the rest; data uniform over the hot ones, output 0.75 n), three placements of the hot symbols in
the 256-entry table: 32 different bank slots (conflict-free), 4 slots x 8 entries (worst), random.
usage (GPU box): python tools/dbg/lut_conflict_probe.py [variant]"""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import pkgload

pkg = pkgload.load()
pkg.lib().shafa_hip_init(0)
if len(sys.argv) > 1:
    pass  # (the atomic-OR window form this switch selected was removed in round 4)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
nb, bs = 64, 64 << 20


def table_for(hot):
    t = pkg.CodeTable()
    hot = list(hot)
    cold = [s for s in range(256) if s not in set(hot)]
    for i, s in enumerate(hot):                   # 6-bit codes 0..31 (prefix 0)
        t.len[s] = 6
        t.bits[s][0] = (i << 2) & 0xFF
    for i, s in enumerate(cold):                  # 9-bit codes 1xxxxxxxx
        v = 256 + i
        t.len[s] = 9
        t.bits[s][0] = (v >> 1) & 0xFF
        t.bits[s][1] = (v & 1) << 7
    return t


cases = {"warm-up (random 32)": list(np.random.RandomState(1).permutation(256)[:32]),     # the first case also pays the clock ramp
         "32 slots (no conflicts)": list(range(32)),
         "32 slots, other values": [7 * 32 + ((5 * i) % 32) for i in range(32)],
         "4 slots x 8 (worst)": [32 * k + j for j in range(4) for k in range(8)],
         "random 32": list(np.random.RandomState(1).permutation(256)[:32]),
         "random 32 (b)": list(np.random.RandomState(2).permutation(256)[:32])}
for name, hot in cases.items():
    m = torch.tensor(hot, dtype=torch.uint8, device=dev)
    blk = m[torch.randint(0, 32, (bs,), device=dev)]
    d_in = blk.repeat(nb)
    t = table_for(hot)
    tabs = pkg.Batch._tables([t] * nb)
    cap = bs
    d_out = torch.empty(nb * cap, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(nb, dtype=torch.int64, device=dev)
    bt = pkg.Batch(nb, bs)
    off, n = [b * bs for b in range(nb)], [bs] * nb
    ooff, ocap = [b * cap for b in range(nb)], [cap] * nb
    torch.cuda.synchronize()
    for _ in range(2):
        bt.sf_encode(st, d_in, off, n, tabs, d_out, ooff, ocap, d_n)
    bt.finish(st, nb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5):
        bt.sf_encode(st, d_in, off, n, tabs, d_out, ooff, ocap, d_n)
    e1.record(st)
    bt.finish(st, nb)
    ms = e0.elapsed_time(e1) / 5
    alg = nb * bs * 1.75
    print(f"{name:28s} {ms:7.3f} ms per {nb} x 64 MiB   roofline {alg / ms / 1e6 / 8000:.3f}   out {int(d_n[0])}")
    bt.close()
    del d_in, d_out
