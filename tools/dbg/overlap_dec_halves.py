#!/usr/bin/env python3
"""Does the SF decoder gain from running its LDS-bound count pass (sfd_scan) of one half of a launch's blocks next to the
vector-ALU-bound symbol pass (sfd_wstage) of the other half?  The headline's 128 x 64 MiB decoded (a) in one call, (b) as two
calls of 64 blocks on two streams (two batch contexts), (c) as four calls of 32 on two streams.  Not what bench.py times."""
import os, sys, time
import numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(R, "tests"))
import torch
import pkgload
pkg = pkgload.load()
synth = pkgload.load_submodule("synth")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bs = 64 << 20
dev = torch.device("cuda", 0)
pkg.lib().shafa_hip_init(0)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
d_in = torch.empty(nb * bs, dtype=torch.uint8, device=dev)
d_map = torch.from_numpy(synth.zipf_mod256_table(1.2)).to(dev)
pkg.gen_bytes(sa, 20260101, 0, d_in, nb * bs, d_map)
off, n = [b * bs for b in range(nb)], [bs] * nb
d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
ba, bb = pkg.Batch(nb, bs), pkg.Batch(nb, bs)
torch.cuda.synchronize()
ba.hist256(sa, d_in, off, n, d_freq); ba.finish(sa, nb)
freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
tabs = [pkg.sf_build_codes(freq[b]) for b in range(nb)]
cap = ((bs + 4096 + 255) // 256) * 256
eoff = [b * cap for b in range(nb)]
d_enc = torch.empty(nb * cap, dtype=torch.uint8, device=dev)
d_n = torch.zeros(nb, dtype=torch.int64, device=dev)
ba.sf_encode(sa, d_in, off, n, tabs, d_enc, eoff, [cap] * nb, d_n); ba.finish(sa, nb)
enc_n = [int(x) for x in d_n.cpu().numpy()]
d_out = torch.empty(nb * bs, dtype=torch.uint8, device=dev)


def run(groups):
    """groups: list of (batch, stream, lo, hi)"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for bt, st, lo, hi in groups:
        bt.sf_decode(st, d_enc, eoff[lo:hi], enc_n[lo:hi], tabs[lo:hi], n[lo:hi], d_out, off[lo:hi])
    for bt, st, lo, hi in groups:
        bt.finish(st, hi - lo)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


h, q = nb // 2, nb // 4
shapes = {"one call": [(ba, sa, 0, nb)],
          "two halves, two streams": [(ba, sa, 0, h), (bb, sb, h, nb)],
          "two halves, one stream": [(ba, sa, 0, h), (ba, sa, h, nb)],
          "four quarters, two streams": [(ba, sa, 0, q), (bb, sb, q, 2 * q), (ba, sa, 2 * q, 3 * q), (bb, sb, 3 * q, nb)]}
for rep in range(3):
    for name, g in shapes.items():
        run(g)
        ms = min(run(g) for _ in range(3))
        print(f"{name:30s} {ms:7.3f} ms")
assert torch.equal(d_out, d_in)
