#!/usr/bin/env python3
"""How much of SF decode (LDS-bound, a quarter of the HBM bandwidth) and SF encode (HBM-bound) overlap when they run at the
same time on two streams: decode of batch i next to encode of batch i + 1 (two batch contexts, two encoded buffers), against
the same work one after the other.  Headline data, 128 x 64 MiB.  Not what bench.py times: its step is encode THEN decode."""
import os
import sys
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
thb = pkg.tile_hist_bytes(bs)
toff = [b * thb for b in range(nb)]
d_th = torch.zeros(nb * thb, dtype=torch.uint8, device=dev)
d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
ba, bb = pkg.Batch(nb, bs), pkg.Batch(nb, bs)
torch.cuda.synchronize()
ba.hist256_tiles(sa, d_in, off, n, d_freq, d_th, toff)
ba.finish(sa, nb)
freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
tables = ba._tables([pkg.sf_build_codes(freq[b]) for b in range(nb)])
cap = ((bs + 4096 + 255) // 256) * 256
ooff = [b * cap for b in range(nb)]
enc = [torch.empty(nb * cap, dtype=torch.uint8, device=dev) for _ in range(2)]
d_n = [torch.zeros(nb, dtype=torch.int64, device=dev) for _ in range(2)]
d_dec = torch.empty(nb * bs, dtype=torch.uint8, device=dev)
for q in range(2):
    ba.sf_encode_tiles(sa, d_in, off, n, tables, d_th, toff, enc[q], ooff, [cap] * nb, d_n[q])
ba.finish(sa, nb)
enc_n = [int(x) for x in d_n[0].cpu().numpy()]

def encode(q): ba.sf_encode_tiles(sa, d_in, off, n, tables, d_th, toff, enc[q], ooff, [cap] * nb, d_n[q])
def decode(q): bb.sf_decode(sb, enc[q], ooff, enc_n, tables, n, d_dec, off)

import time
steps = 10
for mode in ("one after the other", "at the same time", "one after the other", "at the same time"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if mode == "one after the other":
            encode(i & 1); ba.finish(sa, nb); decode(i & 1); bb.finish(sb, nb)
        else:
            encode((i + 1) & 1); decode(i & 1); ba.finish(sa, nb); bb.finish(sb, nb)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{mode:20s}: {ms:7.3f} ms per encode + decode of {nb} x 64 MiB = {nb * bs / 2**30 / (ms / 1e3):7.1f} GiB/s")
assert torch.equal(d_dec, d_in)
