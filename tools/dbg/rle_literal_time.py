#!/usr/bin/env python3
"""K2 rle_encode / K5 rle_decode on data whose every byte is a literal (no zero, no run of four): the copy paths of
rle3_emit and rle_decode_kernel.  usage: tools/dbg/rle_literal_time.py [blocks]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
import pkgload
pkg = pkgload.load()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bs = 64 << 20
dev = torch.device("cuda", 0)
pkg.lib().shafa_hip_init(0)
st = torch.cuda.Stream(device=dev)
g = torch.Generator(device=dev); g.manual_seed(5)
unit = (torch.randint(1, 128, (bs,), generator=g, device=dev, dtype=torch.int32) * 2 - 1).to(torch.uint8)   # odd bytes
unit[1:] = torch.where(unit[1:] == unit[:-1], unit[1:] ^ 2, unit[1:])
d_in = unit.repeat(nb)
off, n = [b * bs for b in range(nb)], [bs] * nb
rcap = 2 * bs + 64
roff = [b * rcap for b in range(nb)]
d_rle = torch.empty(nb * rcap, dtype=torch.uint8, device=dev)
d_rle_n = torch.zeros(nb, dtype=torch.int64, device=dev)
d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
bt = pkg.Batch(nb, rcap)
def timed(fn, steps=3):
    torch.cuda.synchronize(); fn(); bt.finish(st, nb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps): fn()
    e1.record(st); bt.finish(st, nb)
    return e0.elapsed_time(e1) / steps
t_f = timed(lambda: bt.rle_encode(st, d_in, off, n, d_rle, roff, [rcap] * nb, d_rle_n, d_freq))
t_e = timed(lambda: bt.rle_encode(st, d_in, off, n, d_rle, roff, [rcap] * nb, d_rle_n, None))
rle_n = [int(x) for x in d_rle_n.cpu().numpy()]
assert rle_n == n
dcap = bs + 2048
doff = [b * dcap for b in range(nb)]
d_dec = torch.empty(nb * dcap, dtype=torch.uint8, device=dev)
d_dec_n = torch.zeros(nb, dtype=torch.int64, device=dev)
t_d = timed(lambda: bt.rle_decode(st, d_rle, roff, rle_n, d_dec, doff, [bs + 1024] * nb, d_dec_n))
assert torch.equal(d_dec[:bs], d_in[:bs])
tot = 2.0 * nb * bs
print(f"{nb} x 64 MiB of literals: rle_encode + histogram {t_f:.3f} ms ({tot / t_f / 1e6 / 8000:.3f} of peak), rle_encode alone {t_e:.3f} ms "
      f"({tot / t_e / 1e6 / 8000:.3f}), rle_decode {t_d:.3f} ms ({tot / t_d / 1e6 / 8000:.3f})")
