#!/usr/bin/env python3
"""SF encode/decode of a block with a realistic long tail (a few bytes that occur 1..1000 times in 64 MiB: Lmax ~ 20-26):
the generic decode path.  usage: longtail_time.py [blocks=8] [least occurrences of a rare symbol=1]
(1: codes of 20-26 bits; 1500: codes of up to 15-16 bits, the 13..16-bit tables)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pkgload
pkg = pkgload.load()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mincnt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bs = 64 << 20
dev = torch.device("cuda", 0); pkg.lib().shafa_hip_init(0); st = torch.cuda.Stream(device=dev)
zt = pkg.zipf_table(1.2)
zt = np.where(zt >= 200, zt % 200, zt).astype(np.uint8)            # 200 common symbols
d_map = torch.from_numpy(zt).to(dev)
d_in = torch.empty(nb * bs, dtype=torch.uint8, device=dev)
pkg.gen_bytes(None, 99, 0, d_in, nb * bs, d_map)
torch.cuda.synchronize()
g = torch.Generator(device=dev); g.manual_seed(5)
for b in range(nb):
    for k, sym in enumerate(range(200, 256)):                        # 56 rare symbols: 1 .. ~3000 occurrences per block
        cnt = mincnt + (k * k * k) // 60
        pos = torch.randint(0, bs, (cnt,), device=dev, generator=g) + b * bs
        d_in[pos] = sym
bt = pkg.Batch(nb, bs)
off = [b * bs for b in range(nb)]; n = [bs] * nb
d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
bt.hist256(st, d_in, off, n, d_freq); bt.finish(st, nb)
freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
tabs = [pkg.sf_build_codes(freq[b]) for b in range(nb)]
tables = bt._tables(tabs)
lens = np.stack([t.lens() for t in tabs]).astype(np.uint64)
enc_bytes = [int(x) for x in ((freq * lens).sum(axis=1) + 7) // 8]
print("Lmax per block:", [int(l.max()) for l in lens][:4], "ratio %.4f" % (sum(enc_bytes) / (nb * bs)))
cap = ((max(enc_bytes) + 4096 + 255) // 256) * 256
eoff = [b * cap for b in range(nb)]
d_enc = torch.empty(nb * cap, dtype=torch.uint8, device=dev); d_enc_n = torch.zeros(nb, dtype=torch.int64, device=dev)
d_dec = torch.empty(nb * bs, dtype=torch.uint8, device=dev)
def timed(fn, reps=3):
    fn(); bt.finish(st, nb)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record(st)
    for _ in range(reps): fn()
    ev[1].record(st); bt.finish(st, nb)
    return ev[0].elapsed_time(ev[1]) / reps
te = timed(lambda: bt.sf_encode(st, d_in, off, n, tables, d_enc, eoff, [cap] * nb, d_enc_n))
assert [int(x) for x in d_enc_n.cpu().numpy()] == enc_bytes
td = timed(lambda: bt.sf_decode(st, d_enc, eoff, enc_bytes, tables, n, d_dec, off))
assert torch.equal(d_dec, d_in)
gib = nb * bs / 2**30
print(f"long tail, {nb} x 64 MiB: encode {gib / te * 1e3:.0f} GiB/s ({te / gib:.3f} ms/GiB), decode {gib / td * 1e3:.0f} GiB/s ({td / gib:.3f} ms/GiB), round trip identical")
