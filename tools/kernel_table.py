#!/usr/bin/env python3
"""Every kernel family of the path at -b M (64 MiB blocks), timed with HIP events on the launch stream, against its
algorithmic HBM bytes.  Data: Zipf symbols in geometric runs (so RLE has work); SF runs on the RLE
bytes, like the reference's F -> T -> C -> D pipeline.  usage: kernel_table.py [blocks=16]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pkgload  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as mg  # noqa: E402

PEAK = 8000.0
pkg = pkgload.load()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bs = 64 << 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
pkg.lib().shafa_hip_init(0)
st = torch.cuda.Stream(device=dev)
zt = pkg.zipf_table(1.2)
blk = torch.from_numpy(mg.runs_stream(11, bs, zt)).to(dev)
d_in = blk.repeat(nb)
bt = pkg.Batch(nb, 2 * bs + 64)
off = [b * bs for b in range(nb)]
n = [bs] * nb


def timed(fn, reps=5):
    fn(); bt.finish(st, nb)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record(st)
    for _ in range(reps):
        fn()
    ev[1].record(st)
    bt.finish(st, nb)
    return ev[0].elapsed_time(ev[1]) / reps * 1e-3


rows = []
d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
t = timed(lambda: bt.hist256(st, d_in, off, n, d_freq))
rows.append(("hist256 (make_freq)", nb * bs, t))

rcap = 2 * bs + 64
roff = [b * rcap for b in range(nb)]
d_rle = torch.empty(nb * rcap, dtype=torch.uint8, device=dev)
d_rle_n = torch.zeros(nb, dtype=torch.int64, device=dev)
t = timed(lambda: bt.rle_encode(st, d_in, off, n, d_rle, roff, [rcap] * nb, d_rle_n, d_freq))
rle_n = [int(x) for x in d_rle_n.cpu().numpy()]
rows.append(("rle_encode + fused hist (block_compression)", nb * bs + sum(rle_n), t))

freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
tables = bt._tables([pkg.sf_build_codes(freq[b]) for b in range(nb)])
lens = np.stack([tables[b].lens() for b in range(nb)]).astype(np.uint64)
enc_bytes = [int(x) for x in ((freq * lens).sum(axis=1) + 7) // 8]
cap = ((max(enc_bytes) + 4096 + 255) // 256) * 256
eoff = [b * cap for b in range(nb)]
d_enc = torch.empty(nb * cap, dtype=torch.uint8, device=dev)
d_enc_n = torch.zeros(nb, dtype=torch.int64, device=dev)
t = timed(lambda: bt.sf_encode(st, d_rle, roff, rle_n, tables, d_enc, eoff, [cap] * nb, d_enc_n))
assert [int(x) for x in d_enc_n.cpu().numpy()] == enc_bytes
rows.append(("sf_encode (binary_coding), on the RLE bytes", sum(rle_n) + sum(enc_bytes), t))

d_sym = torch.empty(nb * rcap, dtype=torch.uint8, device=dev)
t = timed(lambda: bt.sf_decode(st, d_enc, eoff, enc_bytes, tables, rle_n, d_sym, roff))
rows.append(("sf_decode (shafa_block_decompressor)", sum(enc_bytes) + sum(rle_n), t))
for b in (0, nb - 1):
    assert torch.equal(d_sym[roff[b]:roff[b] + rle_n[b]], d_rle[roff[b]:roff[b] + rle_n[b]])

d_dec = torch.empty(nb * (bs + 2048), dtype=torch.uint8, device=dev)
doff = [b * (bs + 2048) for b in range(nb)]
d_dec_n = torch.zeros(nb, dtype=torch.int64, device=dev)
t = timed(lambda: bt.rle_decode(st, d_sym, roff, rle_n, d_dec, doff, [bs + 1024] * nb, d_dec_n))
rows.append(("rle_decode (rle_block_decompressor)", sum(rle_n) + nb * bs, t))
assert [int(x) for x in d_dec_n.cpu().numpy()] == n
assert torch.equal(d_dec[:bs], blk)

print(f"{nb} x 64 MiB blocks of Zipf(1.2) runs; RLE size {sum(rle_n) / (nb * bs):.4f} n, SF size {sum(enc_bytes) / sum(rle_n):.4f} of that")
print("| kernel (reference function) | algorithmic bytes / block | ms / block | GB/s | fraction of 8 TB/s | GiB/s of original |")
print("|---|---|---|---|---|---|")
for name, alg, t in rows:
    print(f"| {name} | {alg / nb / 2**20:.1f} MiB | {t / nb * 1e3:.4f} | {alg / t / 1e9:.0f} | {alg / t / 1e9 / PEAK:.3f} | {nb * bs / t / 2**30:.0f} |")
