#!/usr/bin/env python3
"""host time of one shafa_hipd_sf_encode_tiles call against its GPU time at small launch sizes: what a back-to-back loop of
calls is bound by.   usage: enc_tiles_host_time.py [blocks] [MiB]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import pkgload
pkg = pkgload.load()
synth = pkgload.load_submodule("synth")
dev = torch.device("cuda", 0); pkg.lib().shafa_hip_init(0); st = torch.cuda.Stream(device=dev)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bs = (int(sys.argv[2]) if len(sys.argv) > 2 else 64) << 20
d_in = torch.from_numpy(synth.gen_bytes(5, bs, synth.zipf_table(1.2))).to(dev).repeat(nb)
bt = pkg.Batch(nb, bs)
off = [b * bs for b in range(nb)]; n = [bs] * nb
d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
thb = pkg.tile_hist_bytes(bs)
d_th = torch.empty(nb * thb, dtype=torch.uint8, device=dev)
thoff = [b * thb for b in range(nb)]
bt.hist256_tiles(st, d_in, off, n, d_freq, d_th, thoff); bt.finish(st, nb)
freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
tables = pkg.sf_build_codes_batch(freq)
cap = bs + bs // 8
eoff = [b * cap for b in range(nb)]
d_enc = torch.empty(nb * cap, dtype=torch.uint8, device=dev); d_n = torch.zeros(nb, dtype=torch.int64, device=dev)
call = lambda: bt.sf_encode_tiles(st, d_in, off, n, tables, d_th, thoff, d_enc, eoff, [cap] * nb, d_n)
for _ in range(3):
    call()
bt.finish(st, nb)
for steps in (8, 8, 32):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(st)
    for _ in range(steps):
        call()
    t1 = time.perf_counter(); e1.record(st)
    bt.finish(st, nb)
    print(f"{nb} block(s), {steps} calls back to back: host {1e6 * (t1 - t0) / steps:.1f} us per call, events {1e3 * e0.elapsed_time(e1) / steps:.1f} us per call")
