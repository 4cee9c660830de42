"""Host time of one shafa_hipd_sf_decode call vs its GPU time (default uniform 128 x 8 MiB): finds host work on the critical
path.  usage: dec_host_time.py [uniform|zipf|runs] [blocks] [block MiB]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch
import pkgload
pkg = pkgload.load()
import oracle_lib
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 128
bs = (int(sys.argv[3]) if len(sys.argv) > 3 else 8) << 20
dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
d_in = torch.empty(nb * bs, dtype=torch.uint8, device=dev)
if dist == "runs":
    synth = pkgload.load_submodule("synth")
    d_in.copy_(torch.from_numpy(synth.runs_stream(11, bs, synth.zipf_table(1.2))).to(dev).repeat(nb))
elif dist == "uniform":
    d_in.copy_(torch.randint(0, 256, (nb * bs,), dtype=torch.uint8, device=dev))
else:
    zt = torch.from_numpy(pkg.zipf_table(1.2)).to(dev)
    d_in.copy_(zt[torch.randint(0, 65536, (nb * bs,), device=dev)])
bt = pkg.Batch(nb, bs + 65536)
off, n = [b * bs for b in range(nb)], [bs] * nb
d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
bt.hist256(st, d_in, off, n, d_freq); bt.finish(st, nb)
freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
tables = bt._tables([pkg.sf_build_codes(freq[b]) for b in range(nb)])
cap = bs + bs // 4
eoff = [b * cap for b in range(nb)]
d_enc = torch.empty(nb * cap, dtype=torch.uint8, device=dev)
d_enc_n = torch.zeros(nb, dtype=torch.int64, device=dev)
bt.sf_encode(st, d_in, off, n, tables, d_enc, eoff, [cap] * nb, d_enc_n); bt.finish(st, nb)
enc_n = [int(x) for x in d_enc_n.cpu().numpy()]
d_out = torch.empty(nb * bs, dtype=torch.uint8, device=dev)
for spec in (1, 0, 1):
    pkg.set_option("sf_decode_speculate", spec)
    for it in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(st)
        bt.sf_decode(st, d_enc, eoff, enc_n, tables, n, d_out, off)
        t1 = time.perf_counter(); e1.record(st)
        bt.finish(st, nb)
        print(f"spec={spec} it={it} host call {1e3*(t1-t0):.3f} ms, events {e0.elapsed_time(e1):.3f} ms")
assert torch.equal(d_out, d_in)
