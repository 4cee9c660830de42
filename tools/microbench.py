import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, numpy as np
import pkgload
pkg = pkgload.load()
L = pkg.lib()
L.shafa_hip_microbench.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
n = 1 << 30
d_in = torch.empty(n, dtype=torch.uint8, device=dev)
zt = torch.from_numpy(pkg.zipf_table(1.2)).to(dev)
st = torch.cuda.Stream()
pkg.gen_bytes(st, 1, 0, d_in, n, zt)
st.synchronize()
lut = torch.from_numpy(((np.arange(256) % 13 + 2).astype(np.uint32) << 16 | (np.arange(256).astype(np.uint32) * 37 % 1024))).to(dev)
out = torch.zeros(n // 16 + 4096, dtype=torch.int32, device=dev)
for (mode, items, threads) in [(6,2,256),(7,2,256),(8,2,256),(6,4,256),(7,4,256),(8,4,256)]:
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            rc = L.shafa_hip_microbench(mode, d_in.data_ptr(), n, lut.data_ptr(), out.data_ptr(), items, threads, C.c_void_p(st.cuda_stream))
            assert rc == 0, rc
        e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"mode={mode} items={items} threads={threads}: {ms:.3f} ms/GiB  {1.0737/ms:.2f} TB/s read")
