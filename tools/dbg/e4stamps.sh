#!/bin/bash
# diagnostic: rebuild libshafa_hip.so with -DE4_STAMPS on the GPU box, run the bench once, print per-phase cycle shares
cd "${GRAFT_REPO_ROOT:-/root/repo}"
# build first, here: touch shafa-cd_amd/csrc/sf_encode4.hip; make -C shafa-cd_amd/csrc FLAGS_EXTRA=-DE4_STAMPS   (then rebuild without the flag)

python3 - <<'PY'
import ctypes as C, os, sys, subprocess, json
import numpy as np
sys.path.insert(0, "tests")
import torch
import pkgload
pkg = pkgload.load()
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu", "--encode-only"]
import bench
try:
    bench.main()
except SystemExit:
    pass
L = pkg.lib()
n = 2048 * 16 * 8
buf = (C.c_ulonglong * n)()
rc = L.shafa_e4_read_stamps(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 16, 8).astype(np.float64)
a = a[a.sum(axis=(1, 2)) > 0]
names = ["w0 lookback", "prefetch+octs", "w0 ticket", "wait A", "offsets+place", "store|request", "wait B", "rotate+w0 publish/lead"]
tot = a.sum(axis=2)
print("workgroups with stamps:", a.shape[0], " mean cycles per wave:", tot.mean())
for w in range(16):
    sh = a[:, w, :].mean(axis=0)
    if sh.sum() == 0: continue
    print(f"wave {w}: " + "  ".join(f"{names[i]}={sh[i] / sh.sum() * 100:.1f}%" for i in range(8)))
PY
