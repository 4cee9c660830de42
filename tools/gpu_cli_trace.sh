#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
D=/dev/shm/shafa_e2e; rm -rf $D; mkdir -p $D
timeout 300 python3 - ${1:-8} $D <<'PY'
import sys, os, torch
sys.path.insert(0, "tests")
from pkgload import load
shafa = load()
nb, d = int(sys.argv[1]), sys.argv[2]
zt = torch.from_numpy(shafa.zipf_table(1.2)).cuda()
buf = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
with open(os.path.join(d, "z"), "wb") as f:
    for b in range(nb):
        shafa.gen_bytes(None, 1234, b * (64 << 20), buf, 64 << 20, zt)
        torch.cuda.synchronize()
        f.write(buf.cpu().numpy().tobytes())
PY
export LD_LIBRARY_PATH=$PWD/shafa-cd_amd:$LD_LIBRARY_PATH
O=shafa-cd_amd/bin/shafa
timeout 300 $O $D/z -m f -b M > /dev/null; timeout 60 $O $D/z.freq -m t > /dev/null
s=$(date +%s%N); SHAFA_TRACE=1 timeout 300 $O $D/z -m c 2>$D/trace_c.txt >/dev/null; e=$(date +%s%N); head -${3:-30} $D/trace_c.txt; echo ...; tail -${2:-40} $D/trace_c.txt; echo "total -m c $(( (e-s)/1000000 )) ms"
cp $D/z $D/z.orig
s=$(date +%s%N); SHAFA_TRACE=1 timeout 300 $O $D/z.shaf -m d 2>&1 >/dev/null | tail -${2:-40}; e=$(date +%s%N); echo "total -m d $(( (e-s)/1000000 )) ms"
cmp $D/z $D/z.orig && echo "round trip identical"
rm -rf $D
