#!/bin/bash
# End-to-end CLI timing on the GPU box (file I/O in tmpfs + PCIe included): our bin/shafa vs the
# reference binary (oracle/_ref/shafa) on the same Zipf file.  usage: gpu_cli_e2e.sh <blocks of 64 MiB>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
NB=${1:-16}
D=/dev/shm/shafa_e2e
rm -rf $D; mkdir -p $D/ours $D/ref
timeout 300 python3 - "$NB" "$D" <<'PY'
import sys, importlib.util, os, torch
sys.path.insert(0, "tests")
from pkgload import load
shafa = load()
nb, d = int(sys.argv[1]), sys.argv[2]
zt = torch.from_numpy(shafa.zipf_table(1.2)).cuda()
buf = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
with open(os.path.join(d, "z.bin"), "wb") as f:
    for b in range(nb):
        shafa.gen_bytes(None, 1234, b * (64 << 20), buf, 64 << 20, zt)
        torch.cuda.synchronize()
        f.write(buf.cpu().numpy().tobytes())
print("generated", nb, "blocks")
PY
cp $D/z.bin $D/ours/z; mv $D/z.bin $D/ref/z
t() { local s=$(date +%s%N); "$@" > /dev/null 2>$D/err.txt; local rc=$?; local e=$(date +%s%N); printf "%-66s rc=%d %6d ms\n" "$*" $rc $(( (e - s) / 1000000 )); }
export LD_LIBRARY_PATH=$PWD/shafa-cd_amd:$LD_LIBRARY_PATH
O=shafa-cd_amd/bin/shafa; R=oracle/_ref/shafa
echo "== ours (3 blocks in flight)"
t timeout 600 $O $D/ours/z -m f -b M
t timeout 600 $O $D/ours/z.freq -m t
t timeout 600 $O $D/ours/z -m c
cp $D/ours/z $D/ours/z.orig
t timeout 600 $O $D/ours/z.shaf -m d
cmp $D/ours/z $D/ours/z.orig && echo "ours: round trip identical"
echo "== ours --no-multithread (1 block in flight)"
rm -f $D/ours/z.shaf
t timeout 600 $O $D/ours/z -m c --no-multithread
t timeout 600 $O $D/ours/z.shaf -m d --no-multithread
echo "== reference binary (thread per block)"
t timeout 900 $R $D/ref/z -m f -b M
t timeout 900 $R $D/ref/z.freq -m t
t timeout 900 $R $D/ref/z -m c
cmp $D/ours/z.shaf $D/ref/z.shaf && echo "z.shaf identical to the reference's"
cmp $D/ours/z.cod $D/ref/z.cod && echo "z.cod identical to the reference's"
t timeout 900 $R $D/ref/z.shaf -m d
rm -rf $D
