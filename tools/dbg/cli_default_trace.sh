cd ${GRAFT_REPO_ROOT:-/root/repo}
export LD_LIBRARY_PATH=$PWD/shafa-cd_amd:$LD_LIBRARY_PATH
D=/dev/shm/shafa_small; rm -rf $D; mkdir -p $D
python3 - 256 $D <<'PY'
import sys, os, torch
sys.path.insert(0, "tests")
from pkgload import load
shafa = load()
mib, d = int(sys.argv[1]), sys.argv[2]
zt = torch.from_numpy(shafa.zipf_table(1.2)).cuda()
buf = torch.empty(mib << 20, dtype=torch.uint8, device="cuda")
shafa.gen_bytes(None, 99, 0, buf, mib << 20, zt)
torch.cuda.synchronize()
open(os.path.join(d, "z"), "wb").write(buf.cpu().numpy().tobytes())
PY
O=shafa-cd_amd/bin/shafa
$O $D/z -m f > /dev/null; $O $D/z.freq -m t > /dev/null
for m in c; do s=$(date +%s%N); SHAFA_TRACE=1 $O $D/z -m c 2>&1 >/dev/null | grep -E "files open|pipe created|loop done|destroyed"; e=$(date +%s%N); echo "total -m c $(( (e-s)/1000000 )) ms"; done
cp $D/z $D/z.orig
s=$(date +%s%N); SHAFA_TRACE=1 $O $D/z.shaf -m d 2>&1 >/dev/null | grep -E "files open|pipe created|loop done|destroyed"; e=$(date +%s%N); echo "total -m d $(( (e-s)/1000000 )) ms"
cmp $D/z $D/z.orig && echo identical
rm -rf $D
