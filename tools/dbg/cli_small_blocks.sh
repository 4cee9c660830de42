#!/bin/bash
# CLI at the reference's DEFAULT block size (64 KiB, file.h) and at -b K (640 KiB): a 256 MiB Zipf file in tmpfs, ours against the
# reference binary when it is there.   usage (through gpurun): tools/dbg/cli_small_blocks.sh [MiB]
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LD_LIBRARY_PATH=$PWD/shafa-cd_amd:$LD_LIBRARY_PATH
MIB=${1:-256}
D=/dev/shm/shafa_small; rm -rf $D; mkdir -p $D
python3 - $MIB $D <<'PY'
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
t() { s=$(date +%s%N); "$@" > /dev/null 2>$D/err; rc=$?; e=$(date +%s%N); echo "$(( (e-s)/1000000 )) ms (rc $rc)"; }
for B in "" "-b K" "-b m"; do
  for exe in shafa-cd_amd/bin/shafa oracle/_ref/shafa; do
    [ -x $exe ] || continue
    rm -f $D/z.*; 
    echo "$exe ${B:-default 64 KiB}: f $(t $exe $D/z -m f $B)  t $(t $exe $D/z.freq -m t)  c $(t $exe $D/z -m c)  d $(t $exe $D/z.shaf -m d)"
  done
done
rm -rf $D
