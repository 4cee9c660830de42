#!/bin/bash
# per-family times of one pipeline leg for several builds of libshafa_hip.so, same box:  tools/dbg/pipe_ab.sh <kind> lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
KIND=$1; shift
cp shafa-cd_amd/libshafa_hip.so /tmp/orig.so
for i in 1 2; do
  for L in "$@"; do
    cp "$L" shafa-cd_amd/libshafa_hip.so
    echo "$(basename $L) $(timeout 300 python bench.py --pipeline-only --pipeline-kind $KIND --steps 3 2>/dev/null | tail -1 | python3 -c 'import json,sys; p=json.loads(sys.stdin.read())["pipeline"]; print(" ".join("%s %.3f" % (k[:2], p[k]["ms"]) for k in ("K1_hist256","K2_rle_encode_hist","K3_sf_encode","K4_sf_decode","K5_rle_decode")))' 2>&1 | tail -1)"
  done
done
cp /tmp/orig.so shafa-cd_amd/libshafa_hip.so
