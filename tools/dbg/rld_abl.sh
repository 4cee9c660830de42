#!/bin/bash
# rle_decode_kernel with parts removed (wrong output; builds from tools/dbg/mkvar.sh rle_decode.hip NAME:-DRLD_ABL_...):
# kernel time of each build on bench.py's two pipeline inputs.   usage (through gpurun): tools/dbg/rld_abl.sh a.so b.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
cp shafa-cd_amd/libshafa_hip.so /tmp/orig.so
for L in "$@"; do
  for kind in runs mixed; do
    cp "$L" shafa-cd_amd/libshafa_hip.so
    echo "$(basename $L) $(timeout 250 python3 tools/dbg/rld_stamps.py $kind 32 nostamps 2>&1 | grep "rle_decode" | tail -1)"
  done
done
cp /tmp/orig.so shafa-cd_amd/libshafa_hip.so
