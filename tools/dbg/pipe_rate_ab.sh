#!/bin/bash
# layer 3 (bin/pipe_rate, 32 x 64 MiB, 3 slots) over builds of libshafa_hip.so on one box, three rounds
# usage (through gpurun): tools/dbg/pipe_rate_ab.sh a.so b.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
export LD_LIBRARY_PATH=$R/shafa-cd_amd:$LD_LIBRARY_PATH
cp shafa-cd_amd/libshafa_hip.so /tmp/orig.so
for i in 1 2 3; do
  for L in "$@"; do
    cp "$L" shafa-cd_amd/libshafa_hip.so
    echo "$(basename $L) $(timeout 120 shafa-cd_amd/bin/pipe_rate 32 64 ${SLOTS:-3} 2>&1 | grep -o "sf_[a-z]*\|[0-9.]* GiB/s\|round trip [a-z]*" | tr '\n' ' ')"
  done
done
cp /tmp/orig.so shafa-cd_amd/libshafa_hip.so
