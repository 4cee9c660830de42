#!/bin/bash
# A/B of two builds of libshafa_hip.so on ONE box (boxes differ by several per cent): alternates the two libraries and
# prints encode / decode ms of bench.py for each run.   usage (through gpurun): tools/dbg/ab.sh <libA.so> <libB.so> [bench args]
# Build the variants here first, e.g.  make -C shafa-cd_amd/csrc FLAGS_EXTRA=-DX && cp shafa-cd_amd/libshafa_hip.so _ab/b.so
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
A=$1; B=$2; shift 2
cp shafa-cd_amd/libshafa_hip.so /tmp/orig.so
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then cp "$A" shafa-cd_amd/libshafa_hip.so; else cp "$B" shafa-cd_amd/libshafa_hip.so; fi
    echo "$v $(timeout 200 python bench.py --no-cpu --no-host-path "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("enc %.3f dec %s" % (d["encode_ms"], d["decode_ms"]))')"
  done
done
cp /tmp/orig.so shafa-cd_amd/libshafa_hip.so
