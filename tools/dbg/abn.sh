#!/bin/bash
# Same-box comparison of several builds of libshafa_hip.so: runs bench.py --no-cpu --no-host-path with each, twice, prints encode / decode ms.
# usage (through gpurun): tools/dbg/abn.sh "<bench args>" lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
ARGS=$1; shift
export SHAFA_BENCH_ORACLE_CHECK=0
cp shafa-cd_amd/libshafa_hip.so /tmp/orig.so
for i in 1 2; do
  for L in "$@"; do
    cp "$L" shafa-cd_amd/libshafa_hip.so
    echo "$(basename $L) $(timeout 300 python bench.py --no-cpu --no-host-path $ARGS 2>&1 | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("enc %.3f dec %s" % (d["encode_ms"], d["decode_ms"]))' 2>&1 | tail -1)"
  done
done
cp /tmp/orig.so shafa-cd_amd/libshafa_hip.so
