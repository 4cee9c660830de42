#!/bin/bash
# Same-box K2/K5 times of builds of libshafa_hip.so over bench.py's two pipeline legs (32 x 64 MiB each), two rounds.
# usage (through gpurun): tools/dbg/abpipe.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
export SHAFA_BENCH_ORACLE_CHECK=0
[ -n "$ABL" ] && export SHAFA_BENCH_ABLATION=$ABL      # ABL=2: builds whose decoders write wrong bytes (timing only)
cp shafa-cd_amd/libshafa_hip.so /tmp/orig.so
for i in 1 2; do
  for L in "$@"; do
    cp "$L" shafa-cd_amd/libshafa_hip.so
    for kind in runs mixed; do
      echo "$(basename $L) $kind $(timeout 300 python bench.py --no-cpu --no-host-path --pipeline-only --pipeline-kind $kind --steps 3 2>&1 | tail -1 | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); p=d["pipeline"]
print("  ".join("%s %.3f ms %.3f" % (k[:2], v["ms"], v["frac"]) for k,v in p.items() if isinstance(v, dict) and k[:2] in ("K2","K5")))' 2>&1 | tail -1)"
    done
  done
done
cp /tmp/orig.so shafa-cd_amd/libshafa_hip.so
