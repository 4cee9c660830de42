#!/bin/bash
# Same-box timing of builds of libshafa_hip.so that differ in sf_encode6.hip only (ablations with wrong output included:
# SHAFA_BENCH_ABLATION=1 skips bench.py's checks): encode ms of each build, three rounds, then the kernel durations of the
# first one.   usage (through gpurun): tools/dbg/ab6.sh "<bench args>" lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
ARGS=$1; shift
export SHAFA_BENCH_ORACLE_CHECK=0 SHAFA_BENCH_ABLATION=1
cp shafa-cd_amd/libshafa_hip.so /tmp/orig.so
for i in 1 2 3; do
  for L in "$@"; do
    cp "$L" shafa-cd_amd/libshafa_hip.so
    echo "$(basename $L) $(timeout 300 python bench.py --no-cpu --no-host-path --no-pipeline --encode-only $ARGS 2>&1 | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("enc %.3f frac %.3f" % (d["encode_ms"], d["roofline_encode"]["frac"]))' 2>&1 | tail -1)"
  done
done
cp "$1" shafa-cd_amd/libshafa_hip.so
unset SHAFA_BENCH_ABLATION
tools/gpu_kstats.sh --no-pipeline --encode-only --steps 8 $ARGS
cp /tmp/orig.so shafa-cd_amd/libshafa_hip.so
