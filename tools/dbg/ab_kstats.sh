#!/bin/bash
# A/B of two builds of libshafa_hip.so by KERNEL durations (rocprofv3 --kernel-trace --stats, min and average per kernel),
# alternating A B A B on one box: for changes smaller than the run-to-run spread of bench.py's whole-step times.
# usage (through gpurun): tools/dbg/ab_kstats.sh <libA.so> <libB.so> <kernel name pattern> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
A=$1; B=$2; PAT=$3; shift 3
cp shafa-cd_amd/libshafa_hip.so /tmp/orig.so
for i in 1 2; do
  for v in A B; do
    if [ $v = A ]; then cp "$A" shafa-cd_amd/libshafa_hip.so; else cp "$B" shafa-cd_amd/libshafa_hip.so; fi
    echo "== $v"
    tools/gpu_kstats.sh --no-pipeline --steps 8 "$@" | grep -E "$PAT"
  done
done
cp /tmp/orig.so shafa-cd_amd/libshafa_hip.so
