#!/bin/bash
# lut_conflict_probe.py (its last two cases) with several ablation builds of libshafa_hip.so on one box
# usage (through gpurun): tools/dbg/probe_abl.sh a.so b.so ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp shafa-cd_amd/libshafa_hip.so /tmp/orig.so
for L in "$@"; do cp $L shafa-cd_amd/libshafa_hip.so; echo "== $L"; python tools/dbg/lut_conflict_probe.py 5 2>&1 | tail -2; done
cp /tmp/orig.so shafa-cd_amd/libshafa_hip.so
