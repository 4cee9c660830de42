#!/bin/bash
# decode kernel durations (rocprofv3) of several builds on one box; ablated builds allowed (SHAFA_BENCH_ABLATION=2)
# usage (through gpurun): tools/dbg/ab_kstats_dec.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
export SHAFA_BENCH_ORACLE_CHECK=0 SHAFA_BENCH_ABLATION=${ABL:-2} TMPDIR=/tmp
cp shafa-cd_amd/libshafa_hip.so /tmp/orig.so
for L in "$@"; do
  cp "$L" shafa-cd_amd/libshafa_hip.so
  out=/tmp/ks_$$; rm -rf $out; mkdir -p $out
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python3 "$R/bench.py" --no-cpu --no-host-path --no-pipeline --steps 5 --warmup 2 > /dev/null 2> $out/err.log)
  echo "== $(basename $L)"
  python3 - $out <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/r_kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if n.startswith("sfd_wstage") or n.startswith("sfd_scan"):
        print(f"   {n[:30]:32s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
done
cp /tmp/orig.so shafa-cd_amd/libshafa_hip.so
