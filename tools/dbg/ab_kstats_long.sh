#!/bin/bash
# decode kernel durations of several builds on data with long codes (Zipf truncated to 256 ranks: Lmax 16..26 at 64 MiB)
# usage (through gpurun): tools/dbg/ab_kstats_long.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
export SHAFA_BENCH_ORACLE_CHECK=0 TMPDIR=/tmp
cp shafa-cd_amd/libshafa_hip.so /tmp/orig.so
for L in "$@"; do
  cp "$L" shafa-cd_amd/libshafa_hip.so
  for S in 1.2 2.0; do
    out=/tmp/ks_$$; rm -rf $out; mkdir -p $out
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python3 "$R/bench.py" --no-cpu --no-host-path --no-pipeline --dist zipf --zipf-s $S --blocks 32 --steps 5 --warmup 2 > $out/line.json 2> $out/err.log)
    echo "== $(basename $L) zipf $S: $(python3 -c "import json,sys; d=json.load(open('$out/line.json')); print('decode %.3f ms' % d['decode_ms'])" 2>/dev/null)"
    python3 - $out <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/r_kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if n.startswith('sfd_wstage') or n.startswith('sfd_scan<false'):
        print(f"   {n[:30]:32s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
  done
done
cp /tmp/orig.so shafa-cd_amd/libshafa_hip.so
