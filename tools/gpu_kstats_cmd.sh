#!/bin/bash
# per-kernel averages of an arbitrary python tool: gpu_kstats_cmd.sh <tag> <script.py> [args]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/kstats_$tag
rm -rf $out; mkdir -p $out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python3 "$@" > $out/stdout.txt 2> $out/err.log
tail -3 $out/stdout.txt
f=$(find $out -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r'\(.*', '', r['Name'].replace('(anonymous namespace)::', '').replace('void ', ''))[:48]
    if float(r['Percentage']) < 0.5: continue
    print(f"{n:50s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f}  {float(r['Percentage']):5.1f}%")
PY
