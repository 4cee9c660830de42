#!/bin/bash
# Quick A/B of kernel durations on the GPU box (run through gpurun): rocprofv3 --kernel-trace --stats of one bench.py
# configuration, top kernels by total time.   usage: tools/gpu_kstats.sh [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
make -C oracle --no-print-directory > /dev/null 2>&1
export SHAFA_BENCH_ORACLE_CHECK=0
export TMPDIR=/tmp
out=$R/gpurun_out/kstats_quick
rm -rf "$out"; mkdir -p "$out"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o r -- python3 "$R/bench.py" --no-cpu --no-host-path "$@" > "$out/bench.json" 2> "$out/err.log"
python3 - "$out" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/r_kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:int(__import__("os").environ.get("KSTATS_ROWS", "9"))]:
    name = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]
    print(f"{name:42s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.1f} min_us {float(r['MinNs'])/1e3:9.1f}")
PY
