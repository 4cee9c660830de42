#!/bin/bash
# encoder variant 4 vs 5 over several workloads on one box (through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
export SHAFA_BENCH_ORACLE_CHECK=0
run() { python bench.py --no-cpu --no-host-path --encode-only --steps 8 "$@" 2>&1 | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("enc %.3f ms frac %.3f" % (d["encode_ms"], d["roofline_encode"]["frac"]))' 2>&1 | tail -1; }
for W in "--dist uniform --block-mib 8 --blocks 128" "--dist zipf" "--blocks 16" "--blocks 32" "--zipf-s 1.6" "--zipf-s 2.0" "--dist zipf --zipf-s 2.0"; do
  echo "[$W] chained: $(run $W)  tiles: $(run $W --tiles)"
done
