cd ${GRAFT_REPO_ROOT:-/root/repo}
export SHAFA_BENCH_ORACLE_CHECK=0
for cfg in "64 32" "16 128" "4 512" "64 8"; do set -- $cfg
for kind in runs mixed; do
echo "$1 MiB x $2 $kind $(timeout 300 python bench.py --no-cpu --no-host-path --pipeline-only --pipeline-kind $kind --steps 3 --block-mib $1 --blocks $2 --pipeline-blocks $2 2>&1 | tail -1 | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); p=d["pipeline"]
print("  ".join("%s %.3f ms %.3f" % (k[:2], v["ms"], v["frac"]) for k,v in p.items() if isinstance(v, dict) and k[:2] in ("K1","K2","K3","K4","K5")))' 2>&1 | tail -1)"
done; done
