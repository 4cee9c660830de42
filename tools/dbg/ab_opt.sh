#!/bin/bash
# Same-box A/B of two option settings of ONE build: alternates `bench.py --no-cpu --no-host-path --opt A` and `--opt B` three times and
# prints encode / decode ms.   usage (through gpurun): tools/dbg/ab_opt.sh name=va name=vb [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
A=$1; B=$2; shift 2
export SHAFA_BENCH_ORACLE_CHECK=0
for i in 1 2 3; do
  for v in "$A" "$B"; do
    echo "$v $(timeout 300 python bench.py --no-cpu --no-host-path --opt "$v" "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("enc %.3f dec %s frac_enc %.3f" % (d["encode_ms"], d["decode_ms"], d["roofline_encode"]["frac"]))')"
  done
done
