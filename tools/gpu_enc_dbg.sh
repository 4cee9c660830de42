#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in 0 1 2 3 4 8 12 15; do
  echo -n "dbg=$d "; SHAFA_ENC_DBG=$d timeout 120 python bench.py --blocks 16 --steps 5 --warmup 2 --no-cpu --encode-only 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('encode_ms', round(j['encode_ms'],3), 'GiB/s', round(j['encode_GiBs'],1))"
done
