#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in 0 8 16 24; do
echo "dbg=$d"; SHAFA_ENC_DBG=$d SHAFA_ENC_PROF=1 timeout 120 python bench.py --blocks 32 --steps 2 --warmup 1 --no-cpu --encode-only 2>&1 | grep "enc prof" | tail -1
SHAFA_ENC_DBG=$d timeout 120 python bench.py --blocks 32 --steps 5 --warmup 2 --no-cpu --encode-only 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('  encode_ms', round(j['encode_ms'],3))"
done
