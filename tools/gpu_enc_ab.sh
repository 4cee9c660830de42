#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 3; do
  for dist in zipf uniform; do
  echo -n "v=$v $dist "; SHAFA_ENC_V=$v timeout 120 python bench.py --blocks 32 --steps 5 --warmup 2 --no-cpu --encode-only --dist $dist 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('encode_ms', round(j['encode_ms'],3), 'GiB/s', round(j['encode_GiBs'],1), 'frac', round(j['roofline_encode']['frac'],3))"
  done
done
