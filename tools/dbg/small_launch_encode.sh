#!/bin/bash
# SF encode of 1, 2, 4 and 8 blocks of 64 MiB per launch: the tile path (sfe6, tile histograms given) against the entry point
# without them (count / scan / pack below six blocks, the chained pass from six), and what the tile histograms cost (K1 of the
# pipeline leg at the same block count).   usage (through gpurun): tools/dbg/small_launch_encode.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SHAFA_BENCH_ORACLE_CHECK=0
for nb in 1 2 4 8; do
  python bench.py --no-cpu --no-host-path --no-pipeline --encode-only --blocks $nb --steps 20 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d.get('encode_chained') or {}
print('$nb blocks: tiles %.1f us (%.3f)   without %.1f us (%.3f)' % (d['encode_ms']*1e3, d['roofline_encode']['frac'], c.get('ms',0)*1e3, c.get('frac',0)))"
  python bench.py --no-cpu --no-host-path --pipeline-only --pipeline-kind runs --blocks $nb --pipeline-blocks $nb --steps 20 2>/dev/null | tail -1 | python3 -c "
import json,sys
p=json.loads(sys.stdin.read())['pipeline']
print('      K1 hist256_tiles %.1f us  K3 sf_encode_tiles %.1f us (runs data)' % (p['K1_hist256']['ms']*1e3, p['K3_sf_encode']['ms']*1e3))"
done
