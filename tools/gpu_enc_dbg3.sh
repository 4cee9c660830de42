#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for d in 0 4 8 12 44; do
  rm -rf $R/gpurun_out/pd; SHAFA_ENC_DBG=$d timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pd -- python3 $R/bench.py --blocks 16 --steps 3 --warmup 1 --no-cpu --encode-only > /dev/null 2>&1
  echo -n "dbg=$d pack max ns: "; grep sfe3_pack $(find $R/gpurun_out/pd -name "*kernel_stats.csv") | awk -F, '{print $(NF-1)}'
done
