#!/bin/bash
# HBM traffic of every kernel: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md §HBM)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--blocks 16 --steps 2 --warmup 1 --no-cpu"
rm -rf $R/gpurun_out/pmcF $R/gpurun_out/pmcW
timeout 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcF -- python3 $R/bench.py $ARGS > /dev/null 2>&1
timeout 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcW -- python3 $R/bench.py $ARGS > /dev/null 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmcF
python3 tools/pmc_summary.py gpurun_out/pmcW
