#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--blocks 8 --steps 2 --warmup 1 --no-cpu"
rm -rf $R/gpurun_out/pmcD1 $R/gpurun_out/pmcD2
timeout 180 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmcD1 -- python3 $R/bench.py $ARGS > /dev/null 2>&1
timeout 180 rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmcD2 -- python3 $R/bench.py $ARGS > /dev/null 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmcD1 | grep -A8 "sfd_count13\|sfd_write13\|sfd_sync16"
python3 tools/pmc_summary.py gpurun_out/pmcD2 | grep -A8 "sfd_count13\|sfd_write13\|sfd_sync16"
