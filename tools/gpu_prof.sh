#!/bin/bash
# One-stop profile of a bench.py configuration on the GPU box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats   -> per-kernel average duration
#   2. two --pmc passes of SQ counters     -> instruction mix, LDS activity / conflicts, waits
#   3. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md §HBM)
# usage: tools/gpu_prof.sh <tag> [bench args...]      output: gpurun_out/prof_<tag>/{summary.txt,...}
# Counters are collected in their own runs (never together with a trace); the oracle library is built before the
# profiler starts so that nothing is spawned from a process the profiler has attached to.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
make -C oracle --no-print-directory > /dev/null 2>&1
export SHAFA_BENCH_ORACLE_CHECK=0
export TMPDIR=/tmp
tag=$1; shift
ARGS="--no-cpu --no-host-path $*"
out=$R/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kstats" -o r -- python3 "$R/bench.py" $ARGS > "$out/bench.json" 2> "$out/err_kstats.log"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d "$out/pmc1" -- python3 "$R/bench.py" $ARGS > /dev/null 2> "$out/err_pmc1.log"
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY --output-format csv -d "$out/pmc2" -- python3 "$R/bench.py" $ARGS > /dev/null 2> "$out/err_pmc2.log"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmcF" -- python3 "$R/bench.py" $ARGS > /dev/null 2> "$out/err_pmcF.log"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmcW" -- python3 "$R/bench.py" $ARGS > /dev/null 2> "$out/err_pmcW.log"
cd "$R"
python3 tools/prof_summary.py "$out" "$tag" $ARGS > "$out/summary.txt" 2>&1
# keep only the summaries (the raw CSVs are large)
find "$out" -name '*.csv' ! -name '*kernel_stats.csv' -delete 2>/dev/null
cat "$out/summary.txt"
