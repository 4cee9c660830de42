#!/bin/bash
# The round's profile set in one call (through gpurun): headline alone, the two pipeline legs alone, config[1].
# Summaries and traffic.json land in gpurun_out/prof_<tag>/; copy them to profiles/<round>_<tag>_* afterwards (tools/copy_profiles.sh r6).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
tools/gpu_prof.sh main --steps 5 --warmup 2 --no-pipeline > /dev/null 2>&1
tools/gpu_prof.sh pipeline_runs --pipeline-only --pipeline-kind runs --steps 3 > /dev/null 2>&1
tools/gpu_prof.sh pipeline_mixed --pipeline-only --pipeline-kind mixed --steps 3 > /dev/null 2>&1
tools/gpu_prof.sh cfg1_uniform_8mib --dist uniform --block-mib 8 --blocks 128 --no-pipeline > /dev/null 2>&1
tools/gpu_prof.sh tiles_main --steps 5 --warmup 2 --no-pipeline --tiles > /dev/null 2>&1
for t in main pipeline_runs pipeline_mixed cfg1_uniform_8mib tiles_main; do echo "=== $t"; head -60 gpurun_out/prof_$t/summary.txt | cut -c1-260; done
