#!/bin/bash
# gpurun_out/prof_<tag>/ of tools/gpu_prof_all.sh -> profiles/<round>_<tag>_{summary.txt,traffic.json,kernel_stats.csv}
# usage: tools/copy_profiles.sh r4
R=/root/repo
for t in main pipeline_runs pipeline_mixed cfg1_uniform_8mib tiles_main; do
  d=$R/gpurun_out/prof_$t
  cp $d/summary.txt $R/profiles/$1_${t}_summary.txt
  cp $d/traffic.json $R/profiles/$1_${t}_traffic.json
  cp "$(find $d/kstats -name '*kernel_stats.csv' | head -1)" $R/profiles/$1_${t}_kernel_stats.csv
done
grep -o "csrc [0-9a-f]*" $R/profiles/$1_main_summary.txt | head -1
