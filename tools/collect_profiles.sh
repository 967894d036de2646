#!/bin/bash
# Copy the summaries of a tools/profile_round.sh run (gpurun_out/<round>/) into profiles/<round>_*
# (tracked): usage tools/collect_profiles.sh r3
set -e
cd "$(dirname "$0")/.."
R=${1:?round tag}
S=gpurun_out/$R
for f in $S/bench*.json $S/extra_configs.jsonl $S/full_attention_L*.json $S/pmc_*.csv $S/neck_pmc_*.csv $S/fa4096_pmc_*.csv $S/decoder_split_ab.txt $S/trunk_autocast.txt; do
  [ -f "$f" ] && cp "$f" profiles/${R}_$(basename "$f")
done
cpk() { [ -f "$S/$1" ] && cp "$S/$1" "profiles/${R}_$2" || true; }
cpk trace_kernel_stats.csv kernel_stats.csv
cpk trace64_kernel_stats.csv tile64_kernel_stats.csv
cpk trace_qk16_kernel_stats.csv qk16_kernel_stats.csv
cpk overlap_trace_kernel_stats.csv overlap_kernel_stats.csv
cpk neck_trace_kernel_stats.csv neck_kernel_stats.csv
cpk fa_trace_kernel_stats.csv full_attention_kernel_stats.csv
cpk fa4096_trace_kernel_stats.csv fa4096_kernel_stats.csv
git status --short profiles | head -40
