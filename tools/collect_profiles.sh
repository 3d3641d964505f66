#!/bin/bash
# collect_profiles.sh: copy gpurun_out/prof_r03/* (tools/profile_r03.sh on the GPU box) into profiles/r03_* and render the per-kernel tables
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/prof_r03
for f in "$S"/*_bench.json "$S"/*_kernel_stats.csv "$S"/*.txt "$S"/n2_gloo_rehearsal_one_gpu.json; do
  [ -s "$f" ] && cp "$f" "profiles/r03_$(basename "$f")"
done
steps() { case "$1" in cfg2|bf16_b64|f16_16x10s|b1) echo 30;; cfg3) echo 12;; *) echo 0;; esac; }  # bench.py: 20 timed + 3 warm-up + 1 + 3 + 3; cfg3: 10 + 2
for c in profiles/r03_*_kernel_stats.csv; do
  n=$(basename "$c" _kernel_stats.csv); n=${n#r03_}
  st=$(steps "$n")
  if [ "$st" -gt 0 ]; then python3 tools/kstats_md.py "$c" "$st" > "profiles/r03_${n}_kernels.md"; else python3 tools/kstats_md.py "$c" > "profiles/r03_${n}_kernels.md"; fi
done
ls profiles | grep r03_
