#!/bin/bash
# collect_profiles.sh: copy gpurun_out/prof_r04/* (tools/profile_r04.sh on the GPU box) into profiles/r04_* and render the per-kernel tables
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/prof_r04
for f in "$S"/*_bench.json "$S"/*_kernel_stats.csv "$S"/*.txt "$S"/n4_gloo_rehearsal_one_gpu.json; do
  [ -s "$f" ] && cp "$f" "profiles/r04_$(basename "$f")"
done
steps() { case "$1" in cfg2|cfg2_f32_matrix_cores) echo 32;; bf16_b64|f16_16x10s|b1) echo 30;; cfg3) echo 12;; *) echo 0;; esac; }  # bench.py: 20 timed + 3 warm-up + 1 + 3 + 3; cfg3: 10 + 2
for c in profiles/r04_*_kernel_stats.csv; do
  n=$(basename "$c" _kernel_stats.csv); n=${n#r04_}
  st=$(steps "$n")
  if [ "$st" -gt 0 ]; then python3 tools/kstats_md.py "$c" "$st" > "profiles/r04_${n}_kernels.md"; else python3 tools/kstats_md.py "$c" > "profiles/r04_${n}_kernels.md"; fi
done
ls profiles | grep r04_
