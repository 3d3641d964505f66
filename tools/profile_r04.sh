#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): one bench JSON + one `rocprofv3 --kernel-trace --stats` summary per number README / DESIGN quote.
#   cfg2        bench.py (the bench line: B = 8 x 3 s fp32; roofline, in-run --pmc traffic, cpu_baseline, legs)
#   bf16_b64    --batch 64 --precision bf16 (cfg3-shaped frame path)       f16_16x10s  --batch 16 --mel-frames 800 --precision f16 (cfg5-shaped)
#   cfg3        --workload cfg3 (tokens -> waveform, 64 x 50 tokens, bf16 frame path)
#   b1          --batch 1 (latency)                                        cfm / full_chain: tools/cfm_bench.py, tools/full_chain_bench.py
# tools/kstats.py turns <name>_kernel_stats.csv into the per-kernel tables under profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r04
rm -rf "$O" && mkdir -p "$O"
export TMPDIR=/tmp
trace() {  # trace NAME script args...: kernel stats of `python3 script args`
  local name=$1; shift
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$O/${name}_trace" -o t --output-format csv -- python3 "$R/$1" "${@:2}" > "$O/${name}_trace.log" 2>&1)
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[profile] $name trace timed out: stopping" >&2; exit $rc; fi
  cp "$O/${name}_trace"/*/*kernel_stats.csv "$O/${name}_kernel_stats.csv" 2>/dev/null || cp "$O/${name}_trace"/*kernel_stats.csv "$O/${name}_kernel_stats.csv" 2>/dev/null
  rm -rf "$O/${name}_trace"
  echo "[profile] $name done (rc $rc)" >&2
}
cd "$R"
timeout -k 10 700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/cfg2_bench.json" 2> "$O/cfg2_bench.err" || echo "[profile] cfg2 bench rc $?" >&2
trace cfg2 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-legs
trace cfg2_f32_matrix_cores bench.py --steps 20 --warmup 5 --precision f32_native --no-cpu-baseline --no-traffic --no-legs
B=8 TILES=5,x5,6,x6,x20,x22 timeout -k 10 300 python3 tools/gemm_bench.py > "$O/gemm_x3_tiles_b8.txt" 2>&1
for cfgline in "bf16_b64 --batch 64 --precision bf16" "f16_16x10s --batch 16 --mel-frames 800 --precision f16" "b1 --batch 1"; do
  set -- $cfgline; name=$1; shift
  timeout -k 10 200 python3 bench.py "$@" --no-cpu-baseline --no-traffic --no-legs > "$O/${name}_bench.json" 2> "$O/${name}_bench.err"
  trace "$name" bench.py "$@" --no-cpu-baseline --no-traffic --no-legs
done
timeout -k 10 200 python3 bench.py --workload cfg3 --steps 10 --warmup 2 > "$O/cfg3_bench.json" 2> "$O/cfg3_bench.err"
trace cfg3 bench.py --workload cfg3 --steps 10 --warmup 2
timeout -k 10 200 python3 tools/cfm_bench.py > "$O/cfm_bench.txt" 2>&1
trace cfm tools/cfm_bench.py
timeout -k 10 200 python3 tools/full_chain_bench.py > "$O/full_chain_bench.txt" 2>&1
trace full_chain tools/full_chain_bench.py
trace full_chain_b1 tools/full_chain_b1.py
STTS_BENCH_ONE_GPU=1 STTS_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 4 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > "$O/n4_gloo_rehearsal_one_gpu.json" 2> "$O/n4_gloo_rehearsal_one_gpu.err"
if [ -x tools/probes/bin/gemm16_probe ]; then ABL=2 timeout -k 10 300 tools/probes/bin/gemm16_probe > "$O/gemm16_ablation.txt" 2>&1; fi
timeout -k 10 300 python3 tools/debug/cap_bench.py > "$O/capacity_bench.txt" 2>&1
ls -la "$O"
