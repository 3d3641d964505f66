#!/bin/bash
# Runs ON THE GPU BOX (via gpurun).  Everything bench.py's JSON claims can be recomputed from what lands here:
#   bench.json            the default run (cfg2, fp32): value, roofline (live per-kernel report, in-run rocprofv3 --pmc traffic), cpu_baseline
#   trace/                rocprofv3 --kernel-trace --stats of the same command (legs off) -> per-kernel durations
#   bench_b1.json         --batch 1: launches per step and microseconds per launch (SURVEY 8d "sanity of the target")
#   bench_cfg3_bf16.json  the cfg3-shaped frame path (B = 64, bf16 operands); bench_cfg5_f16.json: 16 x 10 s, fp16
#   bench_cfg4_n1.json    --workload cfg4 on one GPU (the 256 mixed-length utterances as one rank's work)
# tools/summarize_profile.py turns trace/ into profiles/<prefix>_kernels.md / _summary.json.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_round
rm -rf "$O" && mkdir -p "$O"
cd "$R" && python3 bench.py > "$O/bench.json" 2> "$O/bench.err"
python3 bench.py --batch 1 --no-cpu-baseline --no-traffic > "$O/bench_b1.json" 2>> "$O/bench.err"
python3 bench.py --batch 64 --precision bf16 --no-cpu-baseline --no-traffic > "$O/bench_cfg3_bf16.json" 2>> "$O/bench.err"
python3 bench.py --batch 16 --mel-frames 800 --precision f16 --no-cpu-baseline --no-traffic > "$O/bench_cfg5_f16.json" 2>> "$O/bench.err"
python3 bench.py --workload cfg4 --steps 3 --warmup 1 > "$O/bench_cfg4_n1.json" 2>> "$O/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$O/trace" -o trace --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --no-traffic > "$O/trace.log" 2>&1
rm -f "$O"/trace/*kernel_trace.csv  # per-dispatch rows (tens of MB): the stats file is what gets committed
ls -R "$O" | head -30
