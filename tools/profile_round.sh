#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the bench line, the rocprofv3 kernel-trace summary of the same command, and the two
# HBM-traffic counter passes (collected separately from the trace, as the pool requires).  Results land under
# gpurun_out/prof_round/; tools/summarize_profile.py and tools/pmc_summary.py turn them into profiles/<prefix>_*.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_round
rm -rf "$O" && mkdir -p "$O"
cd "$R" && python3 bench.py > "$O/bench.json" 2> "$O/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$O/trace" -o trace --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline > "$O/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$O/pmc_fetch" -o pmc --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --steps 2 --warmup 1 > "$O/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$O/pmc_write" -o pmc --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --steps 2 --warmup 1 > "$O/pmc_write.log" 2>&1
ls -R "$O" | head -40
