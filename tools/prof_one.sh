#!/bin/bash
# prof_one.sh NAME bench-args...   (on the GPU box): rocprofv3 kernel stats of `python3 bench.py <args>` -> gpurun_out/prof/NAME_kernel_stats.csv + table
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p "$O"
export TMPDIR=/tmp
name=$1; shift
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$O/${name}_trace" -o t --output-format csv -- python3 "$R/bench.py" "$@" > "$O/${name}_trace.log" 2>&1)
rc=$?
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[profile] $name timed out" >&2; exit $rc; fi
cp "$O/${name}_trace"/*/*kernel_stats.csv "$O/${name}_kernel_stats.csv" 2>/dev/null || cp "$O/${name}_trace"/*kernel_stats.csv "$O/${name}_kernel_stats.csv"
rm -rf "$O/${name}_trace"
python3 "$R/tools/kstats_md.py" "$O/${name}_kernel_stats.csv" ${STEPS:-32} > "$O/${name}_kernels.md"
cat "$O/${name}_kernels.md"
