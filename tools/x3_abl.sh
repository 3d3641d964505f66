# split-fp32 K-loop study on the trace build (timing-only ablations: results are wrong by design) + SQ counters of one launch shape
mkdir -p gpurun_out/x3
export STTS_LIB=stylish_tts_amd/libstylish_hip_trace.so B=8 SHAPES="dec conv2,pwconv1,wino plane 768" TILES=x5
for T in 0 2 4 8 32 33 12 44 45 46; do echo "== TUNE=$T (+1 no MFMA, +2 no barrier, +4 no ds_write/split, +8 no global loads, +32 no ds_read)"; TUNE=$T timeout -k 10 120 python tools/gemm_bench.py 2>&1 | grep -v "amdgpu.ids" || exit 1; done
echo "== block timeline x5"; TUNE=64 timeout -k 10 120 python tools/gemm_bench.py 2>&1 | grep -v "amdgpu.ids\|percentiles\|by cu_id\|latest start"
unset STTS_LIB
export TMPDIR=/tmp SHAPES="dec conv2"
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $GRAFT_REPO_ROOT/gpurun_out/x3/pmc -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py > $GRAFT_REPO_ROOT/gpurun_out/x3/pmc.log 2>&1
cd $GRAFT_REPO_ROOT && python3 - <<'P'
import csv, glob, collections
f = glob.glob("gpurun_out/x3/pmc/**/*counter_collection.csv", recursive=True)
print(f)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:90]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    if "conv_gemm" in k: print(k, {c: f"{x:.3g}" for c, x in v.items()})
P
