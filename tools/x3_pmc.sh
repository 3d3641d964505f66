# SQ counters of the split-fp32 contraction (decoder conv2 shape, tile 5): instruction mix, LDS activity and conflicts, memory waits
mkdir -p gpurun_out/x3
export TMPDIR=/tmp SHAPES="dec conv2" B=8 TILES=x5
run() {
  cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/x3/pmc_$1 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py > $GRAFT_REPO_ROOT/gpurun_out/x3/pmc_$1.log 2>&1
  cd $GRAFT_REPO_ROOT && python3 - "$1" <<'P'
import csv, glob, collections, sys
f = glob.glob(f"gpurun_out/x3/pmc_{sys.argv[1]}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    if "conv_gemm" in k: print(k, {c: f"{x:.4g}" for c, x in v.items()})
P
}
run SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES
run SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES
