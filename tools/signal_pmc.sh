# SQ counters + HBM bytes of the bandwidth- / LDS-bound kernels of the 16-bit step at B = 64 (stft, istft frames / overlap-add, AdaIN apply, weight scaling, LayerNorm):
# which unit each of them keeps busy.  Runs ON THE GPU BOX: bash tools/signal_pmc.sh
mkdir -p gpurun_out/sig
export TMPDIR=/tmp
ARGS="--batch 64 --precision bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-legs"
run() {
  local tag=$1; shift
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/sig/pmc_$tag -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/sig/pmc_$tag.log 2>&1) || { echo "pmc pass $tag failed"; tail -3 gpurun_out/sig/pmc_$tag.log; exit 1; }
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
run c FETCH_SIZE
run d WRITE_SIZE
python3 - <<'P'
import csv, glob, collections, json
want = ("stft_kernel", "istft_frames_kernel", "istft_ola_kernel", "adain_apply_kernel", "scale_weight_kernel", "row_layernorm_kernel", "single_channel_conv_kernel", "dwconv_ln_kernel", "pcph_kernel", "cast_rows_kernel")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter()); dur = collections.defaultdict(list)
for tag in "abcd":
    f = glob.glob(f"gpurun_out/sig/pmc_{tag}/**/*counter_collection.csv", recursive=True)
    for r in csv.DictReader(open(f[0])):
        k = next((w for w in want if w in r["Kernel_Name"]), None)
        if not k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    if tag == "a":
        t = glob.glob(f"gpurun_out/sig/pmc_{tag}/**/*kernel_trace.csv", recursive=True)
        for r in csv.DictReader(open(t[0])):
            k = next((w for w in want if w in r["Kernel_Name"]), None)
            if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k in want:
    if k not in acc: continue
    v = {c: acc[k][c] / n[k][c] for c in acc[k]}  # per dispatch
    d = sorted(dur[k])[len(dur[k]) // 2] if dur[k] else 0.0
    wc = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    out[k] = {"us_per_launch_median_under_pmc": round(d, 1), "launches_sampled": int(n[k]["SQ_WAVE_CYCLES"]),
              "hbm_read_MB": round(v.get("FETCH_SIZE", 0) * 1024 * 2 / 1e6, 1), "hbm_write_MB": round(v.get("WRITE_SIZE", 0) * 1024 / 1e6, 1),
              "wave_cycles_issuing_valu": round(v.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3), "wave_cycles_issuing_lds": round(v.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3),
              "wave_cycles_issuing_vmem": round(v.get("SQ_ACTIVE_INST_VMEM", 0) / wc, 3), "wave_cycles_waiting_lds": round(v.get("SQ_WAIT_INST_LDS", 0) / wc, 3),
              "wave_cycles_waiting_any": round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
              "lds_array_busy_of_kernel": round(v.get("SQ_LDS_IDX_ACTIVE", 0) / max(v.get("GRBM_GUI_ACTIVE", 0) * 256 / 8, 1.0), 3) if v.get("GRBM_GUI_ACTIVE") else None,
              "lds_bank_conflict_of_lds_cycles": round(v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0), 1.0), 3),
              "insts_per_launch": {c: int(v[c]) for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if c in v}, "raw_per_dispatch": {c: round(x, 1) for c, x in v.items()}}
    if d: out[k]["hbm_TB_s"] = round((out[k]["hbm_read_MB"] + out[k]["hbm_write_MB"]) / d, 2)  # MB / us = TB/s
json.dump(out, open("gpurun_out/sig/signal_pmc.json", "w"), indent=1)
for k, v in out.items(): print(k, {a: b for a, b in v.items() if a != "raw_per_dispatch"})
P
