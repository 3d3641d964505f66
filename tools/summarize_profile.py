"""Summarise a `rocprofv3 --kernel-trace --stats` run of bench.py into profiles/ (per-kernel ms/step, and the
contraction kernels' average launch duration to set beside bench.py's HIP-event figure)."""
import csv, glob, json, sys

src, out_prefix, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = glob.glob(src + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
lines = ["| kernel | launches/step | ms/step | avg us | share |", "|---|---|---|---|---|"]
g_ns = g_calls = 0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"].replace("stts::", "").replace("void ", "").split("(")[0]
    t, c = float(r["TotalDurationNs"]), int(r["Calls"])
    if "conv_gemm_f32" in n or "wn_layer" in n or "wn_fused" in n:
        g_ns += t
        g_calls += c
    elif "winograd_" in n:  # the transforms of a Winograd-form conv belong to its contraction launch
        g_ns += t
    if t / tot > 0.002:
        lines.append(f"| `{n}` | {c / steps:.1f} | {t / 1e6 / steps:.3f} | {t / c / 1e3:.1f} | {100 * t / tot:.1f} % |")
summary = {
    "source": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline (20 timed + 3 warm-up + 3 profiled steps)",
    "steps_in_trace": steps,
    "gpu_busy_ms_per_step": tot / 1e6 / steps,
    "contraction_kernels": "conv_gemm_f32<*> + wn_fused_kernel / wn_layer_kernel (+ winograd_input/output_kernel of the Winograd-form convs)",
    "contraction_launches_per_step": g_calls / steps,
    "contraction_avg_launch_us": g_ns / g_calls / 1e3,
    "contraction_ms_per_step": g_ns / 1e6 / steps,
}
try:  # algorithmic and executed flops per step: bench.py's own accounting (stts_profile_report)
    rl = json.load(open(out_prefix + "_bench.json"))["roofline"]
    summary["algorithmic_gflop_per_step"] = rl["algorithmic_gflop_per_step"]
    summary["executed_gflop_per_step"] = rl["executed_gflop_per_step"]
    summary["contraction_tflops_from_rocprof"] = rl["algorithmic_gflop_per_step"] * 1e9 / (g_ns / steps * 1e-9) / 1e12
    summary["contraction_executed_tflops_from_rocprof"] = rl["executed_gflop_per_step"] * 1e9 / (g_ns / steps * 1e-9) / 1e12
    summary["bench_event_avg_launch_us"] = 1e3 * rl["avg_launch_ms"]
except Exception:
    pass
# bench.py counts a Winograd-form conv (transforms + contraction) and a plain + remainder launch pair as ONE launch
try:
    logical = json.load(open(out_prefix + "_bench.json"))["roofline"]["launches_per_step"]
    summary["bench_launches_per_step"] = logical
    summary["contraction_avg_us_per_bench_launch"] = g_ns / steps / logical / 1e3
except Exception:
    pass
open(out_prefix + "_summary.json", "w").write(json.dumps(summary, indent=1))
open(out_prefix + "_kernels.md", "w").write("\n".join(lines) + "\n")
print(json.dumps(summary, indent=1))
