#!/usr/bin/env python3
"""bsum.py bench.json...: one-screen summary of bench.py JSON lines (value, roofline families, bandwidth-bound kernels)."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e); continue
    r = d.get("roofline", {})
    print(f"== {f}: {d['value']} {d['unit']}  {d['ms_per_step']:.3f} ms/step  frac {r.get('frac')} executed {r.get('frac_executed')} traffic {r.get('traffic')}")
    print("   ", {k: r.get(k) for k in ("gemm_ms_per_step", "all_kernel_ms_per_step", "all_launches_per_step", "launches_per_step")}, r.get("event_calibration", {}).get("scale"))
    for k in r.get("contraction_kernels", []):
        print(f"    {k['kernel']:24s} n={k['launches_per_step']:3d} {k['ms_per_step']:.3f} ms avg {k['avg_us']:7.1f} us  {k['tflops']:7.1f} TF/s (exec {k['executed_tflops']})")
    tot = 0
    for k in r.get("hbm_kernels", []):
        tot += k['launches_per_step'] * k['avg_us']
        print(f"    {k['kernel']:28s} n={k['launches_per_step']:3d} avg {k['avg_us']:7.1f} us  total {k['launches_per_step']*k['avg_us']:7.1f}  frac {k['frac_of_8tb_s']}")
    print(f"    hbm-kernel total {tot:.0f} us")
    for k in d:
        if k not in ("roofline", "config", "cpu_baseline", "metric", "value", "unit", "ms_per_step") and isinstance(d[k], dict):
            print("   ", k, json.dumps(d[k])[:600])
    if "cpu_baseline" in d: print("    cpu_baseline", d["cpu_baseline"])
