#!/usr/bin/env python3
"""kstats_md.py <name>_kernel_stats.csv [steps]: markdown table of a `rocprofv3 --kernel-trace --stats` summary (kernels above 0.3 % of the GPU time).
With `steps` the launches and milliseconds are per step (bench.py: timed + warm-up + 1 + 3 + 3 steps of the roofline leg)."""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
unit = "/step" if steps else ""
print(f"| kernel | launches{unit} | ms{unit} | avg us | min us | share |\n|---|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    t, c = float(r["TotalDurationNs"]), int(r["Calls"])
    if t / tot < 0.003:
        continue
    n = r["Name"].replace("stts::", "").replace("void ", "").split("(")[0]
    d = steps or 1
    print(f"| `{n}` | {c / d:.1f} | {t / 1e6 / d:.3f} | {t / c / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | {100 * t / tot:.1f} % |")
print(f"\nGPU busy: {tot / 1e6 / (steps or 1):.3f} ms{unit}; {sum(int(r['Calls']) for r in rows) / (steps or 1):.0f} launches{unit}")
