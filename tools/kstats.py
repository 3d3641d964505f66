"""Print the top kernels of a rocprofv3 --kernel-trace --stats run of bench.py (steps = timed + warm-up + profiled)."""
import csv, glob, sys
src = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 26
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
f = glob.glob(src + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    n = r["Name"].replace("stts::", "").replace("void ", "")[:84]
    print(f"{n:84s} {int(r['Calls']) / steps:6.1f} {float(r['TotalDurationNs']) / 1e6 / steps:7.3f} ms/step avg {float(r['AverageNs']) / 1e3:7.1f} us min {float(r['MinNs']) / 1e3:6.1f} {100 * float(r['TotalDurationNs']) / tot:5.1f}%")
print(f"GPU busy {tot / 1e6 / steps:.3f} ms/step")
