mkdir -p gpurun_out/x3
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-traffic > gpurun_out/x3/bench_x3_v3.json 2> gpurun_out/x3/bench_x3_v3.err || { tail gpurun_out/x3/bench_x3_v3.err; exit 1; }
python - <<'P'
import json
d=json.loads(open("gpurun_out/x3/bench_x3_v3.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], [ (k["kernel"],k["ms_per_step"]) for k in d["roofline"]["contraction_kernels"]])
P
bash tools/prof_one.sh x3v3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-legs | head -22
