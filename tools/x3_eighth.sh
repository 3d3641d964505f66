mkdir -p gpurun_out/x3
run() { # name env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-traffic > gpurun_out/x3/ab_$name.json 2> gpurun_out/x3/ab_$name.err || { tail gpurun_out/x3/ab_$name.err; exit 1; }
  python - "$name" <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/x3/ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], [ (k["kernel"],k["launches_per_step"],k["ms_per_step"]) for k in d["roofline"]["contraction_kernels"]])
P
}
run v8 A=1
run v8c2 STTS_NO_WINO_CONV2SC=1
run v8b A=1
run v8c2b STTS_NO_WINO_CONV2SC=1
