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
run base A=1
run fold STTS_FOLD_ROWS=100000
run noc2sc STTS_NO_WINO_CONV2SC=1
run nowino STTS_NO_WINOGRAD=1
run base2 A=1
STTS_PROF_DUMP=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-legs --no-cpu-baseline --no-traffic > /dev/null 2> gpurun_out/x3/prof_dump.txt
grep "\[prof\]" gpurun_out/x3/prof_dump.txt | head -110
