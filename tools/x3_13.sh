mkdir -p gpurun_out/x3
timeout -k 10 600 python -m pytest tests/test_hip_frame_path.py tests/test_hip_benchmarked_path.py tests/test_hip_full_size.py -m gpu -x -q > gpurun_out/x3/tests13.log 2>&1 || { tail -40 gpurun_out/x3/tests13.log; exit 1; }
tail -2 gpurun_out/x3/tests13.log
run() { # name env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-traffic > gpurun_out/x3/ab_$name.json 2> gpurun_out/x3/ab_$name.err || { tail gpurun_out/x3/ab_$name.err; exit 1; }
  python - "$name" <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/x3/ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], [ (k["kernel"],k["launches_per_step"],k["ms_per_step"]) for k in d["roofline"]["contraction_kernels"]])
P
}
run wcopy A=1
run xaff STTS_GRN_XAFF=1
run wcopy2 A=1
run xaff2 STTS_GRN_XAFF=1
