mkdir -p gpurun_out/x3
B=8 TILES=x5,x6,x2,x20,x22,p25 timeout -k 10 300 python tools/gemm_bench.py > gpurun_out/x3/gemm_b8_v7.log 2>&1 || { tail gpurun_out/x3/gemm_b8_v7.log; exit 1; }
cat gpurun_out/x3/gemm_b8_v7.log
timeout -k 10 600 python -m pytest tests/test_hip_split_fp32.py tests/test_hip_frame_path.py -m gpu -x -q > gpurun_out/x3/tests7.log 2>&1 || { tail -30 gpurun_out/x3/tests7.log; exit 1; }
tail -2 gpurun_out/x3/tests7.log
run() { # name env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-traffic > gpurun_out/x3/ab_$name.json 2> gpurun_out/x3/ab_$name.err || { tail gpurun_out/x3/ab_$name.err; exit 1; }
  python - "$name" <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/x3/ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], [ (k["kernel"],k["launches_per_step"],k["ms_per_step"]) for k in d["roofline"]["contraction_kernels"]])
P
}
run nopre7 STTS_NO_X3_PRESPLIT=1
run pre7regs STTS_X3P_GLDS=0
