mkdir -p gpurun_out/x3
timeout -k 10 600 python -m pytest tests/test_hip_split_fp32.py tests/test_hip_frame_path.py tests/test_hip_benchmarked_path.py -m gpu -x -q > gpurun_out/x3/tests16.log 2>&1 || { tail -40 gpurun_out/x3/tests16.log; exit 1; }
tail -2 gpurun_out/x3/tests16.log
B=8 TILES=5,x5,x6 SHAPES="pwconv1,pwconv2,dec conv,wino plane,out_conv 768->1024" timeout -k 10 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu
run() { # name env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-traffic > gpurun_out/x3/ab_$name.json 2> gpurun_out/x3/ab_$name.err || { tail gpurun_out/x3/ab_$name.err; exit 1; }
  python - "$name" <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/x3/ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], [ (k["kernel"],k["launches_per_step"],k["ms_per_step"]) for k in d["roofline"]["contraction_kernels"]])
P
}
run stag A=1
run stag2 A=1
