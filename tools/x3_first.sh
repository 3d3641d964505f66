# first GPU pass of the split-fp32 contractions: parity, per-shape tile sweep, step time (runs on the GPU box)
mkdir -p gpurun_out/x3
timeout -k 10 500 python -m pytest tests/test_hip_split_fp32.py tests/test_hip_frame_path.py -m gpu -x -q -s > gpurun_out/x3/tests.log 2>&1 || { tail -30 gpurun_out/x3/tests.log; exit 1; }
tail -5 gpurun_out/x3/tests.log
B=8 TILES=5,x5,6,x6,8,x8,x20,x21,x22 timeout -k 10 300 python tools/gemm_bench.py > gpurun_out/x3/gemm_b8.log 2>&1 || { tail gpurun_out/x3/gemm_b8.log; exit 1; }
cat gpurun_out/x3/gemm_b8.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-legs > gpurun_out/x3/bench_x3.json 2> gpurun_out/x3/bench_x3.err || { tail gpurun_out/x3/bench_x3.err; exit 1; }
STTS_NO_X3=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-legs > gpurun_out/x3/bench_native.json 2> gpurun_out/x3/bench_native.err || { tail gpurun_out/x3/bench_native.err; exit 1; }
python - <<'P'
import json
for n in ("x3","native"):
    d=json.loads(open(f"gpurun_out/x3/bench_{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], [ (k["kernel"],k["ms_per_step"]) for k in d["roofline"]["contraction_kernels"]])
P
