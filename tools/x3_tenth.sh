mkdir -p gpurun_out/x3
timeout -k 10 600 python -m pytest tests/test_hip_frame_path.py tests/test_hip_split_fp32.py tests/test_hip_benchmarked_path.py -m gpu -x -q > gpurun_out/x3/tests10.log 2>&1 || { tail -40 gpurun_out/x3/tests10.log; exit 1; }
tail -2 gpurun_out/x3/tests10.log
run() { # name env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-traffic > gpurun_out/x3/ab_$name.json 2> gpurun_out/x3/ab_$name.err || { tail gpurun_out/x3/ab_$name.err; exit 1; }
  python - "$name" <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/x3/ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], [ (k["kernel"],k["launches_per_step"],k["ms_per_step"]) for k in d["roofline"]["contraction_kernels"]])
P
}
run blk A=1
run layer STTS_WN_X3B=-1
run blk4 STTS_WN_X3B=4
run blkb A=1
for B in 16 32 64; do
for v in "A=1" "STTS_WN_X3B=-1" "STTS_WN_X3B=3" "STTS_WN_X3=-1"; do
env $v timeout -k 10 300 python bench.py --batch $B --steps 10 --warmup 3 --no-legs --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B $v', d['value'], round(d['ms_per_step'],3), [(k['kernel'],k['ms_per_step']) for k in d['roofline']['contraction_kernels'] if 'wn' in k['kernel']])"
done; done
