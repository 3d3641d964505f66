mkdir -p gpurun_out/x3
timeout -k 10 600 python -m pytest tests/test_hip_frame_path.py -m gpu -x -q -k "flow or wavenet" > gpurun_out/x3/tests15.log 2>&1 || { tail -40 gpurun_out/x3/tests15.log; exit 1; }
tail -2 gpurun_out/x3/tests15.log
for B in 1 2 4; do
for v in "A=1" "STTS_WN_X3=-1" "A=1" "STTS_WN_X3=-1"; do
env $v timeout -k 10 300 python bench.py --batch $B --steps 30 --warmup 5 --no-legs --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B $v', d['value'], round(d['ms_per_step'],3), [(k['kernel'],k['ms_per_step']) for k in d['roofline']['contraction_kernels'] if 'wn' in k['kernel']])"
done; done
