mkdir -p gpurun_out/x3
timeout -k 10 900 python -m pytest tests/test_hip_full_size.py tests/test_hip_frame_path.py -m gpu -x -q > gpurun_out/x3/tests14.log 2>&1 || { tail -40 gpurun_out/x3/tests14.log; exit 1; }
tail -2 gpurun_out/x3/tests14.log
for B in 8 16 32 64; do
timeout -k 10 300 python bench.py --batch $B --steps 10 --warmup 3 --no-legs --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B', d['value'], round(d['ms_per_step'],3))"
done
timeout -k 10 300 python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4', d['value'], round(d['ms_per_step'],3))"
