mkdir -p gpurun_out/x3
timeout -k 10 900 python -m pytest tests/test_hip_frame_path.py tests/test_hip_benchmarked_path.py tests/test_hip_full_size.py tests/test_hip_split_fp32.py -m gpu -x -q > gpurun_out/x3/tests18.log 2>&1 || { tail -40 gpurun_out/x3/tests18.log; exit 1; }
tail -2 gpurun_out/x3/tests18.log
run() { # name env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-traffic $BARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name $BARGS', d['value'], round(d['ms_per_step'],4), d['roofline']['all_launches_per_step'])"
}
run new A=1
run rem STTS_X3_REM=1
for B in 1 2 4 12 16 32 64; do BARGS="--batch $B" run new A=1; BARGS="--batch $B" run rem STTS_X3_REM=1; done
BARGS="--workload cfg4 --steps 5 --warmup 2" 
for v in A=1 STTS_X3_REM=1; do env $v timeout -k 10 300 python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 $v', d['value'], round(d['ms_per_step'],3))"; done
