run() { # name env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], round(d['ms_per_step'],4), [ (k['kernel'],k['launches_per_step'],k['ms_per_step']) for k in d['roofline']['contraction_kernels']], d['roofline']['all_launches_per_step'])"
}
run base A=1
run norem STTS_NO_REM=1
run base A=1
run norem STTS_NO_REM=1
