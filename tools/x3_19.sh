run() { # name env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-traffic $BARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name $BARGS', d['value'], round(d['ms_per_step'],4), [(k['kernel'],k['ms_per_step']) for k in d['roofline']['contraction_kernels'] if 'wn' in k['kernel']])"
}
for B in 8 10 12 16 20 24 32; do BARGS="--batch $B" run auto A=1; BARGS="--batch $B" run rt2 STTS_WN_X3=2; BARGS="--batch $B" run rt4 STTS_WN_X3=4; done
