#!/bin/bash
# Same-box A/B of the bench line under experiment switches (runs on the GPU box): x3_ab.sh "NAME ENV=V ..." ["NAME2 ENV2=V" ...]   BARGS="--batch 16" for other shapes
#   e.g.  bash tools/x3_ab.sh "default A=1" "f32mfma STTS_NO_X3=1" "rem STTS_X3_REM=1" "flow_f32 STTS_WN_X3=-1" "default A=1"
for spec in "$@"; do
  set -- $spec; name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-traffic $BARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$BARGS $name', d['value'], round(d['ms_per_step'],4), [(k['kernel'],k['launches_per_step'],k['ms_per_step']) for k in d['roofline']['contraction_kernels']], d['roofline']['all_launches_per_step'])" || exit 1
done
