mkdir -p gpurun_out/x3
# phoneme-rate contractions in split form: parity (durations bit-equal?) and speed
STTS_PHONEME_X3=1 timeout -k 10 900 python -m pytest tests/test_hip_phoneme_path.py tests/test_hip_phoneme_sizes.py tests/test_hip_modules.py tests/test_hip_capacity.py -m gpu -q > gpurun_out/x3/tests12.log 2>&1; tail -15 gpurun_out/x3/tests12.log
for v in "A=1" "STTS_PHONEME_X3=1" "A=1" "STTS_PHONEME_X3=1"; do
env $v timeout -k 10 300 python bench.py --workload cfg3 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 $v', d['value'], round(d['ms_per_step'],3), d['roofline']['phoneme_rate']['ms'], d['roofline']['frame_rate']['ms'])"
done
for v in "A=1" "STTS_PHONEME_X3=1"; do env $v timeout -k 10 300 python tools/full_chain_bench.py 2>&1 | grep -v amdgpu | head -12; done
