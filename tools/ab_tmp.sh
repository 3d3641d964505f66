mkdir -p gpurun_out/r04j
python -m pytest tests/test_hip_benchmarked_path.py tests/test_hip_frame_path.py tests/test_hip_full_size.py -m gpu -x -q > gpurun_out/r04j/tests.log 2>&1; tail -3 gpurun_out/r04j/tests.log
B() { python bench.py "$@" --no-legs --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], round(d['ms_per_step'],3), d['roofline']['frac'], d['roofline'].get('frac_executed'))"; }
for i in 1 2; do
  echo -n "wino conv2+sc cfg2: "; B
  echo -n "direct cfg2: "; STTS_NO_WINO_CONV2SC=1 B
done
echo -n "wino b16: "; B --batch 16
echo -n "direct b16: "; STTS_NO_WINO_CONV2SC=1 B --batch 16
