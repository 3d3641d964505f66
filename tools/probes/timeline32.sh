# fp32 block timelines at B = 8 (trace build): prologue / K loop / epilogue per block and the shader clock
export STTS_LIB=stylish_tts_amd/libstylish_hip_trace.so B=8
for T in 5 6 8; do echo "== tile $T"; TILES=$T TUNE=64 SHAPES="pwconv1,pwconv2,dec conv2,wino plane" timeout -k 10 120 python tools/gemm_bench.py 2>&1 | grep -v "amdgpu.ids\|percentiles\|by cu_id\|latest start" || exit 1; done
