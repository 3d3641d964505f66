export STTS_LIB=stylish_tts_amd/libstylish_hip_trace.so B=64 SHAPES="out_conv 768->1024,dec conv2"
for T in 1216 1262 64; do echo "== TUNE=$T"; TILES=14,15 TUNE=$T timeout -k 10 120 python tools/gemm_bench.py 2>&1 | grep -v "amdgpu.ids\|percentiles\|by cu_id\|latest start" || exit 1; done
