# 16-bit K-loop study (runs on the GPU box): tiles 14/15 (register staging) vs 16/17 (LDS-DMA), then timing-only ablations on a -DSTTS_GEMM_TRACE build
export B=64 TILES=14,16,15,17 SHAPES="out_conv 768->1024,dec conv2,pwconv2,pwconv1,prior"
echo "== product build, bf16 x16 rows (TUNE=1152)"; TUNE=1152 timeout -k 10 200 python tools/gemm_bench.py || exit 1
if [ -f stylish_tts_amd/libstylish_hip_trace.so ]; then
  export STTS_LIB=stylish_tts_amd/libstylish_hip_trace.so SHAPES="out_conv 768->1024,dec conv2"
  for T in 1152 1154 1160 1184 1194; do echo "== trace build TUNE=$T (+2 no barrier, +4 no ds_write, +8 no global loads, +32 no ds_read)"; TUNE=$T timeout -k 10 120 python tools/gemm_bench.py || exit 1; done
fi
