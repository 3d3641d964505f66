mkdir -p gpurun_out/x3
B=8 TILES=x5,p25,p27,x6,p26,p28 timeout -k 10 300 python tools/gemm_bench.py > gpurun_out/x3/gemm_b8_v6.log 2>&1 || { tail gpurun_out/x3/gemm_b8_v6.log; exit 1; }
cat gpurun_out/x3/gemm_b8_v6.log
