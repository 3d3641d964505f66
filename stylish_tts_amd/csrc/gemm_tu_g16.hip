// Translation unit of conv_gemm16_kernel (gemm16.hip.h: launch_conv_gemm16_main); built in parallel with the others.
#define STTS_GEMM16_TU
#include "gemm16.hip.h"
