// Translation unit of the conv_gemm_f32 instantiations for one operand form (gemm.hip.h: gemm_dispatch_x3); built in parallel with the others.
#define STTS_GEMM_TU_FORM 3
#include "gemm.hip.h"
