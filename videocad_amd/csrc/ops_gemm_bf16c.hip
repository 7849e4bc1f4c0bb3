// ops_gemm_bf16c.hip — bf16 dgrad (lay 1: row-contiguous bf16 weights, dY . W; A and the output bf16 or fp32) instantiations of the register-staged GEMM
#include "gemm_launch.h"

int vc_gemm_launch_bf16_dgrad(GemmCall c, int nsplit, vc_stream_t s) {
    switch ((c.sa == VC_F32) * 2 + (c.to == VC_F32)) {
        case 0: return gemm_launch<vc_bf16, vc_bf16, vc_bf16, vc_bf16, false, true>(c, nsplit, s);
        case 1: return gemm_launch<vc_bf16, vc_bf16, vc_bf16, float, false, true>(c, nsplit, s);
        case 2: return gemm_launch<vc_bf16, float, vc_bf16, vc_bf16, false, true>(c, nsplit, s);
        default: return gemm_launch<vc_bf16, float, vc_bf16, float, false, true>(c, nsplit, s);
    }
}
