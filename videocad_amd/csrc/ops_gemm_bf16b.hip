// ops_gemm_bf16b.hip — bf16 forward / dgrad (B = bf16 weights; A and the output bf16 or fp32) instantiations of the register-staged GEMM
#include "gemm_launch.h"

int vc_gemm_launch_bf16(GemmCall c, int nsplit, int lay, vc_stream_t s) {
    const int key = (c.sa == VC_F32) * 2 + (c.to == VC_F32);
    if (lay == 0) switch (key) {
        case 0: return gemm_launch<vc_bf16, vc_bf16, vc_bf16, vc_bf16, false, false>(c, nsplit, s);
        case 1: return gemm_launch<vc_bf16, vc_bf16, vc_bf16, float, false, false>(c, nsplit, s);
        case 2: return gemm_launch<vc_bf16, float, vc_bf16, vc_bf16, false, false>(c, nsplit, s);
        default: return gemm_launch<vc_bf16, float, vc_bf16, float, false, false>(c, nsplit, s);
    }
    switch (key) {
        case 0: return gemm_launch<vc_bf16, vc_bf16, vc_bf16, vc_bf16, false, true>(c, nsplit, s);
        case 1: return gemm_launch<vc_bf16, vc_bf16, vc_bf16, float, false, true>(c, nsplit, s);
        case 2: return gemm_launch<vc_bf16, float, vc_bf16, vc_bf16, false, true>(c, nsplit, s);
        default: return gemm_launch<vc_bf16, float, vc_bf16, float, false, true>(c, nsplit, s);
    }
}
