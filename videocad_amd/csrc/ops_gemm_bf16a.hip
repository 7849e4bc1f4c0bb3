// ops_gemm_bf16a.hip — bf16 weight-gradient (tra = trb = 1, fp32 output) instantiations of the register-staged GEMM
#include "gemm_launch.h"

int vc_gemm_launch_bf16_wgrad(GemmCall c, int nsplit, vc_stream_t s) {
    switch ((c.sa == VC_F32) * 2 + (c.sb == VC_F32)) {
        case 0: return gemm_launch<vc_bf16, vc_bf16, vc_bf16, float, true, true>(c, nsplit, s);
        case 1: return gemm_launch<vc_bf16, vc_bf16, float, float, true, true>(c, nsplit, s);
        case 2: return gemm_launch<vc_bf16, float, vc_bf16, float, true, true>(c, nsplit, s);
        default: return gemm_launch<vc_bf16, float, float, float, true, true>(c, nsplit, s);
    }
}
