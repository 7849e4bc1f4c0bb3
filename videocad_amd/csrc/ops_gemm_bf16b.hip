// ops_gemm_bf16b.hip — bf16 forward (lay 0: k-contiguous bf16 weights; A and the output bf16 or fp32) instantiations of the register-staged GEMM; the
// dgrad layout lives in ops_gemm_bf16c.hip (own translation unit: the two compile in parallel — together they were the longest file of the build)
#include "gemm_launch.h"

int vc_gemm_launch_bf16_dgrad(GemmCall c, int nsplit, vc_stream_t s);      // lay 1 (ops_gemm_bf16c.hip)

int vc_gemm_launch_bf16(GemmCall c, int nsplit, int lay, vc_stream_t s) {
    if (lay != 0) return vc_gemm_launch_bf16_dgrad(c, nsplit, s);
    switch ((c.sa == VC_F32) * 2 + (c.to == VC_F32)) {
        case 0: return gemm_launch<vc_bf16, vc_bf16, vc_bf16, vc_bf16, false, false>(c, nsplit, s);
        case 1: return gemm_launch<vc_bf16, vc_bf16, vc_bf16, float, false, false>(c, nsplit, s);
        case 2: return gemm_launch<vc_bf16, float, vc_bf16, vc_bf16, false, false>(c, nsplit, s);
        default: return gemm_launch<vc_bf16, float, vc_bf16, float, false, false>(c, nsplit, s);
    }
}
