// ops_gemm_f32.hip — exact-fp32 (v_mfma_f32_32x32x2_f32) instantiations of the register-staged GEMM
#include "gemm_launch.h"

int vc_gemm_launch_f32(GemmCall c, int nsplit, int lay, vc_stream_t s) {
    switch (lay) {
        case 0: return gemm_launch<float, float, float, float, false, false>(c, nsplit, s);
        case 1: return gemm_launch<float, float, float, float, false, true>(c, nsplit, s);
        case 2: return gemm_launch<float, float, float, float, true, false>(c, nsplit, s);
        default: return gemm_launch<float, float, float, float, true, true>(c, nsplit, s);
    }
}
