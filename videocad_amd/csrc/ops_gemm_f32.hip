// ops_gemm_f32.hip — exact-fp32 (v_mfma_f32_32x32x2_f32) instantiations of the register-staged GEMM
#include "gemm_launch.h"

#ifdef VC_H16     // VCAD_F32 engines live in the bf16 build: the fp16-storage build carries the entry point, not the kernels
int vc_gemm_launch_f32(GemmCall, int, int, vc_stream_t) { vc_set_error("vc_gemm: the exact-fp32 GEMM exists in the bf16 build only"); return VC_ERR_UNSUPPORTED; }
#else

int vc_gemm_launch_f32(GemmCall c, int nsplit, int lay, vc_stream_t s) {
    switch (lay) {
        case 0: return gemm_launch<float, float, float, float, false, false>(c, nsplit, s);
        case 1: return gemm_launch<float, float, float, float, false, true>(c, nsplit, s);
        case 2: return gemm_launch<float, float, float, float, true, false>(c, nsplit, s);
        default: return gemm_launch<float, float, float, float, true, true>(c, nsplit, s);
    }
}
#endif
