// ops_gemm_x3.hip — bf16x3 (VCAD_BF16X3: fp32 tensors, hi/lo-split bf16 MFMAs) instantiations of the register-staged GEMM: fp32 activations
// (split while staging) against fp32 or pre-split weights.  The all-pre-split forms live in ops_gemm_x3b.hip (own translation unit: they compile in parallel).
#include "gemm_launch.h"

#ifdef VC_H16     // bf16-only mode: the fp16-storage build (libvcad_hip_f16.so) carries the entry point, not the kernels
int vc_gemm_launch_x3(GemmCall, int, int, vc_stream_t) { vc_set_error("vc_gemm: bf16x3 exists in the bf16 build only"); return VC_ERR_UNSUPPORTED; }
#else

int vc_gemm_launch_x3_pk(GemmCall c, int nsplit, int lay, vc_stream_t s);      // sa == VC_PK (ops_gemm_x3b.hip)

int vc_gemm_launch_x3(GemmCall c, int nsplit, int lay, vc_stream_t s) {
    if (c.sa == VC_PK) return vc_gemm_launch_x3_pk(c, nsplit, lay, s);
    if (c.to != VC_F32) { vc_set_error("vc_gemm: bf16x3 with fp32 activations writes fp32"); return VC_ERR_UNSUPPORTED; }
    if (c.sb == VC_PK) {          // pre-split weights (forward / dgrad layouts)
        if (lay == 0) return gemm_launch<vc_x3, float, vc_pk, float, false, false>(c, nsplit, s);
        if (lay == 1) return gemm_launch<vc_x3, float, vc_pk, float, false, true>(c, nsplit, s);
        vc_set_error("vc_gemm: pre-split B operand in layout %d", lay); return VC_ERR_UNSUPPORTED;
    }
    switch (lay) {
        case 0: return gemm_launch<vc_x3, float, float, float, false, false>(c, nsplit, s);
        case 1: return gemm_launch<vc_x3, float, float, float, false, true>(c, nsplit, s);
        case 2: return gemm_launch<vc_x3, float, float, float, true, false>(c, nsplit, s);
        default: return gemm_launch<vc_x3, float, float, float, true, true>(c, nsplit, s);
    }
}
#endif
