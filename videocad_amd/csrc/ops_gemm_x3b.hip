// ops_gemm_x3b.hip — bf16x3 GEMMs whose operands are BOTH pre-split (hi | lo words, gemm.h vc_pk): what the ViT's Linears run on in the bf16x3 mode
// since r04 — producers (LayerNorm, GEMM epilogues, attention) emit pre-split tensors, so no tile is split more than once (the k-loop of the
// fp32-source form is VALU-bound on the split: 281 VALU per 24 MFMAs, profiles/r03_x3_pmc.md).  Outputs: fp32 (residual stream, LayerNorm
// inputs, weight gradients) or pre-split again (operands of the next GEMM / the attention kernels).
#include "gemm_launch.h"

#ifdef VC_H16     // bf16-only mode: the fp16-storage build (libvcad_hip_f16.so) carries the entry point, not the kernels
int vc_gemm_launch_x3_pk(GemmCall, int, int, vc_stream_t) { vc_set_error("vc_gemm: bf16x3 exists in the bf16 build only"); return VC_ERR_UNSUPPORTED; }
#else

int vc_gemm_launch_x3_pk(GemmCall c, int nsplit, int lay, vc_stream_t s) {
    if (c.sb != VC_PK) { vc_set_error("vc_gemm: a pre-split A operand needs a pre-split B operand"); return VC_ERR_UNSUPPORTED; }
    const bool pk_out = c.to == VC_PK;
    if (lay == 0) return pk_out ? gemm_launch<vc_x3, vc_pk, vc_pk, vc_pk, false, false>(c, nsplit, s) : gemm_launch<vc_x3, vc_pk, vc_pk, float, false, false>(c, nsplit, s);
    if (lay == 1) return pk_out ? gemm_launch<vc_x3, vc_pk, vc_pk, vc_pk, false, true>(c, nsplit, s) : gemm_launch<vc_x3, vc_pk, vc_pk, float, false, true>(c, nsplit, s);
    if (lay == 3 && !pk_out) return gemm_launch<vc_x3, vc_pk, vc_pk, float, true, true>(c, nsplit, s);
    vc_set_error("vc_gemm: pre-split operands in layout %d (to = %d)", lay, c.to); return VC_ERR_UNSUPPORTED;
}
#endif
