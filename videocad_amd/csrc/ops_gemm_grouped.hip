// ops_gemm_grouped.hip — grouped GEMM launch (own translation unit: the kernel instantiations compile in parallel with ops_gemm.hip)
#include "ops.h"

// ---------------------------------------------------------------------------------------------- grouped launch
// Host half: fills `probs` (device-layout descriptors) and `tile_start` (n + 1 entries, every problem padded to a multiple
// of 8 tiles so the XCD-aware order inside a problem lines up with the hardware's round-robin); all problems must share the
// first one's signature.  The caller uploads both arrays to device memory once per plan and launches every step.
int vc_gemm_grouped_prepare(GemmCall* calls, int n, GemmParams* probs, int* tile_start, int tile) {
    if (tile != 64 && tile != 128) { vc_set_error("vc_gemm_grouped: tile %d", tile); return VC_ERR_ARG; }
    int t = 0;
    for (int i = 0; i < n; ++i) {
        GemmCall& c = calls[i];
        if (c.ct != calls[0].ct || c.sa != calls[0].sa || c.sb != calls[0].sb || c.to != calls[0].to || c.tra != calls[0].tra || c.trb != calls[0].trb) {
            vc_set_error("vc_gemm_grouped: mixed signatures"); return VC_ERR_ARG;
        }
        int rc = vc_gemm_prepare(c); if (rc) return rc;
        c.p.k_per_split = VC_CEIL_DIV(c.p.K, 64) * 64; c.p.partial = nullptr; c.p.stagger = 0; c.p.debug_skip = 0;
        probs[i] = c.p; tile_start[i] = t;
        t += VC_CEIL_DIV(VC_CEIL_DIV(c.p.M, tile) * VC_CEIL_DIV(c.p.N, tile), 8) * 8;
    }
    tile_start[n] = t;
    return VC_OK;
}

template <typename CT, typename SA, typename SB, typename TO, bool TRA, bool TRB, int WT = 2>
static int grouped_launch(const GemmCall& sig, GemmGroup grp, int total_tiles, double flops, vc_stream_t s) {
    constexpr size_t lds = gemm_lds_bytes<CT, TRA, TRB, WT>();
#ifndef VC_EMU
    static unsigned attr_set = 0;
    if (!(attr_set & vc_device_bit())) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_grouped_kernel<CT, SA, SB, TO, TRA, TRB, WT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { vc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return VC_ERR_LAUNCH; }
        attr_set |= vc_device_bit();
    }
#endif
    ProfScope ps(sig.role ? sig.role - 1 : (TRA ? VC_CAT_GEMM_WGRAD : (TRB ? VC_CAT_GEMM_DGRAD : VC_CAT_GEMM_FWD)), flops, 0.0, s, VC_TAG_GEMM_GROUPED);
    VC_LAUNCH((gemm_grouped_kernel<CT, SA, SB, TO, TRA, TRB, WT>), dim3((unsigned)total_tiles), dim3(GEMM_THREADS), lds, s, grp);
    return VC_OK;
}

bool vc_gemm_grouped_has_forward() { return true; }
// Device half: probs / tile_start are DEVICE pointers holding what vc_gemm_grouped_prepare produced.  Instantiated: the wgrad layout
// (tra = trb = 1, fp32 output) — what the engine defers — and the bf16 forward layout on fp32 activations (batched K / V projections).
int vc_gemm_grouped_launch(const GemmCall& sig, const GemmParams* probs, const int* tile_start, int n, int total_tiles, double flops, vc_stream_t s, int tile) {
    if (n <= 0) return VC_OK;
    GemmGroup grp{probs, tile_start, n};
    // 64 x 64 tiles (r06): the class-token attention's K / V weight-gradient slices — 64-row outputs, which the 128-row tile stages element-wise (ragged in
    // the contiguous dimension of a transposed operand) — 16-bit operands only
    if (tile == 64) {
        if (sig.tra && sig.trb && sig.to == VC_F32 && sig.ct == VC_BF16 && sig.sa == VC_BF16 && sig.sb == VC_BF16)
            return grouped_launch<vc_bf16, vc_bf16, vc_bf16, float, true, true, 1>(sig, grp, total_tiles, flops, s);
        vc_set_error("vc_gemm_grouped: the 64 x 64 tile is instantiated for 16-bit weight gradients only"); return VC_ERR_UNSUPPORTED;
    }
    // forward layout (r04: the decoder's batched cross-attention K / V projections): fp32 activations x bf16 weights -> bf16, fused bias epilogue
    if (!sig.tra && !sig.trb && sig.ct == VC_BF16 && sig.sa == VC_F32 && sig.sb == VC_BF16 && sig.to == VC_BF16)
        return grouped_launch<vc_bf16, float, vc_bf16, vc_bf16, false, false>(sig, grp, total_tiles, flops, s);
    if (!(sig.tra && sig.trb) || sig.to != VC_F32) { vc_set_error("vc_gemm_grouped: only the wgrad layout (and the bf16 forward layout with fp32 activations) is instantiated"); return VC_ERR_UNSUPPORTED; }
    if (sig.ct == VC_F32) return grouped_launch<float, float, float, float, true, true>(sig, grp, total_tiles, flops, s);
    if (sig.ct == VC_X3) return grouped_launch<vc_x3, float, float, float, true, true>(sig, grp, total_tiles, flops, s);
    if (sig.sa == VC_BF16 && sig.sb == VC_BF16) return grouped_launch<vc_bf16, vc_bf16, vc_bf16, float, true, true>(sig, grp, total_tiles, flops, s);
    if (sig.sa == VC_BF16 && sig.sb == VC_F32) return grouped_launch<vc_bf16, vc_bf16, float, float, true, true>(sig, grp, total_tiles, flops, s);
    if (sig.sa == VC_F32 && sig.sb == VC_BF16) return grouped_launch<vc_bf16, float, vc_bf16, float, true, true>(sig, grp, total_tiles, flops, s);
    return grouped_launch<vc_bf16, float, float, float, true, true>(sig, grp, total_tiles, flops, s);
}
