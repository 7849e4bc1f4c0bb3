// ops_gemm_mid.hip — instantiations and launch of the six-stage DMA-ring GEMM for mid-size problems (gemm_mid.h); own translation unit.
#include "ops.h"
#include "gemm_mid.h"


template <typename TO, bool TRB, int BM, int BN>
static int gemm_launch_mid(GemmCall c, vc_stream_t s) {
    using TL = GmTile<BM, BN>;
#ifndef VC_EMU
    static unsigned attr_set = 0;
    if (!(attr_set & vc_device_bit())) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_mid_kernel<TO, TRB, BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TL::LDS_BYTES);
        if (e != hipSuccess) { vc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return VC_ERR_LAUNCH; }
        attr_set |= vc_device_bit();
    }
#endif
    ProfScope ps(c.role ? c.role - 1 : (TRB ? VC_CAT_GEMM_DGRAD : VC_CAT_GEMM_FWD), 2.0 * c.p.M * c.p.N * c.p.K,
                 (double)c.p.M * c.p.K * 2 + (double)c.p.N * c.p.K * 2 + (double)c.p.M * c.p.N * sizeof(TO), s, VC_TAG_GEMM_MID);
    const int tiles = VC_CEIL_DIV(c.p.M, BM) * (c.p.N / BN);
    // a reduction shorter than the ring (K < 6 x 64: the per-head K = 64 projections of the class-token attention) only ever touches its first stages — asking
    // for those (or the epilogue's fp32 tile, whichever is larger) instead of all 144 KiB lets several workgroups share a CU (r06)
    const int nt = c.p.K / GM_BK;
    size_t lds = (size_t)(nt < GM_STAGES ? nt : GM_STAGES) * TL::STAGE_ELEMS * 2;
    if (lds < (size_t)BM * TL::ES * 4) lds = (size_t)BM * TL::ES * 4;
    VC_LAUNCH((gemm_mid_kernel<TO, TRB, BM, BN>), dim3(tiles, c.p.batch > 1 ? c.p.batch : 1), dim3(GM_THREADS), lds, s, c.p);
    return VC_OK;
}

// tile shape per layout: k-contiguous B -> 128 x 64, row-contiguous B (ds_read_b64_tr_b16 image needs >= 128 columns) -> 64 x 128
int vc_gemm_mid_tile_n(int trb) { return trb ? 128 : 64; }
int vc_gemm_mid_tile_m(int trb) { return trb ? 64 : 128; }
// `batch` problems of one shape in one grid (problem b at A + b bsa, B + b bsb, C + b bsc, element strides): all-16-bit operands, plain / bias epilogue.
// c is validated here (vc_gemm_prepare) — the caller fills the GemmCall like for vc_gemm.
int vc_gemm_mid_batched(GemmCall c, int batch, long bsa, long bsb, long bsc, vc_stream_t s) {
    if (int rc = vc_gemm_prepare(c)) return rc;
    const GemmParams& p = c.p;
    const size_t eo = c.to == VC_F32 ? 4 : 2;
    if (!(c.ct == VC_BF16 && c.sa == VC_BF16 && c.sb == VC_BF16 && !c.tra && p.vecA && p.vecB && p.vecC && p.K % 64 == 0 && p.N % vc_gemm_mid_tile_n(c.trb) == 0 &&
          batch >= 1 && batch <= 65535 && (bsa * 2) % 16 == 0 && (bsb * 2) % 16 == 0 && (bsc * (long)eo) % 16 == 0 && !p.residual && !p.aux && !p.dact_src && !p.rowadd &&
          (double)p.lda * p.M * 2 + (double)batch * bsa * 2 < 2.0e9 && (double)p.ldb * (c.trb ? p.K : p.N) * 2 + (double)batch * bsb * 2 < 4.0e9)) {
        vc_set_error("vc_gemm_mid_batched: unsupported problem (M=%d N=%d K=%d batch=%d trb=%d)", p.M, p.N, p.K, batch, c.trb); return VC_ERR_UNSUPPORTED;
    }
    c.p.batch = batch; c.p.bsa = bsa; c.p.bsb = bsb; c.p.bsc = bsc;
    if (c.kernel_out) *c.kernel_out = VC_TAG_GEMM_MID;
    return vc_gemm_mid_launch(c, s);
}
int vc_gemm_mid_launch(GemmCall c, vc_stream_t s) {
    c.p.partial = nullptr; c.p.k_per_split = c.p.K;
    if (!c.trb) return c.to == VC_F32 ? gemm_launch_mid<float, false, 128, 64>(c, s) : gemm_launch_mid<vc_bf16, false, 128, 64>(c, s);
    return c.to == VC_F32 ? gemm_launch_mid<float, true, 64, 128>(c, s) : gemm_launch_mid<vc_bf16, true, 64, 128>(c, s);
}
