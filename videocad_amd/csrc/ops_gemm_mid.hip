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
    VC_LAUNCH((gemm_mid_kernel<TO, TRB, BM, BN>), dim3(tiles), dim3(GM_THREADS), TL::LDS_BYTES, s, c.p);
    return VC_OK;
}

// tile shape per layout: k-contiguous B -> 128 x 64, row-contiguous B (ds_read_b64_tr_b16 image needs >= 128 columns) -> 64 x 128
int vc_gemm_mid_tile_n(int trb) { return trb ? 128 : 64; }
int vc_gemm_mid_tile_m(int trb) { return trb ? 64 : 128; }
int vc_gemm_mid_launch(GemmCall c, vc_stream_t s) {
    c.p.partial = nullptr; c.p.k_per_split = c.p.K;
    if (!c.trb) return c.to == VC_F32 ? gemm_launch_mid<float, false, 128, 64>(c, s) : gemm_launch_mid<vc_bf16, false, 128, 64>(c, s);
    return c.to == VC_F32 ? gemm_launch_mid<float, true, 64, 128>(c, s) : gemm_launch_mid<vc_bf16, true, 64, 128>(c, s);
}
