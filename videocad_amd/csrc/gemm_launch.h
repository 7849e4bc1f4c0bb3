// gemm_launch.h — launch glue of the register-staged GEMM (gemm.h), shared by the translation units that instantiate its kernels
// (ops_gemm_f32.hip, ops_gemm_x3.hip, ops_gemm_bf16a.hip, ops_gemm_bf16b.hip: one family each, so they compile in parallel).
#pragma once
#include "ops.h"

template <typename CT, typename SA, typename SB, typename TO, bool TRA, bool TRB, int WT>
static int gemm_launch_wt(GemmCall c, int nsplit, vc_stream_t s) {
    constexpr size_t lds = gemm_lds_bytes<CT, TRA, TRB, WT>();
    constexpr int GEMM_BM = 64 * WT, GEMM_BN = 64 * WT;
    if (c.p.k_per_split < 0) c.p.k_per_split = -c.p.k_per_split;
#ifndef VC_EMU
    static unsigned attr_set = 0;
    if (!(attr_set & vc_device_bit())) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<CT, SA, SB, TO, TRA, TRB, WT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { vc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return VC_ERR_LAUNCH; }
        attr_set |= vc_device_bit();
    }
#endif
    ProfScope ps(c.role ? c.role - 1 : (TRA ? VC_CAT_GEMM_WGRAD : (TRB ? VC_CAT_GEMM_DGRAD : VC_CAT_GEMM_FWD)), 2.0 * c.p.M * c.p.N * c.p.K,
                 (double)c.p.M * c.p.K * sizeof(SA) + (double)c.p.N * c.p.K * sizeof(SB) + (double)c.p.M * c.p.N * sizeof(TO), s, VC_TAG_GEMM_REG);
    dim3 grid(VC_CEIL_DIV(c.p.N, GEMM_BN), VC_CEIL_DIV(c.p.M, GEMM_BM), nsplit);
    VC_LAUNCH((gemm_kernel<CT, SA, SB, TO, TRA, TRB, WT>), grid, dim3(GEMM_THREADS), lds, s, c.p);
    if (nsplit > 1) {
        long total = (long)c.p.M * c.p.N;
        if (c.p.vecC && c.p.N % 4 == 0) VC_LAUNCH((gemm_splitk_reduce4_kernel<TO>), dim3((unsigned)VC_CEIL_DIV(total / 4, 256)), dim3(256), 0, s, c.p, nsplit);
        else VC_LAUNCH((gemm_splitk_reduce_kernel<TO>), dim3((unsigned)VC_CEIL_DIV(total, 256)), dim3(256), 0, s, c.p, nsplit);
    }
    return VC_OK;
}

template <typename CT, typename SA, typename SB, typename TO, bool TRA, bool TRB>
static int gemm_launch(GemmCall c, int nsplit, vc_stream_t s) {
    return c.p.k_per_split < 0 ? gemm_launch_wt<CT, SA, SB, TO, TRA, TRB, 1>(c, nsplit, s) : gemm_launch_wt<CT, SA, SB, TO, TRA, TRB, 2>(c, nsplit, s);
}

