// ops_mx8.hip — launchers of the MXFP8 quantiser and GEMM (gemm_mx8.h); own translation unit.
#include "ops.h"
#include "gemm_mx8.h"

#ifdef VC_H16     // bf16-only mode: the fp16-storage build (libvcad_hip_f16.so) carries the entry point, not the kernels
int vc_mx8_quant(int, const void*, long, uint8_t*, uint8_t*, long, int, vc_stream_t) { vc_set_error("vc_mx8_quant: the fp8 forward exists in the bf16 build only"); return VC_ERR_UNSUPPORTED; }
int vc_gemm_mx8(Mx8Params, int, vc_stream_t) { vc_set_error("vc_gemm_mx8: the fp8 forward exists in the bf16 build only"); return VC_ERR_UNSUPPORTED; }
#else

int vc_mx8_quant(int tx, const void* x, long ld, uint8_t* q, uint8_t* sc, long rows, int cols, vc_stream_t s) {
    if (rows <= 0) return VC_OK;
    if (cols % 32) { vc_set_error("mx8_quant: cols %d is not a multiple of the 32-element block", cols); return VC_ERR_ARG; }
    ProfScope ps(VC_CAT_OTHER, 0, (double)rows * cols * ((tx == VC_BF16 ? 2 : 4) + 1.03), s);
    const long n = rows * (cols / 8);
    dim3 g((unsigned)VC_CEIL_DIV(n, 256));
    if (tx == VC_BF16) VC_LAUNCH((mx8_quant_kernel<vc_bf16>), g, dim3(256), 0, s, (const vc_bf16*)x, ld, q, sc, rows, cols);
    else VC_LAUNCH((mx8_quant_kernel<float>), g, dim3(256), 0, s, (const float*)x, ld, q, sc, rows, cols);
    return VC_OK;
}

template <typename TO>
static int mx8_launch(const Mx8Params& q, vc_stream_t s) {
#ifndef VC_EMU
    static unsigned attr_set = 0;
    if (!(attr_set & vc_device_bit())) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_mx8_kernel<TO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MX_LDS_BYTES);
        if (e != hipSuccess) { vc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return VC_ERR_LAUNCH; }
        attr_set |= vc_device_bit();
    }
#endif
    const int tiles = VC_CEIL_DIV(q.g.M, MX_BM) * (q.g.N / MX_BN);
    VC_LAUNCH((gemm_mx8_kernel<TO>), dim3((unsigned)tiles), dim3(MX_THREADS), MX_LDS_BYTES, s, q);
    return VC_OK;
}

// C (type `to`) = epilogue(A8 B8^T): forward layout only (both operands k-contiguous)
int vc_gemm_mx8(Mx8Params q, int to, vc_stream_t s) {
    GemmParams& p = q.g;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) { vc_set_error("vc_gemm_mx8: empty problem"); return VC_ERR_ARG; }
    if (p.N % MX_BN || p.K % MX_BK || (p.lda % 16) || (p.ldb % 16) || ((uintptr_t)p.A % 16) || ((uintptr_t)p.B % 16) || (q.ldsa % 4) || (q.ldsb % 4) ||
        ((uintptr_t)q.sa % 4) || ((uintptr_t)q.sb % 4)) {          // (scale rows are read as 32-bit words)
        vc_set_error("vc_gemm_mx8: needs N %% 128 == 0, K %% 128 == 0 and 16-byte aligned rows (N=%d K=%d)", p.N, p.K); return VC_ERR_UNSUPPORTED; }
    if ((double)p.lda * p.M >= 4.0e9 || (double)p.ldb * p.N >= 4.0e9) { vc_set_error("vc_gemm_mx8: operand larger than 4 GiB"); return VC_ERR_UNSUPPORTED; }
    if (p.act == VC_ACT_GELU) p.act = VC_ACT_GELU_FAST;           // as in the bf16 mode
    if (p.rowadd_div == 0) p.rowadd_div = 1;
    ProfScope ps(VC_CAT_GEMM_FWD, 2.0 * p.M * p.N * p.K, (double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N * (to == VC_BF16 ? 2 : 4), s);
    return to == VC_BF16 ? mx8_launch<vc_bf16>(q, s) : mx8_launch<float>(q, s);
}
#endif
