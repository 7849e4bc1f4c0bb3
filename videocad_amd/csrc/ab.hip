// ab.hip — state and setters of the A/B build (see ab.h); compiled with -DVCAD_AB only, never into libvcad_hip.so.
#include "ops.h"
#ifdef VCAD_AB
VcAb g_ab = {0, 0, 0, -1, 0, 8, 0, 1, 0, 0, 1, 1, 1, 1, 1, 0, 1, 0, 0u};
static void setf(unsigned clear, unsigned set) { g_ab.gemm_flags = (g_ab.gemm_flags & ~clear) | set; }
extern "C" {
void vcad_debug_gemm_policy(int bits) { g_ab.policy = bits; }
void vcad_debug_gemm_stagger(int n) { g_ab.stagger = n > 0 ? n : 0; }
void vcad_debug_gemm_skip(int mask) { g_ab.skip = mask; }
void vcad_debug_gemm_epilogue(int m) { g_ab.epilogue = m; }
void vcad_debug_gemm_variant(int v) { g_ab.variant = v ? 1 : 0; }
void vcad_debug_gemm_waves(int n) { g_ab.waves = n == 4 ? 4 : 8; }
void vcad_debug_attn_variant(int v) { g_ab.attn_variant = v; }
void vcad_debug_split_gelu(int on) { g_ab.split_gelu = on ? 1 : 0; }
void vcad_debug_no_side_stream(int on) { g_ab.no_side = on ? 1 : 0; }
void vcad_debug_res_in_ln(int on) { g_ab.res_in_ln = on ? 1 : 0; }
void vcad_debug_wgrad_bk32(int on) { g_ab.wgrad_bk32 = on ? 1 : 0; }
void vcad_debug_cls_path(int on) { g_ab.cls_path = on ? 1 : 0; }
void vcad_debug_frame_first(int on) { g_ab.frame_first = on ? 1 : 0; }
void vcad_debug_pe_fold(int on) { g_ab.pe_fold = on ? 1 : 0; }
void vcad_debug_dec_h16(int on) { g_ab.dec_h16 = on ? 1 : 0; }
void vcad_debug_splitk_r06(int on) { g_ab.splitk_r06 = on ? 1 : 0; }
void vcad_debug_batch_wgrad(int on) { g_ab.batch_wg = on ? 1 : 0; }
void vcad_debug_attn_prefetch(int frames) { g_ab.attn_pf = frames > 0 ? frames : 0; }
void vcad_debug_force_gemm_tile(int tile) { setf(VC_GF_TILE64 | VC_GF_TILE128, tile == 64 ? VC_GF_TILE64 : (tile == 128 ? VC_GF_TILE128 : 0u)); }
void vcad_debug_gemm_dma(int mode) { setf(VC_GF_DMA_NEVER | VC_GF_DMA_ALWAYS, mode == 0 ? VC_GF_DMA_NEVER : (mode == 1 ? VC_GF_DMA_ALWAYS : 0u)); }
void vcad_debug_gemm_wide(int mode) { setf(VC_GF_WIDE_NEVER | VC_GF_WIDE_ALWAYS, mode == 0 ? VC_GF_WIDE_NEVER : (mode == 1 ? VC_GF_WIDE_ALWAYS : 0u)); }
void vcad_debug_gemm_mid(int mode) { setf(VC_GF_MID_NEVER | VC_GF_MID_ALWAYS, mode == 0 ? VC_GF_MID_NEVER : (mode == 1 ? VC_GF_MID_ALWAYS : 0u)); }
// CU hog for the occupied-CU A/B of the dynamic item claiming (tools/gemm_hog_ab.py): n workgroups of 1024 threads with 160 KiB of LDS each — one per
// CU, nothing else fits beside it — spinning for `us` microseconds on `stream` (the caller launches GEMMs on another stream meanwhile)
__global__ __launch_bounds__(1024) void hog_kernel(long long ticks, int* sink) {
    extern __shared__ int hog_lds[];
    hog_lds[threadIdx.x] = threadIdx.x;
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();          // 100 MHz
    long long t = t0;
    while (t - t0 < ticks) { for (int i = 0; i < 16; ++i) __builtin_amdgcn_s_sleep(32); t = (long long)__builtin_amdgcn_s_memrealtime(); }
    if (hog_lds[(threadIdx.x + 1) & 1023] == -1) sink[0] = 1;
}
int vcad_debug_hog(int n_wg, int us, int* sink, void* stream) {
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1; attr = true; }
    hipLaunchKernelGGL(hog_kernel, dim3(n_wg), dim3(1024), 160 * 1024, (hipStream_t)stream, (long long)us * 100, sink);     // s_memrealtime / cycle counter: 100 MHz
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
void vcad_debug_gemm_xcd_cols(int xn) { setf(15u << VC_GF_XCD_COLS_SHIFT, (unsigned)(xn == 0 ? 1 : ((xn == 2 || xn == 4 || xn == 8) ? xn : 0)) << VC_GF_XCD_COLS_SHIFT); }
}
#endif
