// ops_attn.hip — launchers for the windowed attention kernels.
#include "ops.h"
#include "attn_mfma.h"
#include "attn_f32.h"
#include "attn_x3.h"
#include "attn_cls_kernels.h"

static int clampw(const AttnParams& p) { int mx = (p.Tq + p.qpos) > p.Tk ? (p.Tq + p.qpos) : p.Tk; return p.window > mx ? mx : (p.window < 1 ? 1 : p.window); }
static int max_keys(const AttnParams& p) { int w = clampw(p); return w < p.Tk ? w : p.Tk; }        // visible keys per query
static int max_queries(const AttnParams& p) { int w = clampw(p); return w < p.Tq ? w : p.Tq; }     // queries per key

template <typename T>
static int attn_fwd_t(int D, AttnParams p, vc_stream_t s) {
    const long waves = (long)p.B * p.H * p.Tq;
    dim3 g((unsigned)VC_CEIL_DIV(waves, 4));
    const int np = VC_CEIL_DIV(max_keys(p), 64);
    if (D == 128 && np == 1) VC_LAUNCH((attn_fwd_kernel<T, 2, 1>), g, dim3(256), 0, s, p);
    else if (D == 128 && np <= 3) VC_LAUNCH((attn_fwd_kernel<T, 2, 3>), g, dim3(256), 0, s, p);
    else if (D == 64 && np == 1) VC_LAUNCH((attn_fwd_kernel<T, 1, 1>), g, dim3(256), 0, s, p);
    else if (D == 256 && np == 1) VC_LAUNCH((attn_fwd_kernel<T, 4, 1>), g, dim3(256), 0, s, p);
    else if (D == 256 && np <= 3) VC_LAUNCH((attn_fwd_kernel<T, 4, 3>), g, dim3(256), 0, s, p);
    else if (D == 64 && np <= 3) VC_LAUNCH((attn_fwd_kernel<T, 1, 3>), g, dim3(256), 0, s, p);
    // up to 1 024 visible keys (the reference's max_ep_len is 1 000): long causal horizons of the fp32 modes and of the cached inference (r04)
    else if (D == 256 && np <= 16) VC_LAUNCH((attn_fwd_kernel<T, 4, 16>), g, dim3(256), 0, s, p);
    else if (D == 128 && np <= 16) VC_LAUNCH((attn_fwd_kernel<T, 2, 16>), g, dim3(256), 0, s, p);
    else if (D == 64 && np <= 16) VC_LAUNCH((attn_fwd_kernel<T, 1, 16>), g, dim3(256), 0, s, p);
    else { vc_set_error("attn_fwd: D=%d keys=%d unsupported", D, max_keys(p)); return VC_ERR_UNSUPPORTED; }
    return VC_OK;
}
template <typename T>
static int attn_bwd_t(int D, AttnParams p, vc_stream_t s) {
    dim3 gq((unsigned)VC_CEIL_DIV((long)p.B * p.H * p.Tq, 4)), gk((unsigned)VC_CEIL_DIV((long)p.B * p.H * p.Tk, 4));
    const int npk = VC_CEIL_DIV(max_keys(p), 64), npq = VC_CEIL_DIV(max_queries(p), 64);
    if (D == 128 && npk == 1) VC_LAUNCH((attn_bwd_q_kernel<T, 2, 1>), gq, dim3(256), 0, s, p);
    else if (D == 128 && npk <= 3) VC_LAUNCH((attn_bwd_q_kernel<T, 2, 3>), gq, dim3(256), 0, s, p);
    else if (D == 64 && npk == 1) VC_LAUNCH((attn_bwd_q_kernel<T, 1, 1>), gq, dim3(256), 0, s, p);
    else if (D == 256 && npk == 1) VC_LAUNCH((attn_bwd_q_kernel<T, 4, 1>), gq, dim3(256), 0, s, p);
    else if (D == 256 && npk <= 3) VC_LAUNCH((attn_bwd_q_kernel<T, 4, 3>), gq, dim3(256), 0, s, p);
    else if (D == 64 && npk <= 3) VC_LAUNCH((attn_bwd_q_kernel<T, 1, 3>), gq, dim3(256), 0, s, p);
    else if (D == 256 && npk <= 16) VC_LAUNCH((attn_bwd_q_kernel<T, 4, 16>), gq, dim3(256), 0, s, p);
    else if (D == 128 && npk <= 16) VC_LAUNCH((attn_bwd_q_kernel<T, 2, 16>), gq, dim3(256), 0, s, p);
    else if (D == 64 && npk <= 16) VC_LAUNCH((attn_bwd_q_kernel<T, 1, 16>), gq, dim3(256), 0, s, p);
    else { vc_set_error("attn_bwd_q: D=%d keys=%d unsupported", D, max_keys(p)); return VC_ERR_UNSUPPORTED; }
    if (D == 128 && npq == 1) VC_LAUNCH((attn_bwd_kv_kernel<T, 2, 1>), gk, dim3(256), 0, s, p);
    else if (D == 128 && npq <= 3) VC_LAUNCH((attn_bwd_kv_kernel<T, 2, 3>), gk, dim3(256), 0, s, p);
    else if (D == 64 && npq == 1) VC_LAUNCH((attn_bwd_kv_kernel<T, 1, 1>), gk, dim3(256), 0, s, p);
    else if (D == 256 && npq == 1) VC_LAUNCH((attn_bwd_kv_kernel<T, 4, 1>), gk, dim3(256), 0, s, p);
    else if (D == 256 && npq <= 3) VC_LAUNCH((attn_bwd_kv_kernel<T, 4, 3>), gk, dim3(256), 0, s, p);
    else if (D == 64 && npq <= 3) VC_LAUNCH((attn_bwd_kv_kernel<T, 1, 3>), gk, dim3(256), 0, s, p);
    else if (D == 256 && npq <= 16) VC_LAUNCH((attn_bwd_kv_kernel<T, 4, 16>), gk, dim3(256), 0, s, p);
    else if (D == 128 && npq <= 16) VC_LAUNCH((attn_bwd_kv_kernel<T, 2, 16>), gk, dim3(256), 0, s, p);
    else if (D == 64 && npq <= 16) VC_LAUNCH((attn_bwd_kv_kernel<T, 1, 16>), gk, dim3(256), 0, s, p);
    else { vc_set_error("attn_bwd_kv: D=%d queries=%d unsupported", D, max_queries(p)); return VC_ERR_UNSUPPORTED; }
    return VC_OK;
}
// algorithmic HBM bytes: forward reads q, k, v and writes o (+ the log-sum-exp); backward reads q, k, v, dO and writes dq, dk, dv
static double attn_bytes(const AttnParams& p, int D, int t, bool bwd) {
    const double es = t == VC_BF16 ? 2.0 : 4.0, hd = (double)p.B * p.H * D * es;
    return bwd ? hd * (3.0 * p.Tq + 4.0 * p.Tk) + 8.0 * p.B * p.H * p.Tq : hd * (2.0 * p.Tq + 2.0 * p.Tk) + 4.0 * p.B * p.H * p.Tq;
}
static double attn_flops(const AttnParams& p, int D) {      // 2 GEMM-like contractions over the visible keys
    return 4.0 * p.B * p.H * (double)p.Tq * max_keys(p) * D;
}
// MFMA path: bf16, 64-dim heads, <= 64 tokens, full (non-causal) attention, 16-byte-aligned head slices (the ViT)
static bool mfma_ok(int t, int D, const AttnParams& p, bool bwd) {
    auto al = [](const void* q, long ld) { return ((uintptr_t)q % 16 == 0) && (ld % 8 == 0); };
    bool ok = t == VC_BF16 && D == AM_D && p.Tq == p.Tk && p.Tq <= AM_T && !p.causal && clampw(p) >= p.Tk && !p.qpos && !p.kv_rows &&
              al(p.q, p.ldq) && al(p.k, p.ldk) && al(p.v, p.ldv);
    if (!bwd) return ok && al(p.o, p.ldo);
    return ok && al(p.dout, p.lddo) && p.lse && p.dq && p.dk && p.dv && al(p.dq, p.lddq) && al(p.dk, p.lddk) && al(p.dv, p.lddv);   // (16-byte row stores)
}
// bf16x3 mode (attn_x3.h): fp32 tensors, 64-dim heads, <= 64 tokens, full attention, 16-byte-aligned head slices (the ViT)
static bool x3_ok(int t, int D, const AttnParams& p, bool bwd) {
    auto al = [](const void* q, long ld) { return q && ((uintptr_t)q % 16 == 0) && (ld % 4 == 0); };
    bool ok = t == VC_F32 && p.x3 && D == AM_D && p.Tq == p.Tk && p.Tq <= AM_T && !p.causal && clampw(p) >= p.Tk && !p.qpos && !p.kv_rows &&
              al(p.q, p.ldq) && al(p.k, p.ldk) && al(p.v, p.ldv);
    if (!bwd) return ok && al(p.o, p.ldo);
    return ok && al(p.dout, p.lddo) && p.lse && al(p.dq, p.lddq) && al(p.dk, p.lddk) && al(p.dv, p.lddv);
}
// decoder attention on the matrix cores (attn_mfma.h): bf16, head dim 256, causal (+ window band), T <= 64
static bool dec_mfma_ok(int t, int D, const AttnParams& p, bool bwd) {
    auto al = [](const void* q, long ld) { return ((uintptr_t)q % 16 == 0) && (ld % 8 == 0); };
    bool ok = t == VC_BF16 && (D == 4 * AM_D || D == 2 * AM_D) && p.Tq == p.Tk && p.Tq <= AM_T && p.causal && !p.qpos && !p.kv_rows && al(p.q, p.ldq) && al(p.k, p.ldk) && al(p.v, p.ldv);
    if (!bwd) return ok && al(p.o, p.ldo);
    return ok && al(p.dout, p.lddo) && p.lse && p.dq && p.dk && p.dv && al(p.dq, p.lddq) && al(p.dk, p.lddk) && al(p.dv, p.lddv);   // (16-byte row stores)
}
// ... and for T > 64 (any horizon: NCH waves per 64-row block, streaming over the blocks of the other side)
static bool dec_long_ok(int t, int D, const AttnParams& p, bool bwd) {
    auto al = [](const void* q, long ld) { return ((uintptr_t)q % 16 == 0) && (ld % 8 == 0); };
    bool ok = t == VC_BF16 && (D == 4 * AM_D || D == 2 * AM_D) && p.Tq == p.Tk && p.Tq > AM_T && p.causal && !p.qpos && !p.kv_rows && al(p.q, p.ldq) && al(p.k, p.ldk) && al(p.v, p.ldv) &&
              (double)p.B * p.H * p.Tq * p.Tq < 4294967296.0;                 // (32-bit dropout indices)
    if (!bwd) return ok && al(p.o, p.ldo);
    return ok && al(p.dout, p.lddo) && p.lse && p.delta && p.dq && p.dk && p.dv && al(p.dq, p.lddq) && al(p.dk, p.lddk) && al(p.dv, p.lddv);   // (16-byte row stores)
}
// fp32 tensors on the f32 matrix cores (attn_f32.h): full / causal / banded attention with Tq == Tk <= 64, head dims 64 / 128 / 256
static bool f32_mfma_ok(int t, int D, const AttnParams& p, bool bwd) {
    auto al8 = [](const void* q, long ld) { return q && ((uintptr_t)q % 8 == 0) && (ld % 2 == 0); };
    const bool ok = t == VC_F32 && (D == 64 || D == 128 || D == 256) && p.Tq == p.Tk && p.Tq <= AF_MAXT && !p.qpos && !p.kv_rows;
    if (!bwd) return ok && al8(p.o, p.ldo);
    return ok && p.lse && p.delta && al8(p.dq, p.lddq) && al8(p.dk, p.lddk) && al8(p.dv, p.lddv);
}
template <typename K> static int set_dyn_lds(K kern, size_t bytes) {
#ifndef VC_EMU
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { vc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return VC_ERR_LAUNCH; }
#endif
    return VC_OK;
}
static int check_rows_aligned(int t, const AttnParams& p, bool bwd) {
    const long e = (t == VC_BF16) ? 8 : 4;           // elements per 16 bytes
    auto bad = [&](const void* q, long ld) { return q && (((uintptr_t)q % 16) || (ld % e)); };
    if (bad(p.q, p.ldq) || bad(p.k, p.ldk) || bad(p.v, p.ldv) || (bwd && bad(p.dout, p.lddo))) {
        vc_set_error("attention: q/k/v(/dout) head slices must be 16-byte aligned with leading dimensions multiple of %ld", e);
        return VC_ERR_ARG;
    }
    return VC_OK;
}
#define g_vit_bwd_variant VC_AB(attn_variant, 0)          // 1 = r01 kernels (A/B build only)
int vc_attn_fwd(int t, int D, AttnParams p, vc_stream_t s) {
    p.window = clampw(p);
    if (int rc = check_rows_aligned(t, p, false)) return rc;
    ProfScope ps(VC_CAT_ATTN, attn_flops(p, D), attn_bytes(p, D, t, false), s);
    if (t == VC_BF16 && p.Tq == 1 && !p.causal && p.Tk <= 64 && D == 64 && p.window >= p.Tk && !p.qpos && !p.kv_rows && g_vit_bwd_variant == 0 &&
        ((uintptr_t)p.o % 16 == 0) && (p.ldo % 8 == 0)) {         // cls-only query (last ViT layer)
        VC_LAUNCH(attn_fwd_single_query_bf16_kernel, dim3((unsigned)VC_CEIL_DIV((long)p.B * p.H, 4)), dim3(256), 0, s, p);
        return VC_OK;
    }
    if (x3_ok(t, D, p, false)) {                                   // bf16x3 mode: split operands on the bf16 matrix cores (x3 == 2: pre-split tensors in and out)
        const dim3 g((unsigned)((long)p.B * p.H));
        if (p.x3 == 2) { if (p.drop.key) VC_LAUNCH((attn_vit_fwd2_x3_kernel<true, true>), g, dim3(128), 0, s, p); else VC_LAUNCH((attn_vit_fwd2_x3_kernel<false, true>), g, dim3(128), 0, s, p); }
        else           { if (p.drop.key) VC_LAUNCH((attn_vit_fwd2_x3_kernel<true, false>), g, dim3(128), 0, s, p); else VC_LAUNCH((attn_vit_fwd2_x3_kernel<false, false>), g, dim3(128), 0, s, p); }
        return VC_OK;
    }
    if (p.x3 == 2) { vc_set_error("attention: pre-split tensors (x3 = 2) have no kernel for this shape (D=%d Tq=%d Tk=%d causal=%d)", D, p.Tq, p.Tk, p.causal); return VC_ERR_UNSUPPORTED; }
    if (mfma_ok(t, D, p, false) && g_vit_bwd_variant == 0) {      // two waves per (frame, head)
        if (p.drop.key) VC_LAUNCH((attn_vit_fwd2_kernel<true>), dim3((unsigned)((long)p.B * p.H)), dim3(128), 0, s, p);
        else VC_LAUNCH((attn_vit_fwd2_kernel<false>), dim3((unsigned)((long)p.B * p.H)), dim3(128), 0, s, p);
        return VC_OK;
    }
#ifdef VCAD_AB
    if (mfma_ok(t, D, p, false)) {
        if (p.drop.key) VC_LAUNCH((attn_vit_fwd_mfma_kernel<true>), dim3((unsigned)((long)p.B * p.H)), dim3(64), 0, s, p);
        else VC_LAUNCH((attn_vit_fwd_mfma_kernel<false>), dim3((unsigned)((long)p.B * p.H)), dim3(64), 0, s, p);
        return VC_OK;
    }
#endif
    if (dec_mfma_ok(t, D, p, false) && g_vit_bwd_variant == 0) {       // one wave per head-dim chunk
        const dim3 g((unsigned)((long)p.B * p.H));
        static unsigned attr = 0;
        if (!(attr & vc_device_bit())) {
            if (int rc = set_dyn_lds(attn_dec_fwd_cw_kernel<true, 4>, am_cw_lds_bytes(4))) return rc;
            if (int rc = set_dyn_lds(attn_dec_fwd_cw_kernel<false, 4>, am_cw_lds_bytes(4))) return rc;
            if (int rc = set_dyn_lds(attn_dec_fwd_cw_kernel<true, 2>, am_cw_lds_bytes(2))) return rc;
            if (int rc = set_dyn_lds(attn_dec_fwd_cw_kernel<false, 2>, am_cw_lds_bytes(2))) return rc;
            attr |= vc_device_bit();
        }
        if (D == 4 * AM_D) { if (p.drop.key) VC_LAUNCH((attn_dec_fwd_cw_kernel<true, 4>), g, dim3(256), am_cw_lds_bytes(4), s, p); else VC_LAUNCH((attn_dec_fwd_cw_kernel<false, 4>), g, dim3(256), am_cw_lds_bytes(4), s, p); }
        else               { if (p.drop.key) VC_LAUNCH((attn_dec_fwd_cw_kernel<true, 2>), g, dim3(128), am_cw_lds_bytes(2), s, p); else VC_LAUNCH((attn_dec_fwd_cw_kernel<false, 2>), g, dim3(128), am_cw_lds_bytes(2), s, p); }
        return VC_OK;
    }
#ifdef VCAD_AB
    if (dec_mfma_ok(t, D, p, false)) {
        const dim3 g((unsigned)((long)p.B * p.H));
        if (D == 4 * AM_D) { if (p.drop.key) VC_LAUNCH((attn_dec_fwd_mfma_kernel<true, 4>), g, dim3(64), 0, s, p); else VC_LAUNCH((attn_dec_fwd_mfma_kernel<false, 4>), g, dim3(64), 0, s, p); }
        else               { if (p.drop.key) VC_LAUNCH((attn_dec_fwd_mfma_kernel<true, 2>), g, dim3(64), 0, s, p); else VC_LAUNCH((attn_dec_fwd_mfma_kernel<false, 2>), g, dim3(64), 0, s, p); }
        return VC_OK;
    }
#endif
    if (dec_long_ok(t, D, p, false)) {
        static unsigned attr = 0;
        if (!(attr & vc_device_bit())) {
            if (int rc = set_dyn_lds(attn_dec_fwd_blk_kernel<true, 4>, am_blk_fwd_lds_bytes(4))) return rc;
            if (int rc = set_dyn_lds(attn_dec_fwd_blk_kernel<false, 4>, am_blk_fwd_lds_bytes(4))) return rc;
            if (int rc = set_dyn_lds(attn_dec_fwd_blk_kernel<true, 2>, am_blk_fwd_lds_bytes(2))) return rc;
            if (int rc = set_dyn_lds(attn_dec_fwd_blk_kernel<false, 2>, am_blk_fwd_lds_bytes(2))) return rc;
            attr |= vc_device_bit();
        }
        const dim3 g((unsigned)((long)p.B * p.H * VC_CEIL_DIV(p.Tq, AM_T)));
        if (D == 4 * AM_D) { if (p.drop.key) VC_LAUNCH((attn_dec_fwd_blk_kernel<true, 4>), g, dim3(256), am_blk_fwd_lds_bytes(4), s, p); else VC_LAUNCH((attn_dec_fwd_blk_kernel<false, 4>), g, dim3(256), am_blk_fwd_lds_bytes(4), s, p); }
        else               { if (p.drop.key) VC_LAUNCH((attn_dec_fwd_blk_kernel<true, 2>), g, dim3(128), am_blk_fwd_lds_bytes(2), s, p); else VC_LAUNCH((attn_dec_fwd_blk_kernel<false, 2>), g, dim3(128), am_blk_fwd_lds_bytes(2), s, p); }
        return VC_OK;
    }
    if (f32_mfma_ok(t, D, p, false)) {                            // one wave per 32-query block, everything on the f32 matrix cores
        const dim3 g((unsigned)VC_CEIL_DIV((long)p.B * p.H * VC_CEIL_DIV(p.Tq, 32), 4));
        const bool full = !p.causal && p.window >= p.Tk;
#define VC_AF(NCH_) do { if (full) VC_LAUNCH((attn_f32_fwd_kernel<NCH_, 2, true>), g, dim3(256), 0, s, p); else VC_LAUNCH((attn_f32_fwd_kernel<NCH_, 2, false>), g, dim3(256), 0, s, p); } while (0)
        if (D == 64) VC_AF(1); else if (D == 128) VC_AF(2); else VC_AF(4);
#undef VC_AF
        return VC_OK;
    }
    return t == VC_BF16 ? attn_fwd_t<vc_bf16>(D, p, s) : attn_fwd_t<float>(D, p, s);
}
int vc_attn_bwd(int t, int D, AttnParams p, vc_stream_t s) {
    p.window = clampw(p);
    if (int rc = check_rows_aligned(t, p, true)) return rc;
    ProfScope ps(VC_CAT_ATTN, 2.5 * attn_flops(p, D), attn_bytes(p, D, t, true), s);
    if (p.Tq == 1 && !p.causal && p.Tk <= 64 && D == 64 && p.window >= p.Tk) {     // cls-only query (last ViT layer)
        if (t == VC_BF16) {                                        // (its bf16 form stores 16-byte row pieces)
            auto bad = [](const void* q, long ld) { return ((uintptr_t)q % 16) || (ld % 8); };
            if (bad(p.dq, p.lddq) || bad(p.dk, p.lddk) || bad(p.dv, p.lddv)) { vc_set_error("attention backward (single query): dq/dk/dv head slices must be 16-byte aligned"); return VC_ERR_ARG; }
        }
        dim3 g((unsigned)VC_CEIL_DIV((long)p.B * p.H, 4));
        if (t == VC_BF16) VC_LAUNCH((attn_bwd_single_query_kernel<vc_bf16, 1>), g, dim3(256), 0, s, p);
        else VC_LAUNCH((attn_bwd_single_query_kernel<float, 1>), g, dim3(256), 0, s, p);
        return VC_OK;
    }
    if (x3_ok(t, D, p, true)) {                                    // bf16x3 mode
        static unsigned attr = 0;
        if (!(attr & vc_device_bit())) {
            if (int rc = set_dyn_lds(attn_vit_bwd4_x3_kernel<true, false>, ax_bwd_lds_bytes())) return rc;
            if (int rc = set_dyn_lds(attn_vit_bwd4_x3_kernel<false, false>, ax_bwd_lds_bytes())) return rc;
            if (int rc = set_dyn_lds(attn_vit_bwd4_x3_kernel<true, true>, ax_bwd_lds_bytes())) return rc;
            if (int rc = set_dyn_lds(attn_vit_bwd4_x3_kernel<false, true>, ax_bwd_lds_bytes())) return rc;
            attr |= vc_device_bit();
        }
        const dim3 g((unsigned)((long)p.B * p.H));
        if (p.x3 == 2) { if (p.drop.key) VC_LAUNCH((attn_vit_bwd4_x3_kernel<true, true>), g, dim3(256), ax_bwd_lds_bytes(), s, p); else VC_LAUNCH((attn_vit_bwd4_x3_kernel<false, true>), g, dim3(256), ax_bwd_lds_bytes(), s, p); }
        else           { if (p.drop.key) VC_LAUNCH((attn_vit_bwd4_x3_kernel<true, false>), g, dim3(256), ax_bwd_lds_bytes(), s, p); else VC_LAUNCH((attn_vit_bwd4_x3_kernel<false, false>), g, dim3(256), ax_bwd_lds_bytes(), s, p); }
        return VC_OK;
    }
    if (p.x3 == 2) { vc_set_error("attention backward: pre-split tensors (x3 = 2) have no kernel for this shape (D=%d Tq=%d Tk=%d causal=%d)", D, p.Tq, p.Tk, p.causal); return VC_ERR_UNSUPPORTED; }
    if (mfma_ok(t, D, p, true) && g_vit_bwd_variant == 0) {       // four waves per (frame, head)
        p.pf_frames = p.B >= 512 ? VC_AB(attn_pf, 0) : 0;           // (A/B build only: L2 warm-up experiment of r06, attn_mfma.h — measured slower)
        if (p.drop.key) VC_LAUNCH(attn_vit_bwd4_kernel_drop, dim3((unsigned)((long)p.B * p.H)), dim3(256), 0, s, p);
        else VC_LAUNCH(attn_vit_bwd4_kernel_eval, dim3((unsigned)((long)p.B * p.H)), dim3(256), 0, s, p);
        return VC_OK;
    }
#ifdef VCAD_AB
    if (mfma_ok(t, D, p, true)) {                                  // r01's two-wave kernel, kept for the A/B (vcad_debug_attn_variant(1))
        if (p.drop.key) VC_LAUNCH(attn_vit_bwd_mfma_kernel_drop, dim3((unsigned)((long)p.B * p.H)), dim3(128), 0, s, p);
        else VC_LAUNCH(attn_vit_bwd_mfma_kernel_eval, dim3((unsigned)((long)p.B * p.H)), dim3(128), 0, s, p);
        return VC_OK;
    }
#endif
    if (dec_mfma_ok(t, D, p, true) && g_vit_bwd_variant == 0) {        // one wave per (orientation, head-dim chunk)
        const dim3 g((unsigned)((long)p.B * p.H));
        static unsigned attr = 0;
        if (!(attr & vc_device_bit())) {
            if (int rc = set_dyn_lds(attn_dec_bwd_cw_kernel<true, 4>, am_cwb_lds_bytes(4))) return rc;
            if (int rc = set_dyn_lds(attn_dec_bwd_cw_kernel<false, 4>, am_cwb_lds_bytes(4))) return rc;
            if (int rc = set_dyn_lds(attn_dec_bwd_cw_kernel<true, 2>, am_cwb_lds_bytes(2))) return rc;
            if (int rc = set_dyn_lds(attn_dec_bwd_cw_kernel<false, 2>, am_cwb_lds_bytes(2))) return rc;
            attr |= vc_device_bit();
        }
        if (D == 4 * AM_D) { if (p.drop.key) VC_LAUNCH((attn_dec_bwd_cw_kernel<true, 4>), g, dim3(256), am_cwb_lds_bytes(4), s, p); else VC_LAUNCH((attn_dec_bwd_cw_kernel<false, 4>), g, dim3(256), am_cwb_lds_bytes(4), s, p); }
        else               { if (p.drop.key) VC_LAUNCH((attn_dec_bwd_cw_kernel<true, 2>), g, dim3(256), am_cwb_lds_bytes(2), s, p); else VC_LAUNCH((attn_dec_bwd_cw_kernel<false, 2>), g, dim3(256), am_cwb_lds_bytes(2), s, p); }
        return VC_OK;
    }
#ifdef VCAD_AB
    if (dec_mfma_ok(t, D, p, true)) {
        const dim3 g((unsigned)((long)p.B * p.H));
        if (D == 4 * AM_D) { if (p.drop.key) VC_LAUNCH((attn_dec_bwd_mfma_kernel<true, 4>), g, dim3(128), 0, s, p); else VC_LAUNCH((attn_dec_bwd_mfma_kernel<false, 4>), g, dim3(128), 0, s, p); }
        else               { if (p.drop.key) VC_LAUNCH((attn_dec_bwd_mfma_kernel<true, 2>), g, dim3(128), 0, s, p); else VC_LAUNCH((attn_dec_bwd_mfma_kernel<false, 2>), g, dim3(128), 0, s, p); }
        return VC_OK;
    }
#endif
    if (dec_long_ok(t, D, p, true)) {
        static unsigned attr = 0;
        if (!(attr & vc_device_bit())) {
            if (int rc = set_dyn_lds(attn_dec_bwd_q_blk_kernel<true, 4>, am_blk_bwd_lds_bytes(4))) return rc;
            if (int rc = set_dyn_lds(attn_dec_bwd_q_blk_kernel<false, 4>, am_blk_bwd_lds_bytes(4))) return rc;
            if (int rc = set_dyn_lds(attn_dec_bwd_kv_blk_kernel<true, 4>, am_blk_bwd_lds_bytes(4))) return rc;
            if (int rc = set_dyn_lds(attn_dec_bwd_kv_blk_kernel<false, 4>, am_blk_bwd_lds_bytes(4))) return rc;
            if (int rc = set_dyn_lds(attn_dec_bwd_q_blk_kernel<true, 2>, am_blk_bwd_lds_bytes(2))) return rc;
            if (int rc = set_dyn_lds(attn_dec_bwd_q_blk_kernel<false, 2>, am_blk_bwd_lds_bytes(2))) return rc;
            if (int rc = set_dyn_lds(attn_dec_bwd_kv_blk_kernel<true, 2>, am_blk_bwd_lds_bytes(2))) return rc;
            if (int rc = set_dyn_lds(attn_dec_bwd_kv_blk_kernel<false, 2>, am_blk_bwd_lds_bytes(2))) return rc;
            attr |= vc_device_bit();
        }
        const dim3 g((unsigned)((long)p.B * p.H * VC_CEIL_DIV(p.Tq, AM_T)));
#define VC_LONG_BWD(DROP_, NCH_) do { VC_LAUNCH((attn_dec_bwd_q_blk_kernel<DROP_, NCH_>), g, dim3(64 * NCH_), am_blk_bwd_lds_bytes(NCH_), s, p); \
                                      VC_LAUNCH((attn_dec_bwd_kv_blk_kernel<DROP_, NCH_>), g, dim3(64 * NCH_), am_blk_bwd_lds_bytes(NCH_), s, p); } while (0)
        if (D == 4 * AM_D) { if (p.drop.key) VC_LONG_BWD(true, 4); else VC_LONG_BWD(false, 4); }
        else               { if (p.drop.key) VC_LONG_BWD(true, 2); else VC_LONG_BWD(false, 2); }
#undef VC_LONG_BWD
        return VC_OK;
    }
    if (f32_mfma_ok(t, D, p, true)) {
        const dim3 gq((unsigned)VC_CEIL_DIV((long)p.B * p.H * VC_CEIL_DIV(p.Tq, 32), 4)), gk((unsigned)VC_CEIL_DIV((long)p.B * p.H * VC_CEIL_DIV(p.Tk, 32), 4));
        const bool full = !p.causal && p.window >= p.Tk;
#define VC_AB_(NCH_, F_) do { VC_LAUNCH((attn_f32_bwd_q_kernel<NCH_, 2, F_>), gq, dim3(256), 0, s, p); VC_LAUNCH((attn_f32_bwd_kv_kernel<NCH_, 2, F_>), gk, dim3(256), 0, s, p); } while (0)
#define VC_AB(NCH_) do { if (full) VC_AB_(NCH_, true); else VC_AB_(NCH_, false); } while (0)
        if (D == 64) VC_AB(1); else if (D == 128) VC_AB(2); else VC_AB(4);
#undef VC_AB
#undef VC_AB_
        return VC_OK;
    }
    return t == VC_BF16 ? attn_bwd_t<vc_bf16>(D, p, s) : attn_bwd_t<float>(D, p, s);
}

// ---------------------------------------------------------------------------------------------- class-token attention (attn_cls.h)
bool vc_cls_attn_ok(int D, int H, int P1, int dim_head) { return D == CA_D && H >= 1 && H <= 16 && P1 >= 1 && P1 <= 64 && dim_head == 64; }
static int cls_attn_check(const ClsAttnParams& p, bool bwd) {
    if (p.N < 1 || !vc_cls_attn_ok(CA_D, p.H, p.P1, 64)) { vc_set_error("class-token attention: N=%d H=%d P1=%d outside the kernel's limits (H <= 16, P1 <= 64)", p.N, p.H, p.P1); return VC_ERR_UNSUPPORTED; }
    auto bad = [](const void* q, long ld) { return !q || ((uintptr_t)q % 16) || (ld % 8); };
    if (bad(p.ha, p.ld_ha) || bad(p.g, CA_D) || !p.lse || (!bwd && bad(p.c, CA_D)) || (bwd && (bad(p.dc, CA_D) || bad(p.dg, CA_D) || bad(p.dha, p.ld_dha)))) {
        vc_set_error("class-token attention: tensors must be 16-byte aligned with leading dimensions multiple of 8"); return VC_ERR_ARG;
    }
    if ((double)p.N * p.H * p.P1 >= 4294967296.0) { vc_set_error("class-token attention: dropout index overflows 32 bits"); return VC_ERR_UNSUPPORTED; }
    return VC_OK;
}
int vc_cls_attn_fwd(ClsAttnParams p, vc_stream_t s) {
    if (int rc = cls_attn_check(p, false)) return rc;
    static unsigned attr = 0;
    if (!(attr & vc_device_bit())) { if (int rc = set_dyn_lds(cls_attn_fwd_kernel, cls_attn_fwd_lds(64))) return rc; attr |= vc_device_bit(); }
    const double rows = (double)p.N * p.P1, hd = (double)p.N * p.H * CA_D;
    ProfScope ps(VC_CAT_ATTN, 4.0 * p.N * p.H * p.P1 * CA_D, rows * CA_D * 2 + 2 * hd * 2, s);
    VC_LAUNCH(cls_attn_fwd_kernel, dim3((unsigned)p.N), dim3(256), cls_attn_fwd_lds(p.P1), s, p);
    return VC_OK;
}
int vc_cls_attn_bwd(ClsAttnParams p, vc_stream_t s) {
    if (int rc = cls_attn_check(p, true)) return rc;
    static unsigned attr = 0;
    if (!(attr & vc_device_bit())) { if (int rc = set_dyn_lds(cls_attn_bwd_kernel, cls_attn_bwd_lds(64))) return rc; attr |= vc_device_bit(); }
    const double rows = (double)p.N * p.P1, hd = (double)p.N * p.H * CA_D;
    ProfScope ps(VC_CAT_ATTN, 10.0 * p.N * p.H * p.P1 * CA_D, 2 * rows * CA_D * 2 + 3 * hd * 2, s);
    VC_LAUNCH(cls_attn_bwd_kernel, dim3((unsigned)p.N), dim3(512), cls_attn_bwd_lds(p.P1), s, p);
    return VC_OK;
}
