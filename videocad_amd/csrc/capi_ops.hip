// capi_ops.hip — single-op C entry points (include/vcad.h "vcad_op_*"): thin wrappers over the same launchers
// the engine uses, so the parity tests exercise exactly the shipped kernels.
#include "ops.h"
#include "gemm_mx8.h"
#include "../../include/vcad.h"
#include <string.h>

extern "C" {

int vcad_op_gemm(int ct, int sa, int sb, int to, int tra, int trb, const void* A, const void* B, void* C,
                 int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, const float* bias, int act,
                 const float* residual, int64_t ldr, float alpha, float* scratch, size_t scratch_bytes, uint32_t flags, int* kernel_out, void* stream) {
    GemmCall c; memset(&c, 0, sizeof(c));
    c.flags = flags; c.kernel_out = kernel_out;
    if ((flags & VC_GF_DYNAMIC) && scratch && scratch_bytes >= 256) {       // dynamic item claiming: 64 zeroed bytes at the head of the scratch buffer
        if (int rc = vc_memset_async(scratch, 0, 64, (vc_stream_t)stream)) return rc;
        c.claim = (int*)scratch; scratch += 64; scratch_bytes -= 256;
    }
    c.ct = ct; c.sa = sa; c.sb = sb; c.to = to; c.tra = tra; c.trb = trb;
    c.p.A = A; c.p.B = B; c.p.C = C; c.p.M = M; c.p.N = N; c.p.K = K; c.p.lda = lda; c.p.ldb = ldb; c.p.ldc = ldc;
    c.p.bias = bias; c.p.act = act; c.p.residual = residual; c.p.ldr = ldr; c.p.alpha = alpha; c.p.rowadd_div = 1;
    int rc = vc_gemm(c, scratch, scratch_bytes, (vc_stream_t)stream);
    if (!rc && vc_last_launch_error()) { vc_set_error("vcad_op_gemm: launch failed"); return VC_ERR_LAUNCH; }
    return rc;
}

// n (<= 4) weight gradients dW_i[N_i, K_i] = dY_i[tok, N_i]^T X_i[tok, K_i] over the same tok rows in ONE launch of the persistent kernel (16-bit operands, compact rows)
int vcad_op_wgrad_batched(int n, const void* const* dY, const void* const* X, float* const* dW, const int* N, const int* K, int tok,
                          float* scratch, size_t scratch_bytes, uint32_t flags, void* stream) {
    if (n < 1 || n > 4) { vc_set_error("vcad_op_wgrad_batched: 1..4 problems"); return VC_ERR_ARG; }
    GemmCall calls[4];
    int* claim = nullptr;
    if ((flags & VC_GF_DYNAMIC) && scratch && scratch_bytes >= 256) {
        if (int rc = vc_memset_async(scratch, 0, 64, (vc_stream_t)stream)) return rc;
        claim = (int*)scratch; scratch += 64; scratch_bytes -= 256;
    }
    for (int i = 0; i < n; ++i) {
        GemmCall& c = calls[i]; memset(&c, 0, sizeof(c));
        c.flags = flags; c.claim = claim; c.ct = VC_BF16; c.sa = VC_BF16; c.sb = VC_BF16; c.to = VC_F32; c.tra = 1; c.trb = 1;
        c.p.A = dY[i]; c.p.B = X[i]; c.p.C = dW[i]; c.p.M = N[i]; c.p.N = K[i]; c.p.K = tok; c.p.lda = N[i]; c.p.ldb = K[i]; c.p.ldc = K[i]; c.p.alpha = 1.0f; c.p.rowadd_div = 1;
    }
    int rc = vc_gemm_dma_wgrad_batched(calls, n, scratch, scratch_bytes, (vc_stream_t)stream);
    if (!rc && vc_last_launch_error()) { vc_set_error("vcad_op_wgrad_batched: launch failed"); return VC_ERR_LAUNCH; }
    return rc;
}

// fp32 -> pre-split bf16x3 operand words (VCAD_PK storage: what bf16x3 engines keep as their weight shadow)
int vcad_op_pack_x3(const float* x, void* y, int64_t n, void* stream) {
    int rc = vc_pack_x3(x, (uint32_t*)y, n, (vc_stream_t)stream);
    if (!rc && vc_last_launch_error()) { vc_set_error("vcad_op_pack_x3: launch failed"); return VC_ERR_LAUNCH; }
    return rc;
}
int vcad_op_quant_mx8(int tx, const void* x, int64_t ldx, void* q, void* scales, int64_t rows, int cols, void* stream) {
    int rc = vc_mx8_quant(tx, x, ldx, (uint8_t*)q, (uint8_t*)scales, rows, cols, (vc_stream_t)stream);
    if (!rc && vc_last_launch_error()) { vc_set_error("vcad_op_quant_mx8: launch failed"); return VC_ERR_LAUNCH; }
    return rc;
}
int vcad_op_gemm_mx8(int to, const void* A8, const void* sa, const void* B8, const void* sb, void* C, int M, int N, int K, int64_t ldc,
                     const float* bias, int act, const float* residual, int64_t ldr, void* stream) {
    Mx8Params q; memset(&q, 0, sizeof(q));
    q.g.A = A8; q.g.B = B8; q.g.C = C; q.g.M = M; q.g.N = N; q.g.K = K; q.g.lda = K; q.g.ldb = K; q.g.ldc = ldc; q.g.alpha = 1.0f;
    q.g.bias = bias; q.g.act = act; q.g.residual = residual; q.g.ldr = ldr; q.g.rowadd_div = 1;
    q.sa = (const uint8_t*)sa; q.ldsa = K / 32; q.sb = (const uint8_t*)sb; q.ldsb = K / 32;
    int rc = vc_gemm_mx8(q, to, (vc_stream_t)stream);
    if (!rc && vc_last_launch_error()) { vc_set_error("vcad_op_gemm_mx8: launch failed"); return VC_ERR_LAUNCH; }
    return rc;
}

int vcad_op_layernorm_fwd(int tx, int ty, int C, const void* x, int64_t ldx, const float* gamma, const float* beta,
                          float* y32, void* yt, float* stats, int64_t rows, float eps, void* stream) {
    LnFwdParams p; memset(&p, 0, sizeof(p));
    p.x = x; p.ldx = ldx; p.gamma = gamma; p.beta = beta; p.y32 = y32; p.ldy32 = C; p.yt = yt; p.ldyt = C; p.stats = stats;
    p.rows = rows; p.eps = eps;
    return vc_ln_fwd(tx, ty, C, 0, p, (vc_stream_t)stream);
}

int vcad_op_layernorm_bwd(int td, int ty, int C, const void* dy, const float* x, int64_t ldx, const float* stats,
                          const float* gamma, const float* add_in, float* dx32, void* dxt, float* dgamma, float* dbeta,
                          int64_t rows, float* scratch, size_t scratch_bytes, void* stream) {
    const size_t part = (size_t)vc_ln_bwd_blocks(rows) * 2 * C * 4;
    const size_t cs = (size_t)vc_colsum_chunks(vc_ln_bwd_blocks(rows)) * 2 * C * 4;
    if (part + cs + 512 > scratch_bytes) { vc_set_error("layernorm_bwd: scratch %zu < %zu", scratch_bytes, part + cs + 512); return VC_ERR_WORKSPACE; }
    LnBwdParams p; memset(&p, 0, sizeof(p));
    p.dy = dy; p.lddy = C; p.x = x; p.ldx = ldx; p.stats = stats; p.gamma = gamma; p.add_in = add_in; p.ldadd = C;
    p.dx32 = dx32; p.lddx32 = C; p.dxt = dxt; p.lddxt = C; p.rows = rows;
    float* colws = (float*)((char*)scratch + ((part + 255) & ~(size_t)255));
    return vc_ln_bwd(td, VC_F32, ty, C, 0, p, scratch, dgamma, dbeta, colws, (vc_stream_t)stream);
}

int vcad_op_attention_fwd(int t, int D, const void* q, const void* k, const void* v, void* o, int64_t ldq, int64_t ldk,
                          int64_t ldv, int64_t ldo, float* lse, int B, int H, int Tq, int Tk, int window, int causal,
                          float scale, void* stream) {
    AttnParams p; memset(&p, 0, sizeof(p));
    p.q = q; p.k = k; p.v = v; p.o = o; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lse = lse;
    p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.window = window; p.causal = causal; p.scale = scale;
    if (t == VC_X3) { p.x3 = 1; t = VC_F32; }          // fp32 tensors, bf16x3 arithmetic where a kernel has that form (attn_x3.h)
    else if (t == VC_PK) { p.x3 = 2; t = VC_F32; }     // ... and every tensor as pre-split hi | lo words
    return vc_attn_fwd(t, D, p, (vc_stream_t)stream);
}

int vcad_op_attention_bwd(int t, int D, const void* q, const void* k, const void* v, const void* dout, int64_t ldq,
                          int64_t ldk, int64_t ldv, int64_t lddo, const float* lse, float* delta, void* dq, void* dk,
                          void* dv, int64_t lddq, int64_t lddk, int64_t lddv, int B, int H, int Tq, int Tk, int window,
                          int causal, float scale, void* stream) {
    AttnParams p; memset(&p, 0, sizeof(p));
    p.q = q; p.k = k; p.v = v; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.lse = (float*)lse; p.delta = delta;
    p.dout = dout; p.lddo = lddo; p.dq = dq; p.dk = dk; p.dv = dv; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.window = window; p.causal = causal; p.scale = scale;
    if (t == VC_X3) { p.x3 = 1; t = VC_F32; }          // fp32 tensors, bf16x3 arithmetic where a kernel has that form (attn_x3.h)
    else if (t == VC_PK) { p.x3 = 2; t = VC_F32; }     // ... and every tensor as pre-split hi | lo words
    return vc_attn_bwd(t, D, p, (vc_stream_t)stream);
}

// same, with the saved forward output o (lets the long-sequence decoder kernels take D_i = rowsum(dO * O); see attn_mfma.h)
int vcad_op_attention_bwd_o(int t, int D, const void* q, const void* k, const void* v, const void* o, int64_t ldo, const void* dout, int64_t ldq,
                            int64_t ldk, int64_t ldv, int64_t lddo, const float* lse, float* delta, void* dq, void* dk,
                            void* dv, int64_t lddq, int64_t lddk, int64_t lddv, int B, int H, int Tq, int Tk, int window,
                            int causal, float scale, void* stream) {
    AttnParams p; memset(&p, 0, sizeof(p));
    p.q = q; p.k = k; p.v = v; p.o = (void*)o; p.ldo = ldo; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.lse = (float*)lse; p.delta = delta;
    p.dout = dout; p.lddo = lddo; p.dq = dq; p.dk = dk; p.dv = dv; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.window = window; p.causal = causal; p.scale = scale;
    if (t == VC_X3) { p.x3 = 1; t = VC_F32; }          // fp32 tensors, bf16x3 arithmetic where a kernel has that form (attn_x3.h)
    else if (t == VC_PK) { p.x3 = 2; t = VC_F32; }     // ... and every tensor as pre-split hi | lo words
    return vc_attn_bwd(t, D, p, (vc_stream_t)stream);
}

// class-token attention of the last ViT layer in its re-associated form (attn_cls.h): c = softmax(scale g . ha^T) ha per frame and head (16-bit storage)
int vcad_op_cls_attention_fwd(const void* ha, int64_t ld_ha, const void* g, void* c, float* lse, int N, int H, int P1, float scale, void* stream) {
    ClsAttnParams p; memset(&p, 0, sizeof(p));
    p.ha = ha; p.ld_ha = ld_ha; p.g = g; p.c = c; p.lse = lse; p.N = N; p.H = H; p.P1 = P1; p.scale = scale;
    int rc = vc_cls_attn_fwd(p, (vc_stream_t)stream);
    if (!rc && vc_last_launch_error()) { vc_set_error("vcad_op_cls_attention_fwd: launch failed"); return VC_ERR_LAUNCH; }
    return rc;
}
int vcad_op_cls_attention_bwd(const void* ha, int64_t ld_ha, const void* g, const void* dc, const float* lse, void* dg, void* dha, int64_t ld_dha, float* r0,
                              int N, int H, int P1, float scale, void* stream) {
    ClsAttnParams p; memset(&p, 0, sizeof(p));
    p.ha = ha; p.ld_ha = ld_ha; p.g = g; p.dc = dc; p.lse = (float*)lse; p.dg = dg; p.dha = dha; p.ld_dha = ld_dha; p.r0 = r0;
    p.N = N; p.H = H; p.P1 = P1; p.scale = scale;
    int rc = vc_cls_attn_bwd(p, (vc_stream_t)stream);
    if (!rc && vc_last_launch_error()) { vc_set_error("vcad_op_cls_attention_bwd: launch failed"); return VC_ERR_LAUNCH; }
    return rc;
}

}  // extern "C"
