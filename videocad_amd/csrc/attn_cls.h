// attn_cls.h — the class-token attention of the LAST ViT layer (pool = 'cls': only token 0 of that layer is consumed), re-associated (r06).
//
// Restates vit-pytorch Attention.forward for the one query row that reaches the output (ctor call: reference model/trajectory_model.py:54-65;
// `x[:, 0]` pooling): with h = LayerNorm(x) [P1 tokens x D] of one frame and q = h[0] W_q (one row per head),
//     scores_j = scale * q . (h_j W_k^T) = scale * (q W_k) . h_j        =: scale * g . h_j,       g = q W_k        [H heads x D]
//     out      = sum_j p_j (h_j W_v^T)   = (sum_j p_j h_j) W_v^T        =: c W_v^T,               c = p h          [H x D]
// so the K and V projections of all P1 tokens — two thirds of the layer's QKV Linear, 215 GFLOP per launch at the benchmark shape — and their
// dgrad / wgrad twins in the backward are never formed: per frame the layer needs g (one small per-head GEMM), this kernel on (g, h), and c W_v^T.
// r05 ran, for that layer: K / V projection 237 us + single-query attention 120 us forward; single-query backward 220 us + zero-fill 25 us + the
// full QKV wgrad 345 us + dgrad 326 us backward.  Algebraically exact; in the 16-bit modes the rounding points move (g and c are rounded where k and v
// were, probabilities are rounded to 16 bits on their way into the matrix cores — as in the full layers' kernels) — the golden / full-size gates of tests/ decide.
//
// Backward of the same map, per frame and head (dc = dout W_v arrives from a per-head GEMM):
//     dp~_j = dc . h_j;  dp_j = m_j dp~_j;  Delta = sum_j p_j dp_j;  ds_j = scale p_j (dp_j - Delta)          (m_j: dropout keep-multiplier)
//     dg = sum_j ds_j h_j              ( -> dq = dg W_k^T, dW_k += q^T dg: per-head GEMMs)
//     dh_j = sum_heads ds_j g + p~_j dc          (p~ = m p; the class row additionally receives dq W_q in the caller)
//
// Kernels: attn_cls_kernels.h (one workgroup per frame, the frame's tokens staged once in LDS, every product on the matrix cores).
// 16-bit storage only (the fp32 / bf16x3 parity modes keep the projected-K/V form).  Limits: D = 512, P1 <= 64, H <= 16 (checked by the launcher).
#pragma once
#include "vc_rt.h"

struct ClsAttnParams {
    const void* ha; long ld_ha;       // [N * P1][D] normalised tokens (16-bit), frame n at rows n * P1
    const void* g;                    // [N][H][D] 16-bit
    void* c;                          // forward out: [N][H][D] 16-bit
    float* lse;                       // [N][H]: log-sum-exp of the scaled scores (forward writes, backward reads)
    const void* dc;                   // backward in: [N][H][D] 16-bit
    void* dg;                         // backward out: [N][H][D] 16-bit
    void* dha; long ld_dha;           // backward out: [N * P1][D] 16-bit
    float* r0;                        // backward out (optional): [N][D] fp32 — row 0 of dha before rounding (the caller adds the query path to it)
    int N, H, P1; float scale;
    vc_drop drop;                     // attention-probability dropout, idx = (n * H + h) * P1 + j — the single-query kernels' indexing (attn.h)
};

// LDS plan, in 16-bit elements unless noted.  Row strides are chosen so that the 16-byte fragment reads of eight consecutive rows and the 8-byte
// transpose reads of four consecutive rows fall into different banks: 528 elements = 264 dwords = 8 (mod 64); 72 -> 36 dwords; 40 -> 20 dwords.
constexpr int CA_D = 512, CA_HS = CA_D + 16, CA_PS = 72, CA_TS = 40, CA_SS = 65 /* fp32 score rows */;
VC_HD size_t cls_attn_fwd_lds(int P1) { return (size_t)P1 * CA_HS * 2 + (size_t)16 * CA_HS * 2 + (size_t)16 * CA_SS * 4 + (size_t)32 * CA_PS * 2; }
VC_HD size_t cls_attn_bwd_lds(int P1) { return (size_t)P1 * CA_HS * 2 + (size_t)32 * CA_HS * 2 + (size_t)2 * 16 * CA_SS * 4 + (size_t)32 * CA_PS * 2 + (size_t)64 * CA_TS * 2; }
