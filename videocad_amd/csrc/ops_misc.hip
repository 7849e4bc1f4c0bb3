// ops_misc.hip — launchers for LayerNorm / reductions / elementwise / loss / optimizer kernels.
#include "ops.h"

template <typename TX, typename TY, int MODE>
static int ln_fwd_c(int C, LnFwdParams p, long out_rows, vc_stream_t s) {
    dim3 grid((unsigned)VC_CEIL_DIV(out_rows, 4));
    if (C == 512) VC_LAUNCH((ln_fwd_kernel<TX, TY, 8, MODE>), grid, dim3(256), 0, s, p);
    else if (C == 1024) VC_LAUNCH((ln_fwd_kernel<TX, TY, 16, MODE>), grid, dim3(256), 0, s, p);
    else { vc_set_error("ln_fwd: C=%d unsupported (512/1024)", C); return VC_ERR_UNSUPPORTED; }
    return VC_OK;
}
template <typename TX, typename TY>
static int ln_fwd_m(int C, int mode, LnFwdParams p, vc_stream_t s) {
    if (mode == 0) return ln_fwd_c<TX, TY, 0>(C, p, p.rows, s);
    if (mode == 1) return ln_fwd_c<float, TY, 1>(C, p, p.rows, s);
    if (mode == 2) return ln_fwd_c<TX, TY, 2>(C, p, p.rows, s);      // p.rows = output rows N*(P+1)
    vc_set_error("ln_fwd: bad mode %d", mode); return VC_ERR_ARG;
}
int vc_ln_fwd(int tx, int ty, int C, int mode, LnFwdParams p, vc_stream_t s) {
    if (p.rows <= 0) return VC_OK;
    if (p.add && (mode != 0 || ty == VC_PK)) { vc_set_error("ln_fwd: the fused residual add belongs to the plain mode with an fp32 / 16-bit branch tensor"); return VC_ERR_UNSUPPORTED; }
    ProfScope ps(VC_CAT_NORM, 0, (double)p.rows * C * ((tx == VC_BF16 ? 2 : 4) + (p.y32 ? 4 : 0) + (p.yt ? (ty == VC_BF16 ? 2 : 4) : 0) + (p.add ? (ty == VC_BF16 ? 2 : 4) + (p.sum32 ? 4 : 0) : 0)), s);
    if (tx == VC_F32 && ty == VC_F32) return ln_fwd_m<float, float>(C, mode, p, s);
    if (tx == VC_F32 && ty == VC_BF16) return ln_fwd_m<float, vc_bf16>(C, mode, p, s);
    if (tx == VC_BF16 && ty == VC_BF16) return ln_fwd_m<vc_bf16, vc_bf16>(C, mode, p, s);
    if (tx == VC_F32 && ty == VC_PK) return ln_fwd_m<float, vc_pk>(C, mode, p, s);          // bf16x3 mode: the typed output is a GEMM operand, written pre-split
    vc_set_error("ln_fwd: dtype combo %d %d", tx, ty); return VC_ERR_UNSUPPORTED;
}

long vc_ln_bwd_blocks(long rows) { long b = VC_CEIL_DIV(rows, 4); return b > 512 ? 512 : (b < 1 ? 1 : b); }

template <typename TD, typename TX, typename TY, int MODE>
static int ln_bwd_c(int C, LnBwdParams p, unsigned nblk, vc_stream_t s) {
    if (C == 512) VC_LAUNCH((ln_bwd_kernel<TD, TX, TY, 8, MODE>), dim3(nblk), dim3(256), 0, s, p);
    else if (C == 1024) VC_LAUNCH((ln_bwd_kernel<TD, TX, TY, 16, MODE>), dim3(nblk), dim3(256), 0, s, p);
    else { vc_set_error("ln_bwd: C=%d unsupported", C); return VC_ERR_UNSUPPORTED; }
    return VC_OK;
}
template <typename TD, typename TX, typename TY>
static int ln_bwd_m(int C, int mode, LnBwdParams p, unsigned nblk, vc_stream_t s) {
    if (mode == 0) return ln_bwd_c<TD, TX, TY, 0>(C, p, nblk, s);
    if (mode == 1) return ln_bwd_c<TD, float, TY, 1>(C, p, nblk, s);
    if (mode == 2) return ln_bwd_c<TD, TX, TY, 2>(C, p, nblk, s);
    vc_set_error("ln_bwd: bad mode %d", mode); return VC_ERR_ARG;
}
int vc_ln_bwd(int td, int tx, int ty, int C, int mode, LnBwdParams p, float* partial_ws, float* dgamma, float* dbeta,
              float* colsum_ws, vc_stream_t s, float* dsum_out) {
    if (p.rows <= 0) return VC_OK;
    const unsigned nblk = (unsigned)vc_ln_bwd_blocks(p.rows);
    p.partial = partial_ws;
    p.dsum = dsum_out && partial_ws && (p.dx32 || p.dxt);
    const long PS = (p.dsum ? 3L : 2L) * C;          // partial row stride
    ProfScope ps(VC_CAT_NORM, 0, (double)p.rows * C * ((td == VC_BF16 ? 2 : 4) + 4 + (p.add_in ? 4 : 0) + (p.dx32 ? 4 : 0)), s);
    int rc;
    if (td == VC_F32 && tx == VC_F32 && ty == VC_F32) rc = ln_bwd_m<float, float, float>(C, mode, p, nblk, s);
    else if (td == VC_BF16 && tx == VC_F32 && ty == VC_BF16) rc = ln_bwd_m<vc_bf16, float, vc_bf16>(C, mode, p, nblk, s);
    else if (td == VC_F32 && tx == VC_F32 && ty == VC_BF16) rc = ln_bwd_m<float, float, vc_bf16>(C, mode, p, nblk, s);
    else if (td == VC_F32 && tx == VC_F32 && ty == VC_PK) rc = ln_bwd_m<float, float, vc_pk>(C, mode, p, nblk, s);
    else { vc_set_error("ln_bwd: dtype combo %d %d %d", td, tx, ty); return VC_ERR_UNSUPPORTED; }
    if (rc) return rc;
    if (partial_ws && dgamma) {        // partial is [nblk][2][C]: column-sum it into dgamma (first C) / dbeta (next C); dgamma == null: the caller reduces the rows later
        // partial is [nblk][2 or 3][C] (nblk <= 512): dgamma, dbeta and the emitted gradient's column sums in ONE launch (r02: two or three)
        (void)colsum_ws;
        rc = vc_colsum_seg(partial_ws, PS, nblk, C, p.dsum ? 3 : 2, dgamma, dbeta, dsum_out, s); if (rc) return rc;
    }
    return VC_OK;
}

long vc_colsum_chunks(long rows) { return VC_CEIL_DIV(rows, 128) + VC_CEIL_DIV(VC_CEIL_DIV(rows, 128), 128) + 2; }   // partial rows, both ping-pong levels

// one launch for up to three equally wide column segments of an fp32 partial matrix with <= COLSUM_ROWS rows (the LayerNorm backward's partial rows)
int vc_colsum_seg(const float* x, long ld, long rows, int seg, int nseg, float* out0, float* out1, float* out2, vc_stream_t s) {
    if (rows <= 0 || seg <= 0) return VC_OK;
    if (rows > COLSUM_ROWS || nseg < 1 || nseg > 3) { vc_set_error("vc_colsum_seg: %ld rows / %d segments", rows, nseg); return VC_ERR_ARG; }
    ProfScope ps(VC_CAT_OTHER, 0, (double)rows * seg * nseg * 4, s);
    ColsumParams p = ColsumParams();
    p.x = x; p.ld = ld; p.rows = rows; p.cols = seg * nseg; p.rows_per_block = COLSUM_ROWS; p.out = out0; p.out1 = out1; p.out2 = out2; p.seg = seg;
    VC_LAUNCH((colsum_pass_kernel<float>), dim3(VC_CEIL_DIV(seg * nseg, 64), 1, 1), dim3(256), 0, s, p);
    return VC_OK;
}
int vc_colsum(int tx, const void* x, long ld, long rows, int cols, float* out, int accumulate,
              int batch, long bstride_x, long bstride_out, float* ws, vc_stream_t s) {
    if (rows <= 0 || cols <= 0) return VC_OK;
    ProfScope ps(VC_CAT_OTHER, 0, (double)batch * rows * cols * (tx == VC_BF16 ? 2 : 4), s);
    const void* cur = x; long cur_ld = ld, cur_rows = rows, cur_bs = bstride_x; int cur_t = tx;
    float* wsA = ws; float* wsB = ws + (long)batch * VC_CEIL_DIV(rows, 128) * cols;
    bool useA = true;
    while (true) {
        const bool last = cur_rows <= COLSUM_ROWS;
        const long nblk = last ? 1 : VC_CEIL_DIV(cur_rows, COLSUM_ROWS);
        ColsumParams p; p.seg = 0; p.out1 = p.out2 = nullptr;
        p.x = cur; p.ld = cur_ld; p.rows = cur_rows; p.cols = cols; p.batch_stride_x = cur_bs; p.rows_per_block = COLSUM_ROWS;
        float* dst = last ? out : (useA ? wsA : wsB);
        p.out = dst; p.ld_out_rows = last ? 0 : cols; p.batch_stride_out = last ? bstride_out : nblk * cols; p.accumulate = last ? accumulate : 0;
        dim3 g(VC_CEIL_DIV(cols, 64), (unsigned)nblk, batch);
        if (cur_t == VC_F32) VC_LAUNCH((colsum_pass_kernel<float>), g, dim3(256), 0, s, p);
        else if (cur_t == VC_PK) VC_LAUNCH((colsum_pass_kernel<vc_pk>), g, dim3(256), 0, s, p);          // pre-split words: hi + lo
        else VC_LAUNCH((colsum_pass_kernel<vc_bf16>), g, dim3(256), 0, s, p);
        if (last) break;
        cur = dst; cur_ld = cols; cur_rows = nblk; cur_bs = nblk * cols; cur_t = VC_F32; useA = !useA;
    }
    return VC_OK;
}

int vc_dropout_mul(int ty, const float* in, long ld_in, void* out, long ld_out, long rows, int cols, vc_drop d, vc_stream_t s) {
    if (rows <= 0) return VC_OK;
    if (cols % 4) { vc_set_error("dropout_mul: cols %% 4 != 0"); return VC_ERR_ARG; }
    ProfScope ps(VC_CAT_OTHER, 0, (double)rows * cols * (4 + (ty == VC_BF16 ? 2 : 4)), s);
    dim3 g((unsigned)VC_CEIL_DIV(rows * cols / 4, 256));
    if (ty == VC_BF16) VC_LAUNCH((dropout_mul_kernel<vc_bf16>), g, dim3(256), 0, s, in, ld_in, (vc_bf16*)out, ld_out, rows, cols, d);
    else if (ty == VC_PK) VC_LAUNCH((dropout_mul_kernel<vc_pk>), g, dim3(256), 0, s, in, ld_in, (vc_pk*)out, ld_out, rows, cols, d);
    else VC_LAUNCH((dropout_mul_kernel<float>), g, dim3(256), 0, s, in, ld_in, (float*)out, ld_out, rows, cols, d);
    return VC_OK;
}
int vc_zero_cols(void* p, long ld_bytes, long rows, long width_bytes, vc_stream_t s) {
    if (rows <= 0 || width_bytes <= 0) return VC_OK;
    if (((uintptr_t)p | (uintptr_t)ld_bytes | (uintptr_t)width_bytes) & 15) { vc_set_error("vc_zero_cols: 16-byte alignment"); return VC_ERR_ARG; }
    ProfScope ps(VC_CAT_OTHER, 0, (double)rows * width_bytes, s);
    VC_LAUNCH(zero_cols_kernel, dim3((unsigned)VC_CEIL_DIV(rows * (width_bytes / 16), 256)), dim3(256), 0, s, (char*)p, ld_bytes, rows, (int)width_bytes);
    return VC_OK;
}
int vc_dtanh(int ty, const float* d, const float* y, float* out32, void* outt, long n, vc_stream_t s) {
    dim3 g((unsigned)VC_CEIL_DIV(n, 256));
    if (ty == VC_BF16) VC_LAUNCH((dtanh_kernel<vc_bf16>), g, dim3(256), 0, s, d, y, out32, (vc_bf16*)outt, n);
    else VC_LAUNCH((dtanh_kernel<float>), g, dim3(256), 0, s, d, y, out32, (float*)outt, n);
    return VC_OK;
}
int vc_embed_action(int ty, const float* a, const float* W, const float* b, const float* ts, float* y32, void* yt,
                    long M, int H, int K, int T, vc_stream_t s) {
    dim3 g((unsigned)VC_CEIL_DIV(M * H, 256));
    if (ty == VC_BF16) VC_LAUNCH((embed_action_kernel<vc_bf16>), g, dim3(256), 0, s, a, W, b, ts, y32, (vc_bf16*)yt, M, H, K, T);
    else VC_LAUNCH((embed_action_kernel<float>), g, dim3(256), 0, s, a, W, b, ts, y32, (float*)yt, M, H, K, T);
    return VC_OK;
}
int vc_bcast_tanh(int ts, const void* src, float* out, long M, int H, int T, vc_stream_t s) {
    dim3 g((unsigned)VC_CEIL_DIV(M * H, 256));
    if (ts == VC_BF16) VC_LAUNCH((bcast_tanh_kernel<vc_bf16>), g, dim3(256), 0, s, (const vc_bf16*)src, out, M, H, T);
    else VC_LAUNCH((bcast_tanh_kernel<float>), g, dim3(256), 0, s, (const float*)src, out, M, H, T);
    return VC_OK;
}
int vc_add_inplace(float* a, const float* b, long n, vc_stream_t s) {
    VC_LAUNCH(add_inplace_kernel, dim3((unsigned)VC_CEIL_DIV(n, 256)), dim3(256), 0, s, a, b, n);
    return VC_OK;
}
int vc_scale(const float* x, float* y, long n, float alpha, vc_stream_t s) {
    if (n <= 0) return VC_OK;
    if (((uintptr_t)x | (uintptr_t)y) & 15) {          // (caller-supplied dlogits of vcad_backward* on an fp16 engine: any 4-byte aligned view works)
        long nb = VC_CEIL_DIV(n, 256); if (nb > 8192) nb = 8192;
        VC_LAUNCH(scale_unaligned_kernel, dim3((unsigned)nb), dim3(256), 0, s, x, y, n, alpha);
        return VC_OK;
    }
    long nb = VC_CEIL_DIV(n / 4 + 1, 256); if (nb > 4096) nb = 4096;
    VC_LAUNCH(scale_kernel, dim3((unsigned)nb), dim3(256), 0, s, x, y, n, alpha);
    return VC_OK;
}
int vc_wire_amax(const float* g, long n, float* amax_out, vc_stream_t s) {
    long nb = VC_CEIL_DIV(n, 1024); if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
    VC_LAUNCH(wire_amax_stage1_kernel, dim3((unsigned)nb), dim3(256), 0, s, g, n, amax_out + 1);
    VC_LAUNCH(wire_amax_stage2_kernel, dim3(1), dim3(256), 0, s, (const float*)(amax_out + 1), (int)nb, amax_out);
    return VC_OK;
}
static int wire_check(const void* a, const void* b, const char* what) {
    if (((uintptr_t)a & 15) || ((uintptr_t)b & 7)) { vc_set_error("%s: the gradient range must be 16-byte aligned, the wire buffer 8-byte aligned", what); return VC_ERR_ARG; }
    return VC_OK;
}
int vc_wire_pack(const float* g, void* wire, long n, const float* amax, int world, vc_stream_t s) {
    if (n <= 0) return VC_OK;
    if (int rc = wire_check(g, wire, "vc_wire_pack")) return rc;
    long nb = VC_CEIL_DIV(n / 4 + 1, 256); if (nb > 4096) nb = 4096;
    VC_LAUNCH(wire_pack_kernel, dim3((unsigned)nb), dim3(256), 0, s, g, (vc_bf16*)wire, n, amax, world);
    return VC_OK;
}
int vc_wire_unpack(const void* wire, float* g, long n, const float* amax, int world, vc_stream_t s) {
    if (n <= 0) return VC_OK;
    if (int rc = wire_check(g, wire, "vc_wire_unpack")) return rc;
    long nb = VC_CEIL_DIV(n / 4 + 1, 256); if (nb > 4096) nb = 4096;
    VC_LAUNCH(wire_unpack_kernel, dim3((unsigned)nb), dim3(256), 0, s, (const vc_bf16*)wire, g, n, amax, world);
    return VC_OK;
}
VC_KERNEL __launch_bounds__(256) void pack_x3_kernel(const float* x, uint32_t* y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = vc_pk_pack(x[i]);
}
int vc_pack_x3(const float* x, uint32_t* y, long n, vc_stream_t s) {
    long nb = VC_CEIL_DIV(n, 256); if (nb > 8192) nb = 8192; if (nb < 1) nb = 1;
    VC_LAUNCH(pack_x3_kernel, dim3((unsigned)nb), dim3(256), 0, s, x, y, n);
    return VC_OK;
}
int vc_cast(int ty, const float* x, void* y, long n, vc_stream_t s) {
    dim3 g((unsigned)VC_CEIL_DIV(n, 1024));
    if (ty == VC_BF16) VC_LAUNCH((cast_kernel<vc_bf16>), g, dim3(256), 0, s, x, (vc_bf16*)y, n);
    else VC_LAUNCH((cast_kernel<float>), g, dim3(256), 0, s, x, (float*)y, n);
    return VC_OK;
}

// the ViT MLP's activation as its own pass (norm.h): compact bf16 tensors, cols % 8 == 0, 16-byte aligned
static int act_check(const void* a, const void* b, long rows, int cols, const char* what) {
    if (cols % 8 || ((uintptr_t)a % 16) || ((uintptr_t)b % 16) || (double)rows * cols >= 4294967296.0) { vc_set_error("%s: needs 16-byte aligned compact bf16 tensors, cols %% 8 == 0, < 2^32 elements", what); return VC_ERR_ARG; }
    return VC_OK;
}
int vc_act_fwd_bf16(const void* z, void* g, long rows, int cols, int act, vc_drop d, vc_stream_t s) {
    if (rows <= 0) return VC_OK;
    if (int rc = act_check(z, g, rows, cols, "act_fwd")) return rc;
    if (act == VC_ACT_GELU) act = VC_ACT_GELU_FAST;                 // bf16 mode: the cheap erf of the GEMM epilogues (gemm.h)
    ProfScope ps(VC_CAT_OTHER, 0, (double)rows * cols * 4, s);
    const long n8 = rows * cols / 8;
    VC_LAUNCH(act_fwd_bf16_kernel, dim3((unsigned)VC_CEIL_DIV(n8, 256)), dim3(256), 0, s, (const vc_bf16*)z, (vc_bf16*)g, n8, act, d);
    return VC_OK;
}
// colsum_out (optional): column sums of the result, i.e. the bias gradient of the Linear whose pre-activation z is; partial_ws holds
// vc_dact_bwd_blocks(rows, cols) x cols floats, colsum_ws as for vc_colsum
// (a width the fused partial-row form does not take — more than 256 octets per row, or an octet count that does not divide 256 — has no row blocks: 1,
// never a division by zero; vc_dact_bwd_fused_ok says whether the form applies)
bool vc_dact_bwd_fused_ok(int cols) { const int c8n = cols / 8; return cols % 8 == 0 && c8n >= 1 && c8n <= 256 && 256 % c8n == 0; }
long vc_dact_bwd_blocks(long rows, int cols) {
    if (!vc_dact_bwd_fused_ok(cols)) return 1;
    const long b = VC_CEIL_DIV(rows, (long)(256 / (cols / 8))); return b > 2048 ? 2048 : (b < 1 ? 1 : b);
}
int vc_dact_bwd_bf16(void* dz, const void* z, long rows, int cols, int kind, vc_drop d, vc_stream_t s, float* colsum_out, float* partial_ws, size_t partial_bytes, float* colsum_ws, bool defer_reduce) {
    if (rows <= 0) return VC_OK;
    if (int rc = act_check(dz, z, rows, cols, "dact_bwd")) return rc;
    if (kind == VC_ACT_GELU) kind = VC_ACT_GELU_FAST;
    const int c8n = cols / 8;
    const bool fused = colsum_out && c8n <= 256 && 256 % c8n == 0 && partial_ws && (size_t)vc_dact_bwd_blocks(rows, cols) * cols * 4 <= partial_bytes;
    {
        ProfScope ps(VC_CAT_OTHER, 0, (double)rows * cols * 6, s);
        if (fused) {
            VC_LAUNCH(dact_bwd_bf16_rows_kernel, dim3((unsigned)vc_dact_bwd_blocks(rows, cols)), dim3(256), 0, s, (vc_bf16*)dz, (const vc_bf16*)z, rows, cols, kind, d, partial_ws);
        } else {
            const long n8 = rows * cols / 8;
            VC_LAUNCH(dact_bwd_bf16_kernel, dim3((unsigned)VC_CEIL_DIV(n8, 256)), dim3(256), 0, s, (vc_bf16*)dz, (const vc_bf16*)z, n8, kind, d);
        }
    }
    if (defer_reduce) { if (!fused) { vc_set_error("dact_bwd: deferred reduction needs the fused partial-row form"); return VC_ERR_ARG; } return VC_OK; }
    if (!colsum_out) return VC_OK;
    if (fused) return vc_colsum(VC_F32, partial_ws, cols, vc_dact_bwd_blocks(rows, cols), cols, colsum_out, 0, 1, 0, 0, colsum_ws, s);
    return vc_colsum(VC_BF16, dz, cols, rows, cols, colsum_out, 0, 1, 0, 0, colsum_ws, s);
}

int vc_colsum_grouped(const ColsumJob* jobs, int njobs, int strips, int max_chunks, float* partial, vc_stream_t s) {
    if (njobs <= 0) return VC_OK;
    ProfScope ps(VC_CAT_OTHER, 0, 0, s);
    VC_LAUNCH(colsum_grouped_kernel, dim3((unsigned)strips, (unsigned)max_chunks), dim3(256), 0, s, jobs, njobs, partial, 0);
    VC_LAUNCH(colsum_grouped_kernel, dim3((unsigned)strips, 1), dim3(256), 0, s, jobs, njobs, partial, 1);
    return VC_OK;
}
// LayerNorm affine folded into the Linear behind it (norm.h pe_fold_kernel / pe_fold_bwd_kernel)
int vc_pe_fold(const float* W, const float* b, const float* gamma, const float* beta, void* Wf, float* bf, int D, int K, vc_stream_t s) {
    ProfScope ps(VC_CAT_OTHER, 0, (double)D * K * 6, s);
    VC_LAUNCH(pe_fold_kernel, dim3((unsigned)D), dim3(256), 0, s, W, b, gamma, beta, (vc_bf16*)Wf, bf, K);
    return VC_OK;
}
// dgamma | dbeta [2 K] must be contiguous (norm weight / bias adjacent in the flat buffer); partial_ws >= PE_FOLD_CHUNKS * 2 K floats, colsum_ws as for vc_colsum
enum { PE_FOLD_CHUNKS = 32 };
int vc_pe_fold_bwd(const float* dWf, const float* S, const float* W, const float* gamma, const float* beta, float* dW, float* dgamma_dbeta, int D, int K,
                   float* partial_ws, float* colsum_ws, vc_stream_t s) {
    {
        ProfScope ps(VC_CAT_OTHER, 0, (double)D * K * 12, s);
        VC_LAUNCH(pe_fold_bwd_kernel, dim3((unsigned)VC_CEIL_DIV(K, 256), PE_FOLD_CHUNKS), dim3(256), 0, s, dWf, S, W, gamma, beta, dW, partial_ws, D, K);
    }
    return vc_colsum(VC_F32, partial_ws, 2L * K, PE_FOLD_CHUNKS, 2 * K, dgamma_dbeta, 0, 1, 0, 0, colsum_ws, s);
}
int vc_transpose_bf16(const vc_bf16* src, vc_bf16* dst, int rows, int cols, vc_stream_t s) {
    VC_LAUNCH(transpose_bf16_kernel, dim3((unsigned)VC_CEIL_DIV(cols, 32), (unsigned)VC_CEIL_DIV(rows, 32)), dim3(256), 0, s, src, dst, rows, cols);
    return VC_OK;
}

int vc_transpose_bf16_batched(const vc_bf16* S, vc_bf16* D, const long* src_off, const long* dst_off, const int* rows, const int* cols, int n, vc_stream_t s) {
    for (int j0 = 0; j0 < n; j0 += 32) {
        TransposeBatch tb = TransposeBatch();
        tb.n = n - j0 < 32 ? n - j0 : 32;
        int tiles = 0;
        for (int j = 0; j < tb.n; ++j) {
            tb.tile_start[j] = tiles; tb.src_off[j] = src_off[j0 + j]; tb.dst_off[j] = dst_off[j0 + j]; tb.rows[j] = rows[j0 + j]; tb.cols[j] = cols[j0 + j];
            tiles += VC_CEIL_DIV(rows[j0 + j], 64) * VC_CEIL_DIV(cols[j0 + j], 64);
        }
        tb.tile_start[tb.n] = tiles;
        if (tiles) VC_LAUNCH(transpose_bf16_batched_kernel, dim3((unsigned)tiles), dim3(256), 0, s, S, D, tb);
    }
    return VC_OK;
}

int vc_loss_fwd(LossParams p, vc_stream_t s) {
    ProfScope ps(VC_CAT_LOSS, 0, (double)p.M * (VC_NPARAM * VC_NVAL + VC_NCMD) * 4, s);
    VC_LAUNCH(loss_rows_kernel, dim3((unsigned)VC_CEIL_DIV(p.M * VC_NPARAM, 4)), dim3(256), 0, s, p);
    VC_LAUNCH(loss_cmd_rows_kernel, dim3((unsigned)VC_CEIL_DIV(p.M, 256)), dim3(256), 0, s, p);
    VC_LAUNCH(loss_finalize_kernel, dim3(1), dim3(256), 0, s, p);
    return VC_OK;
}
int vc_loss_bwd(LossParams p, vc_stream_t s) {
    ProfScope ps(VC_CAT_LOSS, 0, (double)p.M * (VC_NPARAM * VC_NVAL + VC_NCMD) * 8, s);
    VC_LAUNCH(loss_dlogits_kernel, dim3((unsigned)VC_CEIL_DIV(p.M * 7, 4)), dim3(256), 0, s, p);
    return VC_OK;
}

int vc_grad_norm(const float* g, long n, float max_norm, float gscale, float* partial, float* norm_out, vc_stream_t s) {
    ProfScope ps(VC_CAT_OPTIM, 0, (double)n * 4, s);
    long nb = VC_CEIL_DIV(n, 1024); if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
    VC_LAUNCH(sumsq_stage1_kernel, dim3((unsigned)nb), dim3(256), 0, s, g, n, partial);
    VC_LAUNCH(sumsq_stage2_kernel, dim3(1), dim3(256), 0, s, (const float*)partial, (int)nb, max_norm, gscale, norm_out);
    return VC_OK;
}
int vc_adam(AdamParams a, vc_stream_t s) {
    ProfScope ps(VC_CAT_OPTIM, 0, (double)a.n * (28 + (a.shadow ? 2 : 0) + (a.shadow_pk ? 4 : 0)), s);
    long nb = VC_CEIL_DIV(a.n, 256); if (nb > 8192) nb = 8192; if (nb < 1) nb = 1;
    VC_LAUNCH(adam_kernel, dim3((unsigned)nb), dim3(256), 0, s, a);
    return VC_OK;
}
