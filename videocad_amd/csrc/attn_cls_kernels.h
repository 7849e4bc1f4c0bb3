// attn_cls_kernels.h — kernels of the re-associated class-token attention (attn_cls.h has the math, the parameter block and the LDS plan).
// Included by ops_attn.hip only.
//
// One workgroup per frame.  The frame's normalised tokens ha [P1][512] (51 KB) and g (/ dc) [H][512] are staged once into padded LDS tiles; every product
// runs on the matrix cores (v_mfma_f32_32x32x16, 16-bit operands, fp32 accumulate):
//     S^T[j][h]  = sum_d ha[j][d] g[h][d]          both operands d-contiguous: 16-byte fragment reads                      (forward, backward; dP~^T alike with dc)
//     C[h][d]    = sum_j P[h][j] ha[j][d]          P from a small [32][72] tile, ha through ds_read_b64_tr_b16 (contraction over the slow index)
//     dG[h][d]   = sum_j dS[h][j] ha[j][d]         same form
//     dHA[j][d]  = sum_k T[j][k] GD[k][d]          k = 32: dS^T | P~^T against the stacked rows of g | dc — two k-steps per tile
// The softmax itself is lane = token, wave reductions (VALU).  Heads are padded to the 32-wide tile (rows >= H are zero), tokens to 64 (rows >= P1 are
// clamped re-reads whose probabilities are zero).  Results leave through the LDS tiles as whole 16-byte row pieces.
// r06 history: a first form of these kernels did the 512-long dots with v_dot2c on the VALU — 104 us forward / 287 us backward per launch at the
// benchmark shape, LDS-broadcast-bound; this form measures in profiles/r06_*.
#pragma once
#include "attn_cls.h"

// `rows` rows of CA_D 16-bit elements from global memory (leading dimension ld) into an LDS tile with row stride lds_stride: one row = 64 lanes x 16 bytes =
// one direct-to-LDS DMA instruction of one wave (global_load_lds_dwordx4 lands the wave's 1 KiB contiguously at a wave-uniform address), rows dealt round-robin
// to the waves — nothing passes through registers and every load of the tile is in flight at once (a register-staged copy loop measured 72 us per forward
// launch against 31 us of HBM time: one memory round trip per loop trip).  The caller waits (vc_wait_vmcnt<0>) before its barrier.
VC_DEV void ca_stage(const vc_bf16* src, long ld, vc_bf16* dst, int lds_stride, int rows, int tid, int nthreads) {
    const int lane = tid & 63, wave = vc_uniform(tid >> 6), nw = nthreads >> 6;
    for (int r = wave; r < rows; r += nw) vc_dma16(src + (long)r * ld + lane * 8, dst + (long)r * lds_stride);
}
VC_DEV void ca_unstage(const vc_bf16* src, int lds_stride, vc_bf16* dst, long ld, int rows, int tid, int nthreads) {
    for (int ck = tid; ck < rows * (CA_D / 8); ck += nthreads) {
        const int r = ck / (CA_D / 8), c8 = ck % (CA_D / 8);
        *reinterpret_cast<vc_u32x4*>(dst + (long)r * ld + c8 * 8) = *reinterpret_cast<const vc_u32x4*>(src + (long)r * lds_stride + c8 * 8);
    }
}
VC_DEV void ca_zero(vc_bf16* dst, int n_elems, int tid, int nthreads) {          // n_elems % 8 == 0, dst 16-byte aligned
    const vc_u32x4 z = {0u, 0u, 0u, 0u};
    for (int ck = tid; ck < n_elems / 8; ck += nthreads) *reinterpret_cast<vc_u32x4*>(dst + ck * 8) = z;
}
// B fragment of an MFMA k-step from a row-major [k][n] LDS tile (contraction index = the slow index): lane (n = lane & 31, half = lane >> 5) receives rows
// k0 + 8 half + 0..7 of column col0 + n — two transpose reads (gemm_dma.h gd_frag<true>, measured semantics in vc_rt.h); rows beyond kmax are re-reads of row kmax
VC_DEV vc_s16x8 ca_frag_tr(const vc_bf16* tile, int stride, int k0, int col0, int kmax, int lane) {
    const int i = lane & 15;
    const int k = k0 + 8 * (lane >> 5) + (i >> 2);
    const int col = col0 + ((lane >> 4) & 1) * 16 + (i & 3) * 4;
    const int ka = k < kmax ? k : kmax, kb = k + 4 < kmax ? k + 4 : kmax;
    const vc_s16x4 lo = vc_ds_read_tr16(tile + (long)ka * stride + col), hi = vc_ds_read_tr16(tile + (long)kb * stride + col);
    vc_s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
VC_DEV vc_s16x8 ca_frag(const vc_bf16* tile, int stride, int row, int k0, int lane) {       // A (or k-contiguous B) fragment: 8 k-values of one row
    return *reinterpret_cast<const vc_s16x8*>(tile + (long)row * stride + k0 + 8 * (lane >> 5));
}
// S^T block (32 tokens x 32 head columns, 16 valid) over the whole D: rows = tokens jb*32.. (clamped to P1 - 1), columns = rows hrow0 + (lane & 15) of `heads`;
// written transposed into dst[h * CA_SS + j]
VC_DEV void ca_scores(const vc_bf16* ha_s, const vc_bf16* heads, int hrow0, int jb, int P1, float* dst, int lane) {
    const int jr = (jb * 32 + (lane & 31) < P1) ? jb * 32 + (lane & 31) : P1 - 1;
    vc_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
    for (int ks = 0; ks < CA_D / 16; ++ks)
        acc = vc_mfma_32x32x16_bf16(ca_frag(ha_s, CA_HS, jr, ks * 16, lane), ca_frag(heads, CA_HS, hrow0 + (lane & 15), ks * 16, lane), acc);
    if ((lane & 31) < 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(lane & 31) * CA_SS + jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = acc[r];
    }
}

VC_KERNEL __launch_bounds__(256) void cls_attn_fwd_kernel(ClsAttnParams p) {
    VC_DYN_SHARED(vc_bf16, lds);
    vc_bf16* ha_s = lds;
    vc_bf16* g_s = ha_s + (long)p.P1 * CA_HS;
    float* sc = reinterpret_cast<float*>(g_s + 16 * CA_HS);
    vc_bf16* p16 = reinterpret_cast<vc_bf16*>(sc + 16 * CA_SS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long n = blockIdx.x;
    ca_stage((const vc_bf16*)p.ha + n * p.P1 * p.ld_ha, p.ld_ha, ha_s, CA_HS, p.P1, tid, 256);
    ca_stage((const vc_bf16*)p.g + n * p.H * CA_D, CA_D, g_s, CA_HS, p.H, tid, 256);
    if (p.H < 16) ca_zero(g_s + p.H * CA_HS, (16 - p.H) * CA_HS, tid, 256);
    ca_zero(p16, 32 * CA_PS, tid, 256);
    vc_wait_vmcnt<0>();
    vc_sync();
    if (wave < 2) ca_scores(ha_s, g_s, 0, wave, p.P1, sc, lane);
    vc_sync();
    const bool on = lane < p.P1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int h = 4 * wave + i;
        if (h < p.H) {                                        // (wave-uniform)
            const float s = on ? sc[h * CA_SS + lane] * p.scale : -INFINITY;
            const float m = vc_wave_max(s);
            float e = on ? expf(s - m) : 0.f;
            const float l = vc_wave_sum(e);
            e *= 1.0f / l;
            if (p.drop.key && on) e *= vc_drop_mul(p.drop, (uint32_t)((n * p.H + h) * p.P1 + lane));
            vc_st(p16 + h * CA_PS + lane, e);
            if (lane == 0) p.lse[n * p.H + h] = m + logf(l);
        }
    }
    vc_sync();
    vc_f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const vc_s16x8 a = ca_frag(p16, CA_PS, lane & 31, ks * 16, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = vc_mfma_32x32x16_bf16(a, ca_frag_tr(ha_s, CA_HS, ks * 16, (4 * wave + t) * 32, p.P1 - 1, lane), acc[t]);
    }
    // C rows h = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) < 16 <=> r < 8: through the (now idle) g tile, then out as whole rows
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 8; ++r) vc_st(g_s + ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CA_HS + (4 * wave + t) * 32 + (lane & 31), acc[t][r]);
    vc_sync();
    ca_unstage(g_s, CA_HS, (vc_bf16*)p.c + n * p.H * CA_D, CA_D, p.H, tid, 256);
}

VC_KERNEL __launch_bounds__(512) void cls_attn_bwd_kernel(ClsAttnParams p) {
    VC_DYN_SHARED(vc_bf16, lds);
    vc_bf16* ha_s = lds;
    vc_bf16* gd_s = ha_s + (long)p.P1 * CA_HS;                    // rows 0..15: g, rows 16..31: dc (rows >= H of either half zero)
    float* sc = reinterpret_cast<float*>(gd_s + 32 * CA_HS);
    float* dp = sc + 16 * CA_SS;
    vc_bf16* ds16 = reinterpret_cast<vc_bf16*>(dp + 16 * CA_SS);  // [32][CA_PS]: dS[h][j]
    vc_bf16* dsT = ds16 + 32 * CA_PS;                             // [64][CA_TS]: row j = dS[0..15][j] | P~[0..15][j]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long n = blockIdx.x;
    ca_stage((const vc_bf16*)p.ha + n * p.P1 * p.ld_ha, p.ld_ha, ha_s, CA_HS, p.P1, tid, 512);
    ca_stage((const vc_bf16*)p.g + n * p.H * CA_D, CA_D, gd_s, CA_HS, p.H, tid, 512);
    ca_stage((const vc_bf16*)p.dc + n * p.H * CA_D, CA_D, gd_s + 16 * CA_HS, CA_HS, p.H, tid, 512);
    if (p.H < 16) { ca_zero(gd_s + p.H * CA_HS, (16 - p.H) * CA_HS, tid, 512); ca_zero(gd_s + (16 + p.H) * CA_HS, (16 - p.H) * CA_HS, tid, 512); }
    ca_zero(ds16, 32 * CA_PS + 64 * CA_TS, tid, 512);
    vc_wait_vmcnt<0>();
    vc_sync();
    if (wave < 4) ca_scores(ha_s, gd_s, (wave >> 1) * 16, wave & 1, p.P1, (wave >> 1) ? dp : sc, lane);
    vc_sync();
    const bool on = lane < p.P1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int h = 2 * wave + i;
        if (h < p.H) {                                        // (wave-uniform)
            const float lse = p.lse[n * p.H + h];
            const float pr = on ? expf(sc[h * CA_SS + lane] * p.scale - lse) : 0.f;
            const float ms = (p.drop.key && on) ? vc_drop_mul(p.drop, (uint32_t)((n * p.H + h) * p.P1 + lane)) : 1.0f;
            const float dpv = on ? dp[h * CA_SS + lane] * ms : 0.f;
            const float dsum = vc_wave_sum(pr * dpv);
            const float ds = p.scale * pr * (dpv - dsum);
            vc_st(ds16 + h * CA_PS + lane, ds);
            vc_st(dsT + lane * CA_TS + h, ds);
            vc_st(dsT + lane * CA_TS + 16 + h, pr * ms);
        }
    }
    vc_sync();
    const int db0 = 2 * wave;                                     // this wave's two 32-wide column blocks of D
    vc_f32x16 ag[2], ah[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ag[t][r] = 0.f; ah[0][t][r] = 0.f; ah[1][t][r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                              // dG = dS ha
        const vc_s16x8 a = ca_frag(ds16, CA_PS, lane & 31, ks * 16, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) ag[t] = vc_mfma_32x32x16_bf16(a, ca_frag_tr(ha_s, CA_HS, ks * 16, (db0 + t) * 32, p.P1 - 1, lane), ag[t]);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                              // dHA = [dS^T | P~^T] [g ; dc]
        vc_s16x8 b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) b[t] = ca_frag_tr(gd_s, CA_HS, ks * 16, (db0 + t) * 32, 31, lane);
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const vc_s16x8 a = ca_frag(dsT, CA_TS, jb * 32 + (lane & 31), ks * 16, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) ah[jb][t] = vc_mfma_32x32x16_bf16(a, b[t], ah[jb][t]);
        }
    }
    if (p.r0 && lane < 32) {                                      // token 0 = accumulator register 0 of the lanes 0..31 of block jb = 0, still fp32
#pragma unroll
        for (int t = 0; t < 2; ++t) p.r0[n * CA_D + (db0 + t) * 32 + lane] = ah[0][t][0];
    }
    vc_sync();                                                    // every wave is done reading the token and g | dc tiles: they become the output tiles
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (j < p.P1) vc_st(ha_s + (long)j * CA_HS + (db0 + t) * 32 + (lane & 31), ah[jb][t][r]);
            }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 8; ++r) vc_st(gd_s + ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CA_HS + (db0 + t) * 32 + (lane & 31), ag[t][r]);
    vc_sync();
    ca_unstage(ha_s, CA_HS, (vc_bf16*)p.dha + n * p.P1 * p.ld_dha, p.ld_dha, p.P1, tid, 512);
    ca_unstage(gd_s, CA_HS, (vc_bf16*)p.dg + n * p.H * CA_D, CA_D, p.H, tid, 512);
}
