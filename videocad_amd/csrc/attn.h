// attn.h — windowed softmax attention, forward and backward, one wave64 per (batch, head, row).
//
// Covers every attention on the hot path with one mask rule: key j is visible to query i iff
//     lo(i) <= j <= hi(i),   lo = max(0, i - window + 1),   hi = causal ? i : Tk - 1
//   * ViT self-attention (50 tokens, 16 x 64):   window >= Tk, causal = 0
//   * decoder self-attention (4 x 256, causal):  window >= T,  causal = 1   (generate_square_subsequent_mask,
//                                                 reference model/autoregressive_transformer.py:180)
//   * decoder cross-attention band:              window = 10,  causal = 1   (reference :182-188) — only the <= 10
//                                                 visible keys are touched; the reference materialises T x T.
// Element (b, t, h, d) of q/k/v/o lives at base + (b*T + t)*ld + h*D + d, so the packed projections
// ([rows, 3*inner]) are consumed in place with pointer offsets — no head-split copies.
//
// Scores: lane = key (each lane walks one K row, q broadcast from a wave-private LDS row);
// softmax: wave-shuffle max / sum;  PV: lane = DPL consecutive output dims, p broadcast lane->wave.
// Backward is the usual two-kernel split (no atomics, deterministic):
//   attn_bwd_q : per query  -> D_i = sum_j P_ij dP_ij, dq_i
//   attn_bwd_kv: per key    -> dk_j, dv_j  (recomputes P from the saved log-sum-exp)
// (The host clamps `window` to max(Tq, Tk) so i + window never overflows.)
#pragma once
#include "vc_rt.h"

struct AttnParams {
    const void* q; const void* k; const void* v; void* o;        // type T
    long ldq, ldk, ldv, ldo;
    float* lse;                                                  // [B][H][Tq]
    int B, H, Tq, Tk, window, causal;
    float scale;
    // backward
    const void* dout; long lddo;                                 // type T
    void* dq; void* dk; void* dv; long lddq, lddk, lddv;         // type T
    float* delta;                                                // [B][H][Tq]  D_i
    vc_drop drop;                                                // attention-probability dropout, idx = ((b*H+h)*Tq+i)*Tk+j (key 0 = off)
    // cached single-step inference (forward, wave-per-row kernel only): query row i sits at absolute position qpos + i, and clip b's
    // keys / values start at row b * kv_rows of the cache (0 = the dense layout: b * Tk)
    int qpos; int kv_rows;
    // fp32 tensors of the bf16x3 compute mode: 1 = the kernels that have an x3 form (attn_x3.h) run on the bf16 matrix cores with hi / lo split operands;
    // 2 = additionally q / k / v / dout / o / dq / dk / dv are pre-split hi | lo words (gemm.h vc_pk) — only shapes with an x3 kernel
    int x3;
    // ViT attention backward (attn_mfma.h, r06): L2 warm-up distance in frames, set by the launcher (0 = off)
    int pf_frames;
};

// sum_d row[d] * bc[d]; bc: LDS, same address for all lanes (broadcast).  The row is walked in 16-byte chunks, four
// chunks in flight per step (a scalar element loop makes hipcc wait for every load in turn).  Rows must be 16-byte aligned.
template <typename T> VC_DEV float attn_dot_chunk(const vc_u32x4& c, const float* bc);
template <> VC_DEV float attn_dot_chunk<float>(const vc_u32x4& c, const float* bc) {
    return vc_bits_f32(c.x) * bc[0] + vc_bits_f32(c.y) * bc[1] + vc_bits_f32(c.z) * bc[2] + vc_bits_f32(c.w) * bc[3];
}
template <> VC_DEV float attn_dot_chunk<vc_bf16>(const vc_u32x4& c, const float* bc) {
    return vc_lo16_f32(c.x) * bc[0] + vc_hi16_f32(c.x) * bc[1] + vc_lo16_f32(c.y) * bc[2] + vc_hi16_f32(c.y) * bc[3] +
           vc_lo16_f32(c.z) * bc[4] + vc_hi16_f32(c.z) * bc[5] + vc_lo16_f32(c.w) * bc[6] + vc_hi16_f32(c.w) * bc[7];
}
template <typename T, int D>
VC_DEV float attn_dot_row(const T* row, const float* bc) {
    constexpr int EPC = 16 / (int)sizeof(T);
    static_assert(D % (4 * EPC) == 0, "head dim must be a multiple of four 16-byte chunks");
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 2
    for (int d = 0; d < D; d += 4 * EPC) {
        const vc_u32x4 c0 = *reinterpret_cast<const vc_u32x4*>(row + d), c1 = *reinterpret_cast<const vc_u32x4*>(row + d + EPC);
        const vc_u32x4 c2 = *reinterpret_cast<const vc_u32x4*>(row + d + 2 * EPC), c3 = *reinterpret_cast<const vc_u32x4*>(row + d + 3 * EPC);
        s0 += attn_dot_chunk<T>(c0, bc + d); s1 += attn_dot_chunk<T>(c1, bc + d + EPC);
        s2 += attn_dot_chunk<T>(c2, bc + d + 2 * EPC); s3 += attn_dot_chunk<T>(c3, bc + d + 3 * EPC);
    }
    return (s0 + s1) + (s2 + s3);
}

// DPL consecutive elements of a row as floats (one vector load for DPL = 4)
template <typename T, int DPL> VC_DEV void attn_row_ld(const T* p, float (&v)[DPL]) {
    if constexpr (DPL == 4 && sizeof(T) == 4) { const vc_u32x4 q = *reinterpret_cast<const vc_u32x4*>(p); v[0] = vc_bits_f32(q.x); v[1] = vc_bits_f32(q.y); v[2] = vc_bits_f32(q.z); v[3] = vc_bits_f32(q.w); }
    else if constexpr (DPL == 4 && sizeof(T) == 2) { const vc_u32x2 q = *reinterpret_cast<const vc_u32x2*>(p); v[0] = vc_lo16_f32(q.x); v[1] = vc_hi16_f32(q.x); v[2] = vc_lo16_f32(q.y); v[3] = vc_hi16_f32(q.y); }
    else {
#pragma unroll
        for (int j = 0; j < DPL; ++j) v[j] = vc_ld(p + j);
    }
}
// acc[j] += sum over rows r = 0..cnt-1 of w_r * row_r[j], w_r = lane r's `w` (broadcast), four rows in flight
template <typename T, int DPL> VC_DEV void attn_accum_rows(float (&acc)[DPL], const T* base, long ld, int cnt, float w) {
    int r = 0;
    for (; r + 4 <= cnt; r += 4) {
        float v[4][DPL];
#pragma unroll
        for (int u = 0; u < 4; ++u) attn_row_ld<T, DPL>(base + (long)(r + u) * ld, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float wr = vc_shfl(w, r + u);
#pragma unroll
            for (int j = 0; j < DPL; ++j) acc[j] += wr * v[u][j];
        }
    }
    for (; r < cnt; ++r) {
        float v[DPL]; attn_row_ld<T, DPL>(base + (long)r * ld, v);
        const float wr = vc_shfl(w, r);
#pragma unroll
        for (int j = 0; j < DPL; ++j) acc[j] += wr * v[j];
    }
}

// NPASS = ceil(max visible keys / 64)
template <typename T, int DPL, int NPASS>
VC_KERNEL __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
    constexpr int D = 64 * DPL;
    VC_SHARED float qs[4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wid = (long)blockIdx.x * 4 + wave;
    const long total = (long)p.B * p.H * p.Tq;
    if (wid >= total) return;
    const int i = (int)(wid % p.Tq); const int h = (int)((wid / p.Tq) % p.H); const int b = (int)(wid / ((long)p.Tq * p.H));
    const int ia = i + p.qpos;                                   // absolute position of this query
    const long kvb = (long)b * (p.kv_rows ? p.kv_rows : p.Tk);   // first key / value row of clip b
    const int lo = (ia - p.window + 1 > 0) ? (ia - p.window + 1) : 0;
    const int hi = p.causal ? (ia < p.Tk - 1 ? ia : p.Tk - 1) : (p.Tk - 1);
    const int nk = hi - lo + 1;
    const T* qrow = (const T*)p.q + ((long)b * p.Tq + i) * p.ldq + h * D;
#pragma unroll
    for (int j = 0; j < DPL; ++j) qs[wave][lane * DPL + j] = vc_ld(qrow + lane * DPL + j);
    vc_wave_barrier();   // wave-private LDS row: in-order per wave, no workgroup barrier needed
    float s[NPASS];
    float m = -INFINITY;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int jj = ps * 64 + lane;
        s[ps] = -INFINITY;
        if (jj < nk) {
            const T* krow = (const T*)p.k + (kvb + lo + jj) * p.ldk + h * D;
            s[ps] = attn_dot_row<T, D>(krow, qs[wave]) * p.scale;
        }
        m = fmaxf(m, s[ps]);
    }
    m = vc_wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) { s[ps] = (ps * 64 + lane < nk) ? expf(s[ps] - m) : 0.f; l += s[ps]; }
    l = vc_wave_sum(l);
    if (p.drop.key) {                                            // dropout acts on the normalised probabilities, not on l
        const long base = (((long)b * p.H + h) * p.Tq + i) * p.Tk + lo;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) if (ps * 64 + lane < nk) s[ps] *= vc_drop_mul(p.drop, base + ps * 64 + lane);
    }
    float acc[DPL];
#pragma unroll
    for (int j = 0; j < DPL; ++j) acc[j] = 0.f;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        int cnt = nk - ps * 64; cnt = cnt > 64 ? 64 : cnt;
        attn_accum_rows<T, DPL>(acc, (const T*)p.v + (kvb + lo + ps * 64) * p.ldv + h * D + lane * DPL, p.ldv, cnt, s[ps]);
    }
    const float inv = 1.0f / l;
    T* orow = (T*)p.o + ((long)b * p.Tq + i) * p.ldo + h * D + lane * DPL;
#pragma unroll
    for (int j = 0; j < DPL; ++j) vc_st(orow + j, acc[j] * inv);
    if (p.lse && lane == 0) p.lse[wid] = m + logf(l);
}

template <typename T, int DPL, int NPASS>
VC_KERNEL __launch_bounds__(256) void attn_bwd_q_kernel(AttnParams p) {
    constexpr int D = 64 * DPL;
    VC_SHARED float qs[4][D];
    VC_SHARED float dos[4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wid = (long)blockIdx.x * 4 + wave;
    const long total = (long)p.B * p.H * p.Tq;
    if (wid >= total) return;
    const int i = (int)(wid % p.Tq); const int h = (int)((wid / p.Tq) % p.H); const int b = (int)(wid / ((long)p.Tq * p.H));
    const int lo = (i - p.window + 1 > 0) ? (i - p.window + 1) : 0;
    const int hi = p.causal ? (i < p.Tk - 1 ? i : p.Tk - 1) : (p.Tk - 1);
    const int nk = hi - lo + 1;
    const T* qrow = (const T*)p.q + ((long)b * p.Tq + i) * p.ldq + h * D;
    const T* dorow = (const T*)p.dout + ((long)b * p.Tq + i) * p.lddo + h * D;
#pragma unroll
    for (int j = 0; j < DPL; ++j) {
        qs[wave][lane * DPL + j] = vc_ld(qrow + lane * DPL + j);
        dos[wave][lane * DPL + j] = vc_ld(dorow + lane * DPL + j);
    }
    vc_wave_barrier();
    const float lse = p.lse[wid];
    float pr[NPASS], dp[NPASS];
    float dsum = 0.f;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int jj = ps * 64 + lane;
        pr[ps] = 0.f; dp[ps] = 0.f;
        if (jj < nk) {
            const T* krow = (const T*)p.k + ((long)b * p.Tk + lo + jj) * p.ldk + h * D;
            const T* vrow = (const T*)p.v + ((long)b * p.Tk + lo + jj) * p.ldv + h * D;
            pr[ps] = expf(attn_dot_row<T, D>(krow, qs[wave]) * p.scale - lse);
            dp[ps] = attn_dot_row<T, D>(vrow, dos[wave]);
            if (p.drop.key) dp[ps] *= vc_drop_mul(p.drop, (((long)b * p.H + h) * p.Tq + i) * p.Tk + lo + jj);   // dP = dP' * mask
        }
        dsum += pr[ps] * dp[ps];
    }
    dsum = vc_wave_sum(dsum);
    float acc[DPL];
#pragma unroll
    for (int j = 0; j < DPL; ++j) acc[j] = 0.f;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const float ds_mine = pr[ps] * (dp[ps] - dsum);
        int cnt = nk - ps * 64; cnt = cnt > 64 ? 64 : cnt;
        attn_accum_rows<T, DPL>(acc, (const T*)p.k + ((long)b * p.Tk + lo + ps * 64) * p.ldk + h * D + lane * DPL, p.ldk, cnt, ds_mine);
    }
    T* dqrow = (T*)p.dq + ((long)b * p.Tq + i) * p.lddq + h * D + lane * DPL;
#pragma unroll
    for (int j = 0; j < DPL; ++j) vc_st(dqrow + j, acc[j] * p.scale);
    if (lane == 0) p.delta[wid] = dsum;
}

// NPASS = ceil(max queries that can see one key / 64)
template <typename T, int DPL, int NPASS>
VC_KERNEL __launch_bounds__(256) void attn_bwd_kv_kernel(AttnParams p) {
    constexpr int D = 64 * DPL;
    VC_SHARED float ks[4][D];
    VC_SHARED float vs[4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wid = (long)blockIdx.x * 4 + wave;
    const long total = (long)p.B * p.H * p.Tk;
    if (wid >= total) return;
    const int jk = (int)(wid % p.Tk); const int h = (int)((wid / p.Tk) % p.H); const int b = (int)(wid / ((long)p.Tk * p.H));
    const int i0 = p.causal ? jk : 0;
    int i1 = jk + p.window - 1; if (i1 > p.Tq - 1) i1 = p.Tq - 1;
    const int nq = i1 - i0 + 1;                                  // may be <= 0 (key seen by nobody)
    const T* krow = (const T*)p.k + ((long)b * p.Tk + jk) * p.ldk + h * D;
    const T* vrow = (const T*)p.v + ((long)b * p.Tk + jk) * p.ldv + h * D;
#pragma unroll
    for (int j = 0; j < DPL; ++j) {
        ks[wave][lane * DPL + j] = vc_ld(krow + lane * DPL + j);
        vs[wave][lane * DPL + j] = vc_ld(vrow + lane * DPL + j);
    }
    vc_wave_barrier();
    float pr[NPASS], ds[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int ii = ps * 64 + lane;
        pr[ps] = 0.f; ds[ps] = 0.f;
        if (ii < nq) {
            const int i = i0 + ii;
            const long sidx = ((long)b * p.H + h) * p.Tq + i;
            const T* qrow = (const T*)p.q + ((long)b * p.Tq + i) * p.ldq + h * D;
            const T* dorow = (const T*)p.dout + ((long)b * p.Tq + i) * p.lddo + h * D;
            pr[ps] = expf(attn_dot_row<T, D>(qrow, ks[wave]) * p.scale - p.lse[sidx]);
            const float ms = p.drop.key ? vc_drop_mul(p.drop, (((long)b * p.H + h) * p.Tq + i) * p.Tk + jk) : 1.0f;
            ds[ps] = pr[ps] * (attn_dot_row<T, D>(dorow, vs[wave]) * ms - p.delta[sidx]);
            pr[ps] *= ms;                                        // dV uses the dropped probabilities
        }
    }
    float dk[DPL], dv[DPL];
#pragma unroll
    for (int j = 0; j < DPL; ++j) { dk[j] = 0.f; dv[j] = 0.f; }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        int cnt = nq - ps * 64; cnt = cnt > 64 ? 64 : cnt;
        attn_accum_rows<T, DPL>(dv, (const T*)p.dout + ((long)b * p.Tq + i0 + ps * 64) * p.lddo + h * D + lane * DPL, p.lddo, cnt, pr[ps]);
        attn_accum_rows<T, DPL>(dk, (const T*)p.q + ((long)b * p.Tq + i0 + ps * 64) * p.ldq + h * D + lane * DPL, p.ldq, cnt, ds[ps]);
    }
    T* dkrow = (T*)p.dk + ((long)b * p.Tk + jk) * p.lddk + h * D + lane * DPL;
    T* dvrow = (T*)p.dv + ((long)b * p.Tk + jk) * p.lddv + h * D + lane * DPL;
#pragma unroll
    for (int j = 0; j < DPL; ++j) { vc_st(dkrow + j, dk[j] * p.scale); vc_st(dvrow + j, dv[j]); }
}


// ---- single-query attention forward, bf16, D = 64 (the cls-only last ViT layer: one query against <= 64 keys per (frame, head)).  One wave per
// (batch, head): lane = key for the scores (each lane walks its own K row), then the V rows are read as 16-byte pieces, eight rows per
// instruction, and the eight row groups are folded with three xor-shuffles — the wave-per-row kernel above read V two bytes per lane, one
// row per instruction (208 us for the frame ViT's 32 768 heads; r02).
VC_KERNEL __launch_bounds__(256) void attn_fwd_single_query_bf16_kernel(AttnParams p) {
    constexpr int D = 64;
    VC_SHARED float qs[4][D];
    VC_SHARED float ps[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wid = (long)blockIdx.x * 4 + wave;
    if (wid >= (long)p.B * p.H) return;
    const int h = (int)(wid % p.H); const long b = wid / p.H;
    const vc_bf16* qrow = (const vc_bf16*)p.q + b * p.ldq + h * D;                     // Tq = 1: query row b
    qs[wave][lane] = vc_ld(qrow + lane);
    vc_wave_barrier();
    const bool on = lane < p.Tk;
    const vc_bf16* krow = (const vc_bf16*)p.k + (b * p.Tk + (on ? lane : 0)) * p.ldk + h * D;
    const float sc = on ? attn_dot_row<vc_bf16, D>(krow, qs[wave]) * p.scale : -INFINITY;
    const float m = vc_wave_max(sc);
    float e = on ? expf(sc - m) : 0.f;
    const float l = vc_wave_sum(e);
    if (p.drop.key && on) e *= vc_drop_mul(p.drop, wid * p.Tk + lane);                 // Tq = 1: idx = (b*H+h)*Tk + j (dropout acts on the normalised probabilities)
    ps[wave][lane] = e;
    vc_wave_barrier();
    const int rg = lane >> 3, cg = lane & 7;
    const vc_bf16* vbase = (const vc_bf16*)p.v + b * p.Tk * p.ldv + h * D + cg * 8;
    vc_u32x4 vv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) { const int r = it * 8 + rg; vv[it] = *reinterpret_cast<const vc_u32x4*>(vbase + (long)(r < p.Tk ? r : p.Tk - 1) * p.ldv); }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int r = it * 8 + rg;
        const float w = r < p.Tk ? ps[wave][r] : 0.f;
        const uint32_t u[4] = {vv[it].x, vv[it].y, vv[it].z, vv[it].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[2 * k] += w * vc_lo16_f32(u[k]); acc[2 * k + 1] += w * vc_hi16_f32(u[k]); }
    }
#pragma unroll
    for (int off = 8; off < 64; off <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += vc_shfl_xor(acc[j], off);
    if (lane < 8) {
        const float inv = 1.0f / l;
        vc_u32x4 o;
        o.x = vc_pack_bf16x2(acc[0] * inv, acc[1] * inv); o.y = vc_pack_bf16x2(acc[2] * inv, acc[3] * inv);
        o.z = vc_pack_bf16x2(acc[4] * inv, acc[5] * inv); o.w = vc_pack_bf16x2(acc[6] * inv, acc[7] * inv);
        *reinterpret_cast<vc_u32x4*>((vc_bf16*)p.o + b * p.ldo + h * D + lane * 8) = o;
    }
    if (p.lse && lane == 0) p.lse[wid] = m + logf(l);
}

// ---- single-query attention backward (the ViT's last layer only consumes the cls token: Tq = 1).  One wave per
// (batch, head): lane = key for the scores / dS (each lane walks its own K and V row), then lane = DPL output dims for
// dq; dk_j = dS_j q and dv_j = P_j dO are written by the key's lane.  Replaces B*H*Tk one-key waves by B*H waves.
template <typename T, int DPL>
VC_KERNEL __launch_bounds__(256) void attn_bwd_single_query_kernel(AttnParams p) {
    constexpr int D = 64 * DPL;
    VC_SHARED float qs[4][D];
    VC_SHARED float dos[4][D];
    VC_SHARED float dss[4][64];
    VC_SHARED float pss[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wid = (long)blockIdx.x * 4 + wave;
    if (wid >= (long)p.B * p.H) return;
    const int h = (int)(wid % p.H); const long b = wid / p.H;
    const T* qrow = (const T*)p.q + b * p.ldq + h * D;                     // Tq = 1: query row b
    const T* dorow = (const T*)p.dout + b * p.lddo + h * D;
#pragma unroll
    for (int j = 0; j < DPL; ++j) {
        qs[wave][lane * DPL + j] = vc_ld(qrow + lane * DPL + j);
        dos[wave][lane * DPL + j] = vc_ld(dorow + lane * DPL + j);
    }
    vc_wave_barrier();
    const float lse = p.lse[wid];
    const bool on = lane < p.Tk;
    const T* krow = (const T*)p.k + (b * p.Tk + (on ? lane : 0)) * p.ldk + h * D;
    const T* vrow = (const T*)p.v + (b * p.Tk + (on ? lane : 0)) * p.ldv + h * D;
    float pr = 0.f, dp = 0.f;
    if (on) { pr = expf(attn_dot_row<T, D>(krow, qs[wave]) * p.scale - lse); dp = attn_dot_row<T, D>(vrow, dos[wave]); }
    const float ms = (p.drop.key && on) ? vc_drop_mul(p.drop, wid * p.Tk + lane) : 1.0f;       // Tq = 1: idx = (b*H+h)*Tk + j
    dp *= ms;
    const float dsum = vc_wave_sum(pr * dp);
    const float ds = pr * (dp - dsum);
    dss[wave][lane] = ds;
    pss[wave][lane] = pr * ms;
    vc_wave_barrier();
    if constexpr (sizeof(T) == 2 && DPL == 1) {
        // bf16, D = 64 (r02): eight key rows per instruction, 16 bytes per lane — dk_j = scale dS_j q, dv_j = P_j dO as whole 128-byte rows,
        // dq = scale sum_j dS_j k_j folded over the eight row groups with three xor-shuffles
        const int rg = lane >> 3, cg = lane & 7;
        float qv[8], dov[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { qv[j] = qs[wave][cg * 8 + j] * p.scale; dov[j] = dos[wave][cg * 8 + j]; }
        const T* kb = (const T*)p.k + b * p.Tk * p.ldk + h * D + cg * 8;
        vc_u32x4 kk[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) { const int r = it * 8 + rg; kk[it] = *reinterpret_cast<const vc_u32x4*>(kb + (long)(r < p.Tk ? r : p.Tk - 1) * p.ldk); }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = it * 8 + rg;
            const bool live = r < p.Tk;
            const float dsj = live ? dss[wave][r] : 0.f, pj = live ? pss[wave][r] : 0.f;
            const uint32_t u[4] = {kk[it].x, kk[it].y, kk[it].z, kk[it].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[2 * k] += dsj * vc_lo16_f32(u[k]); acc[2 * k + 1] += dsj * vc_hi16_f32(u[k]); }
            if (live) {
                vc_u32x4 wk, wv;
                wk.x = vc_pack_bf16x2(dsj * qv[0], dsj * qv[1]); wk.y = vc_pack_bf16x2(dsj * qv[2], dsj * qv[3]);
                wk.z = vc_pack_bf16x2(dsj * qv[4], dsj * qv[5]); wk.w = vc_pack_bf16x2(dsj * qv[6], dsj * qv[7]);
                wv.x = vc_pack_bf16x2(pj * dov[0], pj * dov[1]); wv.y = vc_pack_bf16x2(pj * dov[2], pj * dov[3]);
                wv.z = vc_pack_bf16x2(pj * dov[4], pj * dov[5]); wv.w = vc_pack_bf16x2(pj * dov[6], pj * dov[7]);
                *reinterpret_cast<vc_u32x4*>((T*)p.dk + (b * p.Tk + r) * p.lddk + h * D + cg * 8) = wk;
                *reinterpret_cast<vc_u32x4*>((T*)p.dv + (b * p.Tk + r) * p.lddv + h * D + cg * 8) = wv;
            }
        }
#pragma unroll
        for (int off = 8; off < 64; off <<= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += vc_shfl_xor(acc[j], off);
        if (lane < 8) {
            vc_u32x4 o;
            o.x = vc_pack_bf16x2(acc[0] * p.scale, acc[1] * p.scale); o.y = vc_pack_bf16x2(acc[2] * p.scale, acc[3] * p.scale);
            o.z = vc_pack_bf16x2(acc[4] * p.scale, acc[5] * p.scale); o.w = vc_pack_bf16x2(acc[6] * p.scale, acc[7] * p.scale);
            *reinterpret_cast<vc_u32x4*>((T*)p.dq + b * p.lddq + h * D + lane * 8) = o;
        }
        if (lane == 0) p.delta[wid] = dsum;
        return;
    }
    {   // dk_j = scale * dS_j * q ;  dv_j = P_j * dO   — lane = dims, so every key row is one coalesced store
        float qv[DPL], dov[DPL];
#pragma unroll
        for (int j = 0; j < DPL; ++j) { qv[j] = qs[wave][lane * DPL + j] * p.scale; dov[j] = dos[wave][lane * DPL + j]; }
        for (int jj = 0; jj < p.Tk; ++jj) {
            const float dsj = dss[wave][jj], pj = pss[wave][jj];
            T* dk = (T*)p.dk + (b * p.Tk + jj) * p.lddk + h * D + lane * DPL;
            T* dv = (T*)p.dv + (b * p.Tk + jj) * p.lddv + h * D + lane * DPL;
#pragma unroll
            for (int j = 0; j < DPL; ++j) { vc_st(dk + j, dsj * qv[j]); vc_st(dv + j, pj * dov[j]); }
        }
    }
    float acc[DPL];
#pragma unroll
    for (int j = 0; j < DPL; ++j) acc[j] = 0.f;
    for (int jj = 0; jj < p.Tk; ++jj) {                                     // dq = scale * sum_j dS_j k_j   (lane = dims, rows coalesced)
        const float dsj = dss[wave][jj];
        const T* kr = (const T*)p.k + (b * p.Tk + jj) * p.ldk + h * D + lane * DPL;
#pragma unroll
        for (int j = 0; j < DPL; ++j) acc[j] += dsj * vc_ld(kr + j);
    }
    T* dq = (T*)p.dq + b * p.lddq + h * D + lane * DPL;
#pragma unroll
    for (int j = 0; j < DPL; ++j) vc_st(dq + j, acc[j] * p.scale);
    if (lane == 0) p.delta[wid] = dsum;
}
