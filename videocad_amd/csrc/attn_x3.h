// attn_x3.h — ViT self-attention of the bf16x3 mode (VCAD_BF16X3) on the bf16 matrix cores, r04.
//
// The in-tolerance mode keeps fp32 tensors everywhere and ran every attention on the exact-fp32 matrix pipe (attn_f32.h:
// v_mfma_f32_32x32x2_f32, 157 TF/s = 1/16 of the bf16 pipe) — 16.8 of its 69.8 ms per C2 step, most of it the frame ViT's 2 048 x 16
// (frame, head) problems of 50 x 64.  Here those problems take the route the mode's Linears take (gemm.h, vc_x3): every operand is split while
// staging into hi = RNE bf16(x) and lo = bf16(x - hi) planes and each product is three bf16 MFMAs, lo*hi + hi*lo + hi*hi, fp32 accumulate
// (~5.3x the rate of the fp32 pipe for the same contraction; error ~2^-17 per operand, the GEMMs' level).  Probabilities and dS — fp32 values in
// accumulator registers — are split the same way while they are packed into MFMA operands.
// Structure, orientation tricks, masks and the dropout bit packing are those of attn_vit_fwd2_kernel / attn_vit_bwd4_body (attn_mfma.h): two
// waves (forward) / four waves (backward) per (frame, head), tokens padded 50 -> 64; outputs leave as whole 256-byte fp32 rows through a
// wave-private staging tile.  LDS: two planes per operand tile — 36 KiB (forward), 73 KiB (backward) per workgroup.
// PK = true (AttnParams::x3 == 2): q / k / v / dout arrive as PRE-SPLIT words (written by the bf16x3 GEMM epilogues that produce them: unpacked
// here, not split) and o / dq / dk / dv leave pre-split for the GEMMs that consume them — same hi / lo values either way, so the results are
// bit-identical to the fp32-tensor form.
#pragma once
#include "attn_mfma.h"
#include "attn_f32.h"      // AF_PITCH
#include "gemm.h"          // gemm_split4

constexpr int AX_TILE = AM_T * AM_S;                 // elements of one plane
constexpr int AX_STAGE_FLOATS = 32 * AF_PITCH;       // wave-private fp32 store staging: 32 rows x 68 floats = 8 704 bytes

// PK: the tensor holds pre-split hi | lo words (gemm.h vc_pk: written by the producing GEMM epilogue) — unpacked (4 byte-permutes per quad), not split
template <bool PK> VC_DEV void ax_split4(const vc_u32x4& a, vc_u32x2& hi, vc_u32x2& lo) { if (PK) gemm_unpack4(a, hi, lo); else gemm_split4(a, hi, lo); }
// stage one fp32 (or pre-split) [T x 64] head slice as hi / lo bf16 planes with NT threads (rows >= T zero-filled); every load is issued before the first split
template <int NT, bool PK>
VC_DEV void ax_stage_nt(vc_bf16* hi, vc_bf16* lo, const float* g, long ld, int T, int tid) {
    constexpr int N = AM_T * 8 / NT;
    vc_u32x4 v[N][2];
#pragma unroll
    for (int it = 0; it < N; ++it) {
        const int c = tid + NT * it, row = c >> 3, col = (c & 7) * 8;
        const vc_u32x4* s = reinterpret_cast<const vc_u32x4*>(g + (long)(row < T ? row : T - 1) * ld + col);
        v[it][0] = s[0]; v[it][1] = s[1];
    }
#pragma unroll
    for (int it = 0; it < N; ++it) {
        const int c = tid + NT * it, row = c >> 3, col = (c & 7) * 8;
        vc_u32x2 h0, l0, h1, l1;
        ax_split4<PK>(v[it][0], h0, l0); ax_split4<PK>(v[it][1], h1, l1);
        vc_u32x4 wh, wl;
        wh.x = h0.x; wh.y = h0.y; wh.z = h1.x; wh.w = h1.y; wl.x = l0.x; wl.y = l0.y; wl.z = l1.x; wl.w = l1.y;
        if (row >= T) { wh.x = wh.y = wh.z = wh.w = 0u; wl.x = wl.y = wl.z = wl.w = 0u; }
        *reinterpret_cast<vc_u32x4*>(hi + row * AM_S + col) = wh;
        *reinterpret_cast<vc_u32x4*>(lo + row * AM_S + col) = wl;
    }
}
// acc[i] += X[rows i*32..][d] * Y[rows t*32..][d]^T on split operands: small terms first, then hi*hi
VC_DEV void ax_mm_nt1(vc_f32x16 (&acc)[2], const vc_bf16* Xh, const vc_bf16* Xl, const vc_bf16* Yh, const vc_bf16* Yl, int t, int lane) {
#pragma unroll
    for (int ks = 0; ks < AM_D / 16; ++ks) {
        const vc_s16x8 bh = am_frag(Yh, t * 32, ks, lane), bl = am_frag(Yl, t * 32, ks, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const vc_s16x8 ah = am_frag(Xh, i * 32, ks, lane), al = am_frag(Xl, i * 32, ks, lane);
            acc[i] = vc_mfma_32x32x16_bf16(al, bh, acc[i]);
            acc[i] = vc_mfma_32x32x16_bf16(ah, bl, acc[i]);
            acc[i] = vc_mfma_32x32x16_bf16(ah, bh, acc[i]);
        }
    }
}
// accumulator registers 8s..8s+7 of a 32x32 tile (times the dropout keep-multipliers) -> hi / lo bf16 fragments
template <bool DROP>
VC_DEV void ax_pack(const vc_f32x16& a, int s, uint32_t keep, int ti, float scale, vc_s16x8& hi, vc_s16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float w = DROP ? a[8 * s + j] * (((keep >> (ti * 16 + 8 * s + j)) & 1) ? scale : 0.0f) : a[8 * s + j];
        const vc_bf16 h = vc_f32_to_bf16(w);
        hi[j] = (short)h.bits; lo[j] = (short)vc_f32_to_bf16(w - vc_bf16_to_f32(h)).bits;
    }
}
// out[dt][d-row (registers)][token (lane)] += sum over the 64 contracted tokens (register rows of W[tt]) of W * Y[token][d], split operands
template <bool DROP>
VC_DEV void ax_mm_tok1(vc_f32x16 (&out)[2], const vc_f32x16 (&W)[2], const vc_bf16* Yh, const vc_bf16* Yl, int lane, uint32_t keep, float scale) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            vc_s16x8 bh, bl;
            ax_pack<DROP>(W[tt], s, keep, tt, scale, bh, bl);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const vc_s16x8 ah = am_frag_tr(Yh, tt * 32 + 16 * s, dt * 32, lane), al = am_frag_tr(Yl, tt * 32 + 16 * s, dt * 32, lane);
                out[dt] = vc_mfma_32x32x16_bf16(al, bh, out[dt]);
                out[dt] = vc_mfma_32x32x16_bf16(ah, bl, out[dt]);
                out[dt] = vc_mfma_32x32x16_bf16(ah, bh, out[dt]);
            }
        }
}
// rows t*32 .. t*32+31 (< T) of a [token][64] fp32 result whose registers walk the head dim (lane = token row, 4 consecutive columns per
// accumulator quad): transposed through a wave-private 32 x AF_PITCH fp32 staging tile (16-byte LDS writes), then written as whole 256-byte rows
template <bool PK>
VC_DEV void ax_store_rows(float* stage, float* g, long ld, const vc_f32x16 (&acc)[2], int t, int T, int lane, float mul) {
    float* w = stage + (lane & 31) * AF_PITCH + 4 * (lane >> 5);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            vc_u32x4 q;
            if (PK) { q.x = vc_pk_pack(acc[dt][4 * gq] * mul); q.y = vc_pk_pack(acc[dt][4 * gq + 1] * mul);
                      q.z = vc_pk_pack(acc[dt][4 * gq + 2] * mul); q.w = vc_pk_pack(acc[dt][4 * gq + 3] * mul); }
            else { q.x = vc_f32_bits(acc[dt][4 * gq] * mul); q.y = vc_f32_bits(acc[dt][4 * gq + 1] * mul);
                   q.z = vc_f32_bits(acc[dt][4 * gq + 2] * mul); q.w = vc_f32_bits(acc[dt][4 * gq + 3] * mul); }
            *reinterpret_cast<vc_u32x4*>(w + dt * 32 + 8 * gq) = q;
        }
    vc_wave_barrier();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + (lane >> 4), c = (lane & 15) * 4;
        const vc_u32x4 v = *reinterpret_cast<const vc_u32x4*>(stage + row * AF_PITCH + c);
        if (t * 32 + row < T) *reinterpret_cast<vc_u32x4*>(g + (long)(t * 32 + row) * ld + c) = v;
    }
    vc_wave_barrier();
}

// ------------------------------------------------------------------------------------------------------------ forward
template <bool DROP, bool PK>
VC_KERNEL __launch_bounds__(128, 2) void attn_vit_fwd2_x3_kernel(AttnParams p) {
    VC_SHARED __attribute__((aligned(16))) vc_bf16 tiles[4][AX_TILE];          // Q hi, Q lo (then V hi, V lo) ; K hi, K lo (then the store staging)
    const int tid = threadIdx.x, lane = tid & 63, t = vc_uniform(tid >> 6);
    int h; long n; am_block_to_frame_head((int)blockIdx.x, p.B, p.H, n, h);
    const int T = p.Tq;
    const long rowq = n * T;
    ax_stage_nt<128, PK>(tiles[0], tiles[1], (const float*)p.q + rowq * p.ldq + h * AM_D, p.ldq, T, tid);
    ax_stage_nt<128, PK>(tiles[2], tiles[3], (const float*)p.k + rowq * p.ldk + h * AM_D, p.ldk, T, tid);
    // V is requested NOW (into registers) and parked in Q's tiles once S is done: one memory round trip per workgroup instead of two
    constexpr int NV = AM_T * 8 / 128;
    vc_u32x4 vreg[NV][2];
    {
        const float* gv = (const float*)p.v + rowq * p.ldv + h * AM_D;
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int c = tid + 128 * it, row = c >> 3, col = (c & 7) * 8;
            const vc_u32x4* s = reinterpret_cast<const vc_u32x4*>(gv + (long)(row < T ? row : T - 1) * p.ldv + col);
            vreg[it][0] = s[0]; vreg[it][1] = s[1];
        }
    }
    vc_sync();
    const uint32_t dbase0 = (uint32_t)((n * p.H + h) * T) * (uint32_t)T;
    uint32_t keep = 0;
    if (DROP) keep = am_keep_bits1<true>(p.drop, dbase0, T, t, lane);
    vc_f32x16 st[2];                          // S^T[key tile][query tile t], lane column = query
    am_zero1(st);
    ax_mm_nt1(st, tiles[2], tiles[3], tiles[0], tiles[1], t, lane);
    vc_sync();                                // both waves are done with Q (and K): Q's tiles now receive V
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const int c = tid + 128 * it, row = c >> 3, col = (c & 7) * 8;
        vc_u32x2 h0, l0, h1, l1;
        ax_split4<PK>(vreg[it][0], h0, l0); ax_split4<PK>(vreg[it][1], h1, l1);
        vc_u32x4 wh, wl;
        wh.x = h0.x; wh.y = h0.y; wh.z = h1.x; wh.w = h1.y; wl.x = l0.x; wl.y = l0.y; wl.z = l1.x; wl.w = l1.y;
        if (row >= T) { wh.x = wh.y = wh.z = wh.w = 0u; wl.x = wl.y = wl.z = wl.w = 0u; }
        *reinterpret_cast<vc_u32x4*>(tiles[0] + row * AM_S + col) = wh;
        *reinterpret_cast<vc_u32x4*>(tiles[1] + row * AM_S + col) = wl;
    }
    {
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + am_row(r, lane);
                const float sv = (key < T) ? st[kt][r] * p.scale : -INFINITY;
                st[kt][r] = sv; m = fmaxf(m, sv);
            }
        m = fmaxf(m, vc_shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float e = expf(st[kt][r] - m); st[kt][r] = e; l += e; }
        l += vc_shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int query = t * 32 + (lane & 31);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] *= DROP ? inv * am_keep1(keep, kt, r, p.drop.scale) : inv;
        if (p.lse && lane < 32 && query < T) p.lse[(n * p.H + h) * T + query] = m + logf(l);
    }
    vc_sync();                                // V has landed
    vc_f32x16 o[2];
    am_zero1(o);
    ax_mm_tok1<false>(o, st, tiles[0], tiles[1], lane, 0u, 1.0f);        // O[query][d] = sum_key P[query][key] V[key][d]
    ax_store_rows<PK>(reinterpret_cast<float*>(tiles[2]) + t * AX_STAGE_FLOATS, (float*)p.o + rowq * p.ldo + h * AM_D, p.ldo, o, t, T, lane, 1.0f);   // (K is dead since the second barrier)
}

// ------------------------------------------------------------------------------------------------------------ backward
template <bool DROP, bool PK>
VC_KERNEL __launch_bounds__(256, 2) void attn_vit_bwd4_x3_kernel(AttnParams p) {
    VC_DYN_SHARED(vc_bf16, tiles);                                            // Q, K, V, dO: hi and lo plane each (8 planes) + lse / D_i
    float* lse_s = reinterpret_cast<float*>(tiles + 8 * AX_TILE);
    float* del_s = lse_s + AM_T;
    const int tid = threadIdx.x, lane = tid & 63, wave = vc_uniform(tid >> 6);
    int h; long n; am_block_to_frame_head((int)blockIdx.x, p.B, p.H, n, h);
    const int T = p.Tq;
    const long rowq = n * T;
    vc_bf16 *Qh = tiles, *Ql = tiles + AX_TILE, *Kh = tiles + 2 * AX_TILE, *Kl = tiles + 3 * AX_TILE;
    vc_bf16 *Vh = tiles + 4 * AX_TILE, *Vl = tiles + 5 * AX_TILE, *Oh = tiles + 6 * AX_TILE, *Ol = tiles + 7 * AX_TILE;
    ax_stage_nt<256, PK>(Qh, Ql, (const float*)p.q + rowq * p.ldq + h * AM_D, p.ldq, T, tid);
    ax_stage_nt<256, PK>(Kh, Kl, (const float*)p.k + rowq * p.ldk + h * AM_D, p.ldk, T, tid);
    ax_stage_nt<256, PK>(Vh, Vl, (const float*)p.v + rowq * p.ldv + h * AM_D, p.ldv, T, tid);
    ax_stage_nt<256, PK>(Oh, Ol, (const float*)p.dout + rowq * p.lddo + h * AM_D, p.lddo, T, tid);
    if (tid < AM_T) lse_s[tid] = (tid < T) ? p.lse[(n * p.H + h) * T + tid] : 0.f;
    vc_sync();
    const uint32_t dbase0 = (uint32_t)((n * p.H + h) * T) * (uint32_t)T;
    const int t = wave & 1;
    if (wave < 2) {    // ---------------- lane = query of tile t:  D_i, dQ
        uint32_t keep = 0;
        if (DROP) keep = am_keep_bits1<true>(p.drop, dbase0, T, t, lane);
        vc_f32x16 st[2], dpt[2];
        am_zero1(st); am_zero1(dpt);
        ax_mm_nt1(st, Kh, Kl, Qh, Ql, t, lane);      // S^T[key][query]
        ax_mm_nt1(dpt, Vh, Vl, Oh, Ol, t, lane);     // dP^T[key][query]
        const int query = t * 32 + (lane & 31);
        const float lse = lse_s[query];
        float dsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + am_row(r, lane);
                const bool ok = key < T && query < T;
                const float pr = ok ? expf(st[kt][r] * p.scale - lse) : 0.f;
                if (DROP) dpt[kt][r] *= am_keep1(keep, kt, r, p.drop.scale);
                st[kt][r] = pr; dsum += pr * dpt[kt][r];
            }
        dsum += vc_shfl_xor(dsum, 32);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = st[kt][r] * (dpt[kt][r] - dsum);      // dS^T (scale folded into the store)
        if (lane < 32) del_s[query] = dsum;
        if (p.delta && lane < 32 && query < T) p.delta[(n * p.H + h) * T + query] = dsum;
        vc_sync();                           // publish D_i to the key waves; nobody reads V any more: its planes become the store staging
        vc_f32x16 dq[2];
        am_zero1(dq);
        ax_mm_tok1<false>(dq, st, Kh, Kl, lane, 0u, 1.0f);              // dQ[query][d] = sum_key dS[query][key] K[key][d]
        ax_store_rows<PK>(reinterpret_cast<float*>(Vh) + t * AX_STAGE_FLOATS, (float*)p.dq + rowq * p.lddq + h * AM_D, p.lddq, dq, t, T, lane, p.scale);
    } else {           // ---------------- lane = key of tile t:  dV, dK
        uint32_t keep = 0;
        if (DROP) keep = am_keep_bits1<false>(p.drop, dbase0, T, t, lane);
        vc_f32x16 sn[2];
        am_zero1(sn);
        ax_mm_nt1(sn, Qh, Ql, Kh, Kl, t, lane);      // S[query][key]
        const int key = t * 32 + (lane & 31);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int query = qt * 32 + am_row(r, lane);
                sn[qt][r] = (key < T && query < T) ? expf(sn[qt][r] * p.scale - lse_s[query]) : 0.f;       // P
            }
        vc_f32x16 dv[2];
        am_zero1(dv);
        ax_mm_tok1<DROP>(dv, sn, Oh, Ol, lane, keep, p.drop.scale);      // dV[key][d] = sum_query P'[query][key] dO[query][d]
        vc_f32x16 dp[2];
        am_zero1(dp);
        ax_mm_nt1(dp, Oh, Ol, Vh, Vl, t, lane);      // dP'[query][key]
        vc_sync();                           // D_i from the query waves; nobody reads V / dO any more: dO's planes become the store staging
        float* stage = reinterpret_cast<float*>(Oh) + t * AX_STAGE_FLOATS;
        ax_store_rows<PK>(stage, (float*)p.dv + rowq * p.lddv + h * AM_D, p.lddv, dv, t, T, lane, 1.0f);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int query = qt * 32 + am_row(r, lane);
                const float ms = DROP ? am_keep1(keep, qt, r, p.drop.scale) : 1.0f;
                dp[qt][r] = sn[qt][r] * (dp[qt][r] * ms - del_s[query]);                                    // dS
            }
        vc_f32x16 dk[2];
        am_zero1(dk);
        ax_mm_tok1<false>(dk, dp, Qh, Ql, lane, 0u, 1.0f);              // dK[key][d] = sum_query dS[query][key] Q[query][d]
        ax_store_rows<PK>(stage, (float*)p.dk + rowq * p.lddk + h * AM_D, p.lddk, dk, t, T, lane, p.scale);
    }
}
constexpr size_t ax_bwd_lds_bytes() { return (size_t)8 * AX_TILE * 2 + 2 * AM_T * sizeof(float); }
