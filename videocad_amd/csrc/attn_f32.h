// attn_f32.h — fp32 attention on the f32 matrix cores (v_mfma_f32_32x32x2_f32: an exact fp32 FMA chain), r03.
//
// The fp32-storage modes (VCAD_F32 parity mode, VCAD_BF16X3 in-tolerance mode) ran every attention on the wave-per-row kernels of
// attn.h, where each lane walks a whole K row from global memory: 88 of the 203 ms of a C2 step (profiles/r02_bench_f32.json).  These
// kernels keep the same mask rule and the same two-kernel backward split (no atomics, deterministic) but put one wave on a
// 32-query (or 32-key) block and all four contractions on the matrix cores.  They need no workgroup barrier and use LDS only as a wave-private 8.5 KiB transposer for the row tiles:
//
//   * the score product is issued SWAPPED, S^T = K Q^T, with the k-index of one MFMA step paired as (d, d + 32): lane (row, half)
//     supplies elements [half*32 + kk] of ITS OWN K / Q row — 32 consecutive floats in registers (fetched coalesced, af_ld32_rows);
//   * in the S^T accumulator a lane owns one query (column) — softmax statistics are lane-local plus ONE exchange with lane ^ 32;
//   * accumulator register r of a tile holds P[i][j_r] in lanes 0-31 and P[i][j_r + 4] in lanes 32-63 — exactly an MFMA A fragment
//     whose two k-steps are keys (j_r, j_r + 4), so P (and dS) feed the second product straight from the accumulator registers;
//     the matching B fragment V[j_r + 4*half][n] is a coalesced row read (lanes = 32 x 8-byte column pairs);
//   * the backward's key-side kernel recomputes the UNSWAPPED orientation (lane = key) so P^T / dS^T are A fragments for dV / dK.
//
// Element (b, t, h, d) of q/k/v/o lives at base + (b*T + t)*ld + h*D + d (packed projections consumed in place), as in attn.h.
// Masks: key j visible to query i iff max(0, i - window + 1) <= j <= (causal ? i : Tk - 1).  Covers Tq == Tk <= 32*NKT.
#pragma once
#include "attn.h"

constexpr int AF_MAXT = 64;            // NKT = 2 tiles of 32 keys: the ViT (50 tokens) and the decoder at seq_len <= 64

VC_DEV void af_ld32(const float* p, float (&v)[32]) {            // 32 consecutive floats, 16-byte aligned
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const vc_u32x4 c = reinterpret_cast<const vc_u32x4*>(p)[q];
        v[4 * q] = vc_bits_f32(c.x); v[4 * q + 1] = vc_bits_f32(c.y); v[4 * q + 2] = vc_bits_f32(c.z); v[4 * q + 3] = vc_bits_f32(c.w);
    }
}
// The same 32 floats [half*32, half*32 + 32) of row (row0 + lane % 32) of a [rows][64-float chunk] block at g0 — but fetched COALESCED:
// 16 lanes cover one 256-byte row piece, four rows per load instruction, then transposed through a wave-private LDS tile (row pitch 68
// floats: the 16-byte reads of 16 different rows hit 64 different banks).  When every lane read its own row straight from memory a load
// instruction touched 64 different cache lines for 1 KiB; the score phases then ran at 2.5 TB/s and cost 35-40 % of the kernels
// (profiles/r03_attn_rowload_ab.txt).  Rows past nvalid repeat row nvalid - 1 (masked later, never stored).
constexpr int AF_PITCH = 68;
// (two halves: the coalesced loads into eight staging quads, then the trip through the LDS tile into the operand registers)
VC_DEV void af_gload(vc_u32x4 (&stg)[8], const float* g0, long ld, int row0, int nvalid, int lane) {
    const int r = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        int rr = row0 + it * 4 + r; rr = rr < nvalid ? rr : nvalid - 1;
        stg[it] = *reinterpret_cast<const vc_u32x4*>(g0 + (long)rr * ld + c4);
    }
}
VC_DEV void af_transpose(float* tile, const vc_u32x4 (&stg)[8], float (&v)[32], int lane) {
    const int r = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
    for (int it = 0; it < 8; ++it) *reinterpret_cast<vc_u32x4*>(tile + (it * 4 + r) * AF_PITCH + c4) = stg[it];
    vc_wave_barrier();
    af_ld32(tile + (lane & 31) * AF_PITCH + (lane >> 5) * 32, v);
    vc_wave_barrier();                                  // the next tile's writes stay behind these reads
}
VC_DEV void af_ld32_rows(float* tile, const float* g0, long ld, int row0, int nvalid, float (&v)[32], int lane) {
    vc_u32x4 stg[8];
    af_gload(stg, g0, ld, row0, nvalid, lane);
    af_transpose(tile, stg, v, lane);
}
VC_DEV void af_zero(vc_f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
VC_DEV int af_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }     // accumulator row of register r (MFMA 32x32 D layout)
// the 16 B-fragment rows of one 32-row tile: dst[r] = the 8-byte column pair at `col` of row (row0 + af_row(r, half)), clamped to nrows - 1
VC_DEV void af_rows16(vc_u32x2 (&dst)[16], const float* col, long rowbase, long ld, int row0, int half, int nrows, bool skip) {
    if (skip) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int j = row0 + af_row(r, half), jl = j < nrows ? j : nrows - 1;
        dst[r] = *reinterpret_cast<const vc_u32x2*>(col + (rowbase + jl) * ld);
    }
}
// keeps the memory operations above it above (the prefetch of the next work item must be issued BEFORE the current item's MFMAs)
#ifndef VC_EMU
VC_DEV void af_pin() { asm volatile("" ::: "memory"); }
#else
VC_DEV void af_pin() {}
#endif

// ------------------------------------------------------------------------------------------------------------ forward
// one wave per (batch, head, 32-query block)
// FULL: no causal / window mask (the ViT): the per-element visibility test reduces to the padding bound
template <int NCH, int NKT, bool FULL>
VC_KERNEL __launch_bounds__(256, 2) void attn_f32_fwd_kernel(AttnParams p) {
    constexpr int D = 64 * NCH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, il = lane & 31, half = lane >> 5;
    const int nqb = (p.Tq + 31) >> 5;
    const long unit = (long)blockIdx.x * 4 + wave;
    if (unit >= (long)p.B * p.H * nqb) return;
    const int qb = (int)(unit % nqb), h = (int)((unit / nqb) % p.H); const long b = unit / ((long)nqb * p.H);
    const int i = qb * 32 + il, iq = i < p.Tq ? i : p.Tq - 1;          // this lane's query (padding lanes repeat the last one; never stored)
    const int lo = iq - p.window + 1 > 0 ? iq - p.window + 1 : 0, hi = p.causal ? (iq < p.Tk - 1 ? iq : p.Tk - 1) : p.Tk - 1;
    // key tiles this block needs (wave-uniform): tile t holds keys 32t .. 32t+31
    const int i_first = qb * 32, i_last = (qb * 32 + 31 < p.Tq ? qb * 32 + 31 : p.Tq - 1);
    const int j_min = i_first - p.window + 1 > 0 ? i_first - p.window + 1 : 0, j_max = p.causal ? (i_last < p.Tk - 1 ? i_last : p.Tk - 1) : p.Tk - 1;
    VC_SHARED float af_tile[4][32 * AF_PITCH];
    float* tile = af_tile[wave];
    const float* qblk = (const float*)p.q + (b * p.Tq) * p.ldq + (long)h * D;
    const float* kblk = (const float*)p.k + (b * p.Tk) * p.ldk + (long)h * D;
    vc_f32x16 st[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) af_zero(st[t]);
    // (requesting tile n + 1's rows into staging registers before tile n's MFMAs, with the order pinned, was measured: 860 us against 596 us
    // for the ViT forward — the fences that pin it also stop hipcc from interleaving the LDS reads with the matrix-core work; left to the
    // compiler and the second wave on the SIMD)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float qf[32]; af_ld32_rows(tile, qblk + c * 64, p.ldq, qb * 32, p.Tq, qf, lane);
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            if (t * 32 > j_max || t * 32 + 31 < j_min) continue;
            float kf[32]; af_ld32_rows(tile, kblk + c * 64, p.ldk, t * 32, p.Tk, kf, lane);
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) st[t] = vc_mfma_32x32x2_f32(kf[kk], qf[kk], st[t]);
        }
    }
    vc_u32x2 vb[2][16];
    af_rows16(vb[0], (const float*)p.v + (long)h * D + 2 * il, b * p.Tk, p.ldv, 0, half, p.Tk, 0 > j_max || 31 < j_min);
    af_pin();
    // softmax over this lane's query: its 16*NKT keys here + the partner half's
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = t * 32 + af_row(r, half);
            const float s = (FULL ? j < p.Tk : (j >= lo && j <= hi)) ? vc_mul_rn(st[t][r], p.scale) : -INFINITY;     // rounded on its own: the backward forms the same product (see there)
            st[t][r] = s; m = fmaxf(m, s);
        }
    m = fmaxf(m, vc_shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float e = vc_expf_fast(st[t][r] - m); st[t][r] = e; l += e; }       // v_exp_f32 (exp(-inf) = 0 for masked keys, exp(0) = 1 exactly)
    l += vc_shfl_xor(l, 32);
    const float inv = 1.0f / l;
    const long dbase = ((b * p.H + h) * p.Tq + iq) * p.Tk;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float pr = st[t][r] * inv;
            if (p.drop.key) { const int j = t * 32 + af_row(r, half); if (j < p.Tk) pr *= vc_drop_mul(p.drop, (uint32_t)(dbase + j)); }
            st[t][r] = pr;
        }
    if (p.lse && half == 0 && i < p.Tq) p.lse[(b * p.H + h) * p.Tq + i] = m + logf(l);
    // O[i][d] = sum_j P[i][j] V[j][d]: A = P from the accumulator registers, B = V rows (lane = column pair 2*il, 2*il + 1 of the chunk).
    // Work items (chunk, key tile): the 16 row reads of item n + 1 are in flight while item n's 32 MFMAs run (the first item's were
    // requested before the softmax) — with one read per MFMA pair a wave waits out a memory round trip per pair.
    vc_f32x16 o0, o1; af_zero(o0); af_zero(o1);
#pragma unroll
    for (int it = 0; it < NCH * NKT; ++it) {
        const int c = it / NKT, t = it % NKT;
        if (it + 1 < NCH * NKT) af_rows16(vb[(it + 1) & 1], (const float*)p.v + (long)h * D + ((it + 1) / NKT) * 64 + 2 * il, b * p.Tk, p.ldv, ((it + 1) % NKT) * 32, half, p.Tk, ((it + 1) % NKT) * 32 > j_max || ((it + 1) % NKT) * 32 + 31 < j_min);
        af_pin();
        if (!(t * 32 > j_max || t * 32 + 31 < j_min)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o0 = vc_mfma_32x32x2_f32(st[t][r], vc_bits_f32(vb[it & 1][r].x), o0);
                o1 = vc_mfma_32x32x2_f32(st[t][r], vc_bits_f32(vb[it & 1][r].y), o1);
            }
        }
        if (t == NKT - 1) {
            float* ocol = (float*)p.o + (long)h * D + c * 64 + 2 * il;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int io = qb * 32 + af_row(r, half);
                if (io < p.Tq) { vc_u32x2 w; w.x = vc_f32_bits(o0[r]); w.y = vc_f32_bits(o1[r]); *reinterpret_cast<vc_u32x2*>(ocol + (b * p.Tq + io) * p.ldo) = w; }
            }
            af_zero(o0); af_zero(o1);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ backward, query side
// one wave per (batch, head, 32-query block): D_i = sum_j P_ij dP_ij -> delta,  dq_i = scale * sum_j dS_ij k_j
template <int NCH, int NKT, bool FULL>
VC_KERNEL __launch_bounds__(256, 2) void attn_f32_bwd_q_kernel(AttnParams p) {
    constexpr int D = 64 * NCH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, il = lane & 31, half = lane >> 5;
    const int nqb = (p.Tq + 31) >> 5;
    const long unit = (long)blockIdx.x * 4 + wave;
    if (unit >= (long)p.B * p.H * nqb) return;
    const int qb = (int)(unit % nqb), h = (int)((unit / nqb) % p.H); const long b = unit / ((long)nqb * p.H);
    const int i = qb * 32 + il, iq = i < p.Tq ? i : p.Tq - 1;
    const int lo = iq - p.window + 1 > 0 ? iq - p.window + 1 : 0, hi = p.causal ? (iq < p.Tk - 1 ? iq : p.Tk - 1) : p.Tk - 1;
    const int i_first = qb * 32, i_last = (qb * 32 + 31 < p.Tq ? qb * 32 + 31 : p.Tq - 1);
    const int j_min = i_first - p.window + 1 > 0 ? i_first - p.window + 1 : 0, j_max = p.causal ? (i_last < p.Tk - 1 ? i_last : p.Tk - 1) : p.Tk - 1;
    VC_SHARED float af_tile[4][32 * AF_PITCH];
    float* tile = af_tile[wave];
    const float* qblk = (const float*)p.q + (b * p.Tq) * p.ldq + (long)h * D;
    const float* doblk = (const float*)p.dout + (b * p.Tq) * p.lddo + (long)h * D;
    const float* kblk = (const float*)p.k + (b * p.Tk) * p.ldk + (long)h * D;
    const float* vblk = (const float*)p.v + (b * p.Tk) * p.ldv + (long)h * D;
    vc_f32x16 st[NKT], dpt[NKT];          // S^T and dP^T = V dO^T, same layout
#pragma unroll
    for (int t = 0; t < NKT; ++t) { af_zero(st[t]); af_zero(dpt[t]); }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float qf[32], df[32]; af_ld32_rows(tile, qblk + c * 64, p.ldq, qb * 32, p.Tq, qf, lane); af_ld32_rows(tile, doblk + c * 64, p.lddo, qb * 32, p.Tq, df, lane);
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            if (t * 32 > j_max || t * 32 + 31 < j_min) continue;
            float kf[32]; af_ld32_rows(tile, kblk + c * 64, p.ldk, t * 32, p.Tk, kf, lane);
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) st[t] = vc_mfma_32x32x2_f32(kf[kk], qf[kk], st[t]);
            float vf[32]; af_ld32_rows(tile, vblk + c * 64, p.ldv, t * 32, p.Tk, vf, lane);
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) dpt[t] = vc_mfma_32x32x2_f32(vf[kk], df[kk], dpt[t]);
        }
    }
    vc_u32x2 kb[2][16];
    af_rows16(kb[0], (const float*)p.k + (long)h * D + 2 * il, b * p.Tk, p.ldk, 0, half, p.Tk, 0 > j_max || 31 < j_min);
    af_pin();
    const float lse = p.lse[(b * p.H + h) * p.Tq + iq];
    const long dbase = ((b * p.H + h) * p.Tq + iq) * p.Tk;
    float dsum = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = t * 32 + af_row(r, half);
            const bool vis = FULL ? j < p.Tk : (j >= lo && j <= hi);
            const float pr = vis ? vc_expf_fast(vc_mul_rn(st[t][r], p.scale) - lse) : 0.f;     // the forward's rounded product: a query with ONE visible key gets P = 1 and dS = 0 exactly, as in the reference
            float dp = vis ? dpt[t][r] : 0.f;
            if (p.drop.key && vis) dp = vc_mul_rn(dp, vc_drop_mul(p.drop, (uint32_t)(dbase + j)));        // dP = dP' * mask
            st[t][r] = pr; dpt[t][r] = dp; dsum += pr * dp;
        }
    dsum += vc_shfl_xor(dsum, 32);
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[t][r] = st[t][r] * (dpt[t][r] - dsum) * p.scale;      // dS (scale folded in)
    if (half == 0 && i < p.Tq) p.delta[(b * p.H + h) * p.Tq + i] = dsum;
    vc_f32x16 o0, o1; af_zero(o0); af_zero(o1);
#pragma unroll
    for (int it = 0; it < NCH * NKT; ++it) {          // work items (chunk, key tile), K rows of item n + 1 in flight during item n's MFMAs
        const int c = it / NKT, t = it % NKT;
        if (it + 1 < NCH * NKT) af_rows16(kb[(it + 1) & 1], (const float*)p.k + (long)h * D + ((it + 1) / NKT) * 64 + 2 * il, b * p.Tk, p.ldk, ((it + 1) % NKT) * 32, half, p.Tk, ((it + 1) % NKT) * 32 > j_max || ((it + 1) % NKT) * 32 + 31 < j_min);
        af_pin();
        if (!(t * 32 > j_max || t * 32 + 31 < j_min)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o0 = vc_mfma_32x32x2_f32(st[t][r], vc_bits_f32(kb[it & 1][r].x), o0);
                o1 = vc_mfma_32x32x2_f32(st[t][r], vc_bits_f32(kb[it & 1][r].y), o1);
            }
        }
        if (t == NKT - 1) {
            float* dqcol = (float*)p.dq + (long)h * D + c * 64 + 2 * il;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int io = qb * 32 + af_row(r, half);
                if (io < p.Tq) { vc_u32x2 w; w.x = vc_f32_bits(o0[r]); w.y = vc_f32_bits(o1[r]); *reinterpret_cast<vc_u32x2*>(dqcol + (b * p.Tq + io) * p.lddq) = w; }
            }
            af_zero(o0); af_zero(o1);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ backward, key side
// one wave per (batch, head, 32-key block): dk_j = scale * sum_i dS_ij q_i,  dv_j = sum_i P'_ij dO_i   (P' = dropped probabilities)
// NQT = query tiles of 32.  Unswapped orientation: accumulator column = key (lane), row = query.
template <int NCH, int NQT, bool FULL>
VC_KERNEL __launch_bounds__(256, 2) void attn_f32_bwd_kv_kernel(AttnParams p) {
    constexpr int D = 64 * NCH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, il = lane & 31, half = lane >> 5;
    const int nkb = (p.Tk + 31) >> 5;
    const long unit = (long)blockIdx.x * 4 + wave;
    if (unit >= (long)p.B * p.H * nkb) return;
    const int kb = (int)(unit % nkb), h = (int)((unit / nkb) % p.H); const long b = unit / ((long)nkb * p.H);
    const int j = kb * 32 + il, jk = j < p.Tk ? j : p.Tk - 1;          // this lane's key
    // queries that can see any key of this block: i in [i_min, i_max]
    const int j_first = kb * 32, j_last = (kb * 32 + 31 < p.Tk ? kb * 32 + 31 : p.Tk - 1);
    const int i_min = p.causal ? j_first : 0;
    int i_max = j_last + p.window - 1; if (i_max > p.Tq - 1) i_max = p.Tq - 1;
    VC_SHARED float af_tile[4][32 * AF_PITCH];
    float* tile = af_tile[wave];
    const float* qblk = (const float*)p.q + (b * p.Tq) * p.ldq + (long)h * D;
    const float* doblk = (const float*)p.dout + (b * p.Tq) * p.lddo + (long)h * D;
    const float* kblk = (const float*)p.k + (b * p.Tk) * p.ldk + (long)h * D;
    const float* vblk = (const float*)p.v + (b * p.Tk) * p.ldv + (long)h * D;
    vc_f32x16 s[NQT], dp[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) { af_zero(s[t]); af_zero(dp[t]); }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float kf[32], vf[32]; af_ld32_rows(tile, kblk + c * 64, p.ldk, kb * 32, p.Tk, kf, lane); af_ld32_rows(tile, vblk + c * 64, p.ldv, kb * 32, p.Tk, vf, lane);
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            if (t * 32 > i_max || t * 32 + 31 < i_min) continue;
            float qf[32]; af_ld32_rows(tile, qblk + c * 64, p.ldq, t * 32, p.Tq, qf, lane);
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) s[t] = vc_mfma_32x32x2_f32(qf[kk], kf[kk], s[t]);
            float df[32]; af_ld32_rows(tile, doblk + c * 64, p.lddo, t * 32, p.Tq, df, lane);
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) dp[t] = vc_mfma_32x32x2_f32(df[kk], vf[kk], dp[t]);
        }
    }
    vc_u32x2 rb[2][16];
    af_rows16(rb[0], (const float*)p.dout + (long)h * D + 2 * il, b * p.Tq, p.lddo, 0, half, p.Tq, 0 > i_max || 31 < i_min);
    af_pin();
    // P'^T (-> s) and dS^T (-> dp): row = query i_r, column = this lane's key j
#pragma unroll
    for (int t = 0; t < NQT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = t * 32 + af_row(r, half);
            bool vis = i < p.Tq && j < p.Tk;
            if constexpr (!FULL) { const int lo = i - p.window + 1 > 0 ? i - p.window + 1 : 0, hi = p.causal ? (i < p.Tk - 1 ? i : p.Tk - 1) : p.Tk - 1; vis = vis && j >= lo && j <= hi; }
            float pr = 0.f, ds = 0.f;
            if (vis) {
                const long sidx = (b * p.H + h) * p.Tq + i;
                pr = vc_expf_fast(vc_mul_rn(s[t][r], p.scale) - p.lse[sidx]);
                const float ms = p.drop.key ? vc_drop_mul(p.drop, (uint32_t)(sidx * p.Tk + j)) : 1.0f;
                ds = pr * (vc_mul_rn(dp[t][r], ms) - p.delta[sidx]) * p.scale;        // (the query-side kernel rounded dP * mask before it summed D_i)
                pr *= ms;
            }
            s[t][r] = pr; dp[t][r] = ds;
        }
    // work items (chunk, product, query tile): dV = P'^T dO (B rows = dO), then dK = dS^T Q (B rows = Q); rows of item n + 1 in flight
    // during item n's 32 MFMAs
    vc_f32x16 a0, a1; af_zero(a0); af_zero(a1);
    constexpr int NIT = NCH * 2 * NQT;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it / (2 * NQT), pr = (it / NQT) & 1, t = it % NQT;
        if (it + 1 < NIT) {
            const int c1 = (it + 1) / (2 * NQT), pr1 = ((it + 1) / NQT) & 1, t1 = (it + 1) % NQT;
            af_rows16(rb[(it + 1) & 1], (pr1 ? (const float*)p.q : (const float*)p.dout) + (long)h * D + c1 * 64 + 2 * il, b * p.Tq, pr1 ? p.ldq : p.lddo, t1 * 32, half, p.Tq,
                      t1 * 32 > i_max || t1 * 32 + 31 < i_min);
        }
        af_pin();
        if (!(t * 32 > i_max || t * 32 + 31 < i_min)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float av = pr ? dp[t][r] : s[t][r];
                a0 = vc_mfma_32x32x2_f32(av, vc_bits_f32(rb[it & 1][r].x), a0);
                a1 = vc_mfma_32x32x2_f32(av, vc_bits_f32(rb[it & 1][r].y), a1);
            }
        }
        if (t == NQT - 1) {
            float* ocol = (pr ? (float*)p.dk : (float*)p.dv) + (long)h * D + c * 64 + 2 * il;
            const long ldo_ = pr ? p.lddk : p.lddv;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jo = kb * 32 + af_row(r, half);
                if (jo < p.Tk) { vc_u32x2 w; w.x = vc_f32_bits(a0[r]); w.y = vc_f32_bits(a1[r]); *reinterpret_cast<vc_u32x2*>(ocol + (b * p.Tk + jo) * ldo_) = w; }
            }
            af_zero(a0); af_zero(a1);
        }
    }
}
