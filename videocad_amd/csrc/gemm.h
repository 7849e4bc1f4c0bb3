// gemm.h — MFMA GEMM for every Linear on the hot path (forward, dgrad, wgrad).
//
//   C[m,n] = epilogue( sum_k opA(m,k) * opB(n,k) )
//   opA(m,k) = TRA ? A[k*lda + m] : A[m*lda + k]       opB(n,k) = TRB ? B[k*ldb + n] : B[n*ldb + k]
//
//   forward  Y = X W^T (+b)        : A = X [M,K],  B = W [N,K]            (TRA=0, TRB=0)   nn.Linear
//   dgrad    dX = dY W             : A = dY [M,N'], B = W [N',K'] read TRB=1 (or the W^T shadow with TRB=0)
//   wgrad    dW = dY^T X           : A = dY [tok,N'] TRA=1, B = X [tok,K'] TRB=1, reduction over tokens
//
// Compute types: CT=float  -> v_mfma_f32_32x32x2_f32 (exact fp32 fma chain: the parity mode)
//                CT=vc_bf16 -> v_mfma_f32_32x32x16_bf16 (fp32 accumulate: the throughput mode)
//                CT=vc_x3   -> "bf16x3" (r03): fp32 operands are split while staging into hi = bf16(x) and lo = bf16(x - hi)
//                              (two LDS planes per operand) and every product runs as three bf16 MFMAs, lo*hi + hi*lo + hi*hi,
//                              into the fp32 accumulator: |error| per product ~2^-16 relative (the dropped lo*lo term and the
//                              16-bit operand representation) against 2^-9 for plain bf16 — the in-tolerance mode (logits within
//                              1e-3 of the fp32 reference) at 3/16 of the f32-MFMA cost
// Source types SA/SB may be float while CT is bf16 (converted while staging into LDS), so the fp32
// residual stream / fp32 residual-gradients feed bf16 MFMA without an extra HBM round trip.
//
// Tile: 128x128 per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA 32x32 tiles, 64 fp32
// accumulator registers), BK = 64 (bf16) / 32 (f32), LDS double-buffered, one barrier per K-tile, global
// loads for tile t+1 issued before the MFMAs of tile t (register staging).  LDS rows are padded by one
// 16-byte slot (bf16) / one dword (f32) so the ds_read_b128 / ds_read_b32 fragment reads are conflict-free.
#pragma once
#include "vc_rt.h"
#include <type_traits>

enum { VC_ACT_NONE = 0, VC_ACT_GELU = 1, VC_ACT_RELU = 2, VC_ACT_TANH = 3, VC_ACT_GELU_FAST = 4 /* internal: bf16-mode GELU */ };

struct GemmParams {
    const void* A; const void* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    int vecA, vecB;                 // 16-byte vector loads legal for A / B (host-checked alignment)
    int vecC;                       // every epilogue tensor (C, residual, aux, dact_src, rowadd) allows 4-column vector rows
    // split-K: gridDim.z slices of k_per_split; partials (fp32, [z][M][N]) go to `partial`, epilogue runs in the reducer
    int k_per_split; float* partial;
    // epilogue (applied in this order): v = alpha*acc (+bias[n]) (+rowadd[row(m)][n]); aux[m,n]=v; v=act(v); v*=dropmask(m*N+n); v*=dact(src[m,n]); v+=residual[m,n]
    float alpha;
    const float* bias;
    const float* rowadd; int rowadd_div; int rowadd_mod; long ld_rowadd;   // row = rowadd_mod ? m % div : m / div
    void* aux; long ldaux;          // type TO
    int act;
    const void* dact_src; long lddact; int dact_kind;                     // type TO
    const float* residual; long ldr;                                      // fp32, may alias C
    vc_drop drop;                                                         // applied after act, before dact / residual (key 0 = off)
    int stagger;                                                          // first-wave start offset in units of s_sleep(127) (~3.4 us); 0 = off
    int n_group;                                                          // register-staged kernel: tile columns per sweep (0 = all: n fastest over the whole width); see gemm_tile_program
    int debug_skip;                                                       // ablation only (tools/gemm_ablate.py): 1 = no global loads in the loop, 2 = no LDS stores, 4 = no MFMA
    // gemm_mid.h only (r06): `batch` independent problems of this shape in one grid (blockIdx.y), problem b at A + b bsa, B + b bsb, C + b bsc (elements) —
    // the per-head projections around the class-token attention (attn_cls.h).  0 / 1 = a single problem.
    int batch; long bsa, bsb, bsc;
};

VC_DEV float vc_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
VC_DEV float vc_dgelu(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}
// bf16-mode GELU: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 + fp32 rounding — three orders below the bf16 rounding of
// the value it feeds) — one exp, one reciprocal, five FMAs instead of libm's erff; the fp32 parity mode keeps erff.
// e = exp(-x^2/2) is shared by erf(x/sqrt2) and the Gaussian term of the derivative.
VC_DEV float vc_erf_sqrt2_fast(float x, float e) {          // erf(x / sqrt(2)) given e = exp(-x*x/2)
    const float ax = fabsf(x) * 0.70710678118654752f;
    const float t = 1.0f / (1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * e;
    return x < 0.0f ? -r : r;
}
VC_DEV float vc_gelu_fast(float x) { const float e = vc_expf_fast(-0.5f * x * x); return 0.5f * x * (1.0f + vc_erf_sqrt2_fast(x, e)); }
VC_DEV float vc_dgelu_fast(float x) {
    const float e = vc_expf_fast(-0.5f * x * x);
    return 0.5f * (1.0f + vc_erf_sqrt2_fast(x, e)) + x * 0.3989422804014327f * e;
}

VC_DEV float vc_apply_act(float v, int act) {
    if (act == VC_ACT_GELU_FAST) return vc_gelu_fast(v);
    if (act == VC_ACT_GELU) return vc_gelu(v);
    if (act == VC_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == VC_ACT_TANH) return tanhf(v);
    return v;
}
VC_DEV float vc_apply_dact(float v, float s, int kind) {
    if (kind == VC_ACT_GELU_FAST) return v * vc_dgelu_fast(s);
    if (kind == VC_ACT_GELU) return v * vc_dgelu(s);
    if (kind == VC_ACT_RELU) return (s > 0.0f) ? v : 0.0f;
    if (kind == VC_ACT_TANH) return v * (1.0f - s * s);
    return v;
}

// single-element epilogue (split-K reducer and edge tiles)
template <typename TO>
VC_DEV void gemm_epilogue_store(const GemmParams& p, int m, int n, float acc, float bias_n) {
    float v = p.alpha * acc + bias_n;
    if (p.rowadd) {
        int r = p.rowadd_mod ? (m % p.rowadd_div) : (m / p.rowadd_div);
        v += p.rowadd[(long)r * p.ld_rowadd + n];
    }
    if (p.aux) vc_st(((TO*)p.aux) + (long)m * p.ldaux + n, v);
    v = vc_apply_act(v, p.act);
    if (p.drop.key) v *= vc_drop_mul(p.drop, (long)m * p.N + n);
    if (p.dact_src) v = vc_apply_dact(v, vc_ld(((const TO*)p.dact_src) + (long)m * p.lddact + n), p.dact_kind);
    if (p.residual) v += p.residual[(long)m * p.ldr + n];
    vc_st(((TO*)p.C) + (long)m * p.ldc + n, v);
}

// one 32x32 accumulator tile of an INTERIOR block: every optional input is loaded for all 16 rows first (loads in
// flight together), then the math, then the 16 stores.  mrow(r) = mbase + (r&3) + 8*(r>>2).
template <typename TO>
VC_DEV void gemm_epilogue_tile(const GemmParams& p, int mbase, int n, const vc_f32x16& acc, float bias_n) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = p.alpha * acc[r] + bias_n;
    if (p.rowadd) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mbase + (r & 3) + 8 * (r >> 2);
            const int rr = p.rowadd_mod ? (m % p.rowadd_div) : (m / p.rowadd_div);
            v[r] += p.rowadd[(long)rr * p.ld_rowadd + n];
        }
    }
    if (p.aux) {
#pragma unroll
        for (int r = 0; r < 16; ++r) vc_st(((TO*)p.aux) + (long)(mbase + (r & 3) + 8 * (r >> 2)) * p.ldaux + n, v[r]);
    }
    if (p.act) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = vc_apply_act(v[r], p.act);
    }
    if (p.drop.key) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] *= vc_drop_mul(p.drop, (long)(mbase + (r & 3) + 8 * (r >> 2)) * p.N + n);
    }
    if (p.dact_src) {
        float sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = vc_ld(((const TO*)p.dact_src) + (long)(mbase + (r & 3) + 8 * (r >> 2)) * p.lddact + n);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = vc_apply_dact(v[r], sv[r], p.dact_kind);
    }
    if (p.residual) {
        float q[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] = p.residual[(long)(mbase + (r & 3) + 8 * (r >> 2)) * p.ldr + n];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += q[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) vc_st(((TO*)p.C) + (long)(mbase + (r & 3) + 8 * (r >> 2)) * p.ldc + n, v[r]);
}

// 4-column vector accesses for the row-wise epilogue
VC_DEV void quad_ld_f32(const float* p, float* v) {
    const vc_u32x4 q = *reinterpret_cast<const vc_u32x4*>(p);
    v[0] = vc_bits_f32(q.x); v[1] = vc_bits_f32(q.y); v[2] = vc_bits_f32(q.z); v[3] = vc_bits_f32(q.w);
}
template <typename T> VC_DEV void quad_ld(const T* p, float* v);
template <> VC_DEV void quad_ld<float>(const float* p, float* v) { quad_ld_f32(p, v); }
template <> VC_DEV void quad_ld<vc_bf16>(const vc_bf16* p, float* v) {
    const vc_u32x2 q = *reinterpret_cast<const vc_u32x2*>(p);
    v[0] = vc_lo16_f32(q.x); v[1] = vc_hi16_f32(q.x); v[2] = vc_lo16_f32(q.y); v[3] = vc_hi16_f32(q.y);
}
template <> VC_DEV void quad_ld<vc_pk>(const vc_pk* p, float* v) {          // hi + lo of each pre-split word
    const vc_u32x4 q = *reinterpret_cast<const vc_u32x4*>(p);
    v[0] = vc_bits_f32(q.x & 0xFFFF0000u) + vc_bits_f32(q.x << 16); v[1] = vc_bits_f32(q.y & 0xFFFF0000u) + vc_bits_f32(q.y << 16);
    v[2] = vc_bits_f32(q.z & 0xFFFF0000u) + vc_bits_f32(q.z << 16); v[3] = vc_bits_f32(q.w & 0xFFFF0000u) + vc_bits_f32(q.w << 16);
}
template <typename T> VC_DEV void quad_st(T* p, const float* v);
template <> VC_DEV void quad_st<float>(float* p, const float* v) {
    vc_u32x4 q; q.x = vc_f32_bits(v[0]); q.y = vc_f32_bits(v[1]); q.z = vc_f32_bits(v[2]); q.w = vc_f32_bits(v[3]);
    *reinterpret_cast<vc_u32x4*>(p) = q;
}
template <> VC_DEV void quad_st<vc_bf16>(vc_bf16* p, const float* v) {
    vc_u32x2 q; q.x = vc_pack_bf16x2(v[0], v[1]); q.y = vc_pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<vc_u32x2*>(p) = q;
}
// bf16x3 GEMM output consumed only by other bf16x3 GEMMs / attention kernels: written PRE-SPLIT (hi | lo words, vc_rt.h vc_pk_pack — exactly the
// two values a consumer's staging split would produce), so the consumers unpack (4 byte-permutes per quad) instead of splitting (10 VALU per quad)
// every time a tile of it is staged — N / BN times per element for an A operand (r04)
template <> VC_DEV void quad_st<vc_pk>(vc_pk* p, const float* v) {
    vc_u32x4 q; q.x = vc_pk_pack(v[0]); q.y = vc_pk_pack(v[1]); q.z = vc_pk_pack(v[2]); q.w = vc_pk_pack(v[3]);
    *reinterpret_cast<vc_u32x4*>(p) = q;
}

// The fused epilogue on four consecutive columns n..n+3 of row m (16-byte / 8-byte accesses): alpha, bias, row-broadcast add,
// pre-activation side output, activation, dropout, activation-derivative, residual, store.
template <typename TO>
VC_DEV void gemm_epilogue_quad(const GemmParams& p, int m, int n, float (&v)[4], const float (&b4)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = p.alpha * v[k] + b4[k];
    if (p.rowadd) {
        const int rr = p.rowadd_mod ? (m % p.rowadd_div) : (m / p.rowadd_div);
        float a4[4]; quad_ld_f32(p.rowadd + (long)rr * p.ld_rowadd + n, a4);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += a4[k];
    }
    if (p.aux) quad_st<TO>(((TO*)p.aux) + (long)m * p.ldaux + n, v);
    if (p.act) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = vc_apply_act(v[k], p.act);
    }
    if (p.drop.key) {
        { float dm[4]; vc_drop_mul4(p.drop, (uint32_t)((long)m * p.N + n), dm);          // N % 4 == 0, n % 4 == 0: even index
          for (int k = 0; k < 4; ++k) v[k] *= dm[k]; }
    }
    if (p.dact_src) {
        float s4[4]; quad_ld<TO>(((const TO*)p.dact_src) + (long)m * p.lddact + n, s4);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = vc_apply_dact(v[k], s4[k], p.dact_kind);
    }
    if (p.residual) {
        float q4[4]; quad_ld_f32(p.residual + (long)m * p.ldr + n, q4);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += q4[k];
    }
    quad_st<TO>(((TO*)p.C) + (long)m * p.ldc + n, v);
}

struct vc_x3 { uint16_t bits; };      // compute-type tag of the bf16x3 mode: LDS elements are bf16, two planes (hi, lo) per operand tile
template <typename CT> struct GemmCfg;
template <> struct GemmCfg<float> {
    static constexpr int BK = 32, CHUNK = 4, STRIDE = 33, KSTEP = 2, PLANES = 1;
};
template <> struct GemmCfg<vc_bf16> {
    static constexpr int BK = 64, CHUNK = 8, STRIDE = 72, KSTEP = 16, PLANES = 1;
};
// BK = 32 and UNPADDED planes (k-contiguous image: 64-byte rows, the 16-byte chunk index XOR-ed with bits 2-3 of the row; row-contiguous
// image of a 128-row tile: 256-byte k-rows, the column XOR-ed with (k & 3) * 32 — both conflict-free for their fragment reads) keep the
// double-buffered stage set of a 128 x 128 tile at 64 KiB, under the 66 KiB epilogue tile: two workgroups share a CU's 160 KiB.  (The
// first version padded the rows — 80 KiB per workgroup, one workgroup per CU, every phase of the loop exposed: 150 TF/s in the model.)
template <> struct GemmCfg<vc_x3> {
    static constexpr int BK = 32, CHUNK = 4, STRIDE = 32, KSTEP = 16, PLANES = 2;      // a staged chunk = ONE 16-byte fp32 quad (whole 128-byte lines per 8 lanes)
};
// hi / lo split of two fp32 values into packed bf16 pairs: hi = RNE(x), lo = RNE(x - hi) (exact subtraction)
VC_DEV void gemm_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = vc_pack_bf16x2(a, b);
    lo = vc_pack_bf16x2(a - vc_bits_f32(hi << 16), b - vc_bits_f32(hi & 0xFFFF0000u));
}
// the same for one 16-byte quad, written on 2-element vectors so that hipcc emits v_cvt_pk_bf16_f32 / v_pk_add_f32 on the (x0, x1), (x2, x3)
// register pairs the load delivered: 10 VALU instructions per quad.  Left to its own pairing (it chose (x0, x2), (x1, x3)) the compiler
// spent 16-20 on moves and sub-dword ORs — and the k-loop of the bf16x3 kernel is VALU-bound (281 VALU per 24 MFMAs, profiles/r03_x3_pmc.md)
VC_DEV void gemm_split4(const vc_u32x4& a, vc_u32x2& hi, vc_u32x2& lo) {
#ifndef VC_EMU
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 h2 __attribute__((ext_vector_type(2)));
    const f2 x01 = {vc_bits_f32(a.x), vc_bits_f32(a.y)}, x23 = {vc_bits_f32(a.z), vc_bits_f32(a.w)};
    hi.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(x01, h2));
    hi.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(x23, h2));
    const f2 t01 = {vc_bits_f32(hi.x << 16), vc_bits_f32(hi.x & 0xFFFF0000u)}, t23 = {vc_bits_f32(hi.y << 16), vc_bits_f32(hi.y & 0xFFFF0000u)};
    lo.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(x01 - t01, h2));
    lo.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(x23 - t23, h2));
#else
    gemm_split2(vc_bits_f32(a.x), vc_bits_f32(a.y), hi.x, lo.x); gemm_split2(vc_bits_f32(a.z), vc_bits_f32(a.w), hi.y, lo.y);
#endif
}
// Pre-split operand (bf16x3 engines keep their weights this way, engine.hip): one 32-bit word per element, hi bf16 in the upper half and lo
// bf16 in the lower — the same two values gemm_split2 produces, so a GEMM gives bit-identical results either way; staging such a tile costs
// four byte-permutes per quad instead of the ~10 VALU of the split (the bf16x3 kernel is VALU-bound: profiles/r03_x3_pmc.md).
VC_DEV void gemm_unpack4(const vc_u32x4& a, vc_u32x2& hi, vc_u32x2& lo) {
#ifndef VC_EMU
    hi.x = __builtin_amdgcn_perm(a.y, a.x, 0x07060302u); hi.y = __builtin_amdgcn_perm(a.w, a.z, 0x07060302u);
    lo.x = __builtin_amdgcn_perm(a.y, a.x, 0x05040100u); lo.y = __builtin_amdgcn_perm(a.w, a.z, 0x05040100u);
#else
    hi.x = (a.x >> 16) | (a.y & 0xFFFF0000u); hi.y = (a.z >> 16) | (a.w & 0xFFFF0000u);
    lo.x = (a.x & 0xFFFFu) | (a.y << 16); lo.y = (a.z & 0xFFFFu) | (a.w << 16);
#endif
}
struct gemm_true { static constexpr bool value = true; };
struct gemm_false { static constexpr bool value = false; };
// pipeline-stage ablation switches (tools/gemm_ablate*.py) exist only in the -DVCAD_AB build used by tools/; the shipped kernels carry no such branches
#ifdef VCAD_AB
#define VC_ABL(bit) (p.debug_skip & (bit))
#else
#define VC_ABL(bit) false
#endif
constexpr int GEMM_THREADS = 256;      // 4 waves as 2 x 2; each wave owns WT x WT MFMA 32x32 tiles => block tile = (64*WT)^2

// LDS image of one operand tile:
//   direct   (k contiguous in memory)       : [128 rows][BK + pad]           fragment = one ds_read_b128
//   bf16 TR  (row dim contiguous in memory) : [BK k-rows][128 + 32 pad]      natural layout, coalesced ds_write_b128;
//            fragment = two ds_read_b64_tr_b16 (hardware transpose); the 64-byte row pad puts the 4 k-rows a
//            transpose-read touches on disjoint banks
//   f32  TR  : transposed on the way into LDS (scalar ds_write_b32), same image as direct
template <typename CT> struct gemm_is_x3 { static constexpr bool value = false; };
template <> struct gemm_is_x3<vc_x3> { static constexpr bool value = true; };
template <typename CT, int ROWS> constexpr bool gemm_tswz() { return gemm_is_x3<CT>::value && ROWS == 128; }      // row-contiguous image without padding (XOR swizzle)
template <typename CT, int ROWS> constexpr int gemm_tstride() { return gemm_tswz<CT, ROWS>() ? ROWS : ROWS + 32; }
template <typename CT, bool TR, int ROWS> constexpr int gemm_plane_elems() {
    return (sizeof(CT) == 2 && TR) ? GemmCfg<CT>::BK * gemm_tstride<CT, ROWS>() : ROWS * GemmCfg<CT>::STRIDE;
}
template <typename CT, bool TR, int ROWS> constexpr int gemm_tile_elems() { return GemmCfg<CT>::PLANES * gemm_plane_elems<CT, TR, ROWS>(); }
template <typename CT, bool TRA, bool TRB, int WT> constexpr size_t gemm_lds_bytes() {
    constexpr size_t stage = 2ul * (gemm_tile_elems<CT, TRA, 64 * WT>() + gemm_tile_elems<CT, TRB, 64 * WT>()) * sizeof(CT);
    constexpr size_t epi = (size_t)(64 * WT) * (64 * WT + 4) * 4;          // fp32 staging tile of the row-wise epilogue
    return stage > epi ? stage : epi;
}

// One staged chunk = CHUNK elements of CT = 16 bytes, kept as a raw 16-byte register quad (never repacked
// element-wise: that would consume the load results immediately and serialise the global loads with the MFMAs).
template <typename CT, typename ST>
VC_DEV vc_u32x4 gemm_pack_chunk(const float (&f)[GemmCfg<CT>::CHUNK]) {
    vc_u32x4 r;
    if constexpr (sizeof(CT) == 2) { r.x = vc_pack_bf16x2(f[0], f[1]); r.y = vc_pack_bf16x2(f[2], f[3]); r.z = vc_pack_bf16x2(f[4], f[5]); r.w = vc_pack_bf16x2(f[6], f[7]); }
    else { r.x = vc_f32_bits(f[0]); r.y = vc_f32_bits(f[1]); r.z = vc_f32_bits(f[2]); r.w = vc_f32_bits(f[3]); }
    return r;
}

template <typename T> VC_DEV float gemm_src_f32(T v) { return vc_cvt<T>::to_f32(v); }
VC_DEV float gemm_src_f32(vc_pk) { return 0.0f; }       // never evaluated (the packed path returns before it); keeps the template well-formed

// Stage one ROWS x BK operand tile.
template <typename CT, typename ST, bool TR, int ROWS>
struct GemmStager {
    static constexpr int BK = GemmCfg<CT>::BK, CH = GemmCfg<CT>::CHUNK, STRIDE = GemmCfg<CT>::STRIDE;
    static constexpr int NCH = ROWS * BK / CH / GEMM_THREADS;     // chunks per thread (4 for 128 rows, 2 for 64; bf16x3: 2 and 1)
    static constexpr int TS = gemm_tstride<CT, ROWS>();
    static constexpr bool X3 = gemm_is_x3<CT>::value;             // registers hold the RAW fp32 chunk (one quad = 4 elements); the hi / lo split happens in store()
    static constexpr int PLANE = gemm_plane_elems<CT, TR, ROWS>();
    static constexpr bool PK = std::is_same<ST, vc_pk>::value;    // pre-split source words: unpacked, not split, in store()
    static_assert(!X3 || sizeof(ST) == 4, "bf16x3 splits fp32 sources");
    static_assert(!PK || X3, "pre-split operands feed the bf16x3 kernel only");
    vc_u32x4 regs[NCH];

    // interior tile + 16-byte-aligned operand: straight-line vector loads (no per-chunk branch, so all loads of a
    // K-tile are in flight together; a divergent bounds test per chunk makes hipcc drain vmcnt after every load)
    VC_DEV void load_fast(const ST* base, long ld, int r0, int k0, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + GEMM_THREADS * i;
            const ST* p;
            if constexpr (!TR) p = base + (long)(r0 + c / (BK / CH)) * ld + (k0 + (c % (BK / CH)) * CH);
            else p = base + (long)(k0 + c / (ROWS / CH)) * ld + (r0 + (c % (ROWS / CH)) * CH);
            if constexpr (X3) {
                regs[i] = *reinterpret_cast<const vc_u32x4*>(p);
            } else if constexpr (sizeof(ST) == sizeof(CT)) {
                regs[i] = *reinterpret_cast<const vc_u32x4*>(p);
            } else {   // fp32 source feeding bf16 MFMA
                const vc_u32x4 lo = reinterpret_cast<const vc_u32x4*>(p)[0], hi = reinterpret_cast<const vc_u32x4*>(p)[1];
                regs[i].x = vc_pack_bf16x2(vc_bits_f32(lo.x), vc_bits_f32(lo.y)); regs[i].y = vc_pack_bf16x2(vc_bits_f32(lo.z), vc_bits_f32(lo.w));
                regs[i].z = vc_pack_bf16x2(vc_bits_f32(hi.x), vc_bits_f32(hi.y)); regs[i].w = vc_pack_bf16x2(vc_bits_f32(hi.z), vc_bits_f32(hi.w));
            }
        }
    }
    // tile that is ragged in the ROW dimension only (whole k-tile, 16-byte-aligned operand, k contiguous in memory): load_fast's vector loads on a
    // row index clamped to the last valid row.  The rows past R then hold a copy of row R - 1: they feed accumulator rows (A) / columns (B) that no
    // epilogue path stores.  (r04: with the element-wise path below, the last 64-row tile of the decoder's 2 976-row Linears at the maximum horizon
    // — 16 of 752 workgroups — set the duration of the whole launch: 45-71 us against 14-22 us at 2 048 rows, profiles/r04_t186_kernel_stats.txt.)
    VC_DEV void load_fast_rows(const ST* base, long ld, int r0, int k0, int R, int tid) {
        static_assert(!TR, "row clamping needs the k-contiguous layout");
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + GEMM_THREADS * i;
            int row = r0 + c / (BK / CH); row = row < R ? row : R - 1;
            const ST* p = base + (long)row * ld + (k0 + (c % (BK / CH)) * CH);
            if constexpr (X3) {
                regs[i] = *reinterpret_cast<const vc_u32x4*>(p);
            } else if constexpr (sizeof(ST) == sizeof(CT)) {
                regs[i] = *reinterpret_cast<const vc_u32x4*>(p);
            } else {   // fp32 source feeding bf16 MFMA
                const vc_u32x4 lo = reinterpret_cast<const vc_u32x4*>(p)[0], hi = reinterpret_cast<const vc_u32x4*>(p)[1];
                regs[i].x = vc_pack_bf16x2(vc_bits_f32(lo.x), vc_bits_f32(lo.y)); regs[i].y = vc_pack_bf16x2(vc_bits_f32(lo.z), vc_bits_f32(lo.w));
                regs[i].z = vc_pack_bf16x2(vc_bits_f32(hi.x), vc_bits_f32(hi.y)); regs[i].w = vc_pack_bf16x2(vc_bits_f32(hi.z), vc_bits_f32(hi.w));
            }
        }
    }
    // edge tiles / unaligned operands: per-element bounds-checked loads, zero fill.
    // R = extent of the row dimension (M or N), Kend = end of this block's k-range
    VC_DEV void load(const ST* base, long ld, int r0, int k0, int R, int Kend, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + GEMM_THREADS * i;
            int nv; const ST* p;
            if constexpr (!TR) {
                int row = c / (BK / CH), kc = (c % (BK / CH)) * CH;
                nv = (r0 + row < R) ? (Kend - (k0 + kc)) : 0;
                p = base + (long)(r0 + row) * ld + (k0 + kc);
            } else {
                int k = c / (ROWS / CH), rc = (c % (ROWS / CH)) * CH;
                nv = (k0 + k < Kend) ? (R - (r0 + rc)) : 0;
                p = base + (long)(k0 + k) * ld + (r0 + rc);
            }
            if constexpr (PK) {       // raw words, zero fill (hi = lo = 0)
                const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
                regs[i].x = 0 < nv ? w[0] : 0u; regs[i].y = 1 < nv ? w[1] : 0u; regs[i].z = 2 < nv ? w[2] : 0u; regs[i].w = 3 < nv ? w[3] : 0u;
                continue;
            }
            float f[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) f[j] = (j < nv) ? gemm_src_f32(p[j]) : 0.0f;
            if constexpr (X3) { regs[i].x = vc_f32_bits(f[0]); regs[i].y = vc_f32_bits(f[1]); regs[i].z = vc_f32_bits(f[2]); regs[i].w = vc_f32_bits(f[3]); }
            else regs[i] = gemm_pack_chunk<CT, ST>(f);
        }
    }
    // "the loaded values are used HERE": without it hipcc hoists the hi / lo split of a bf16x3 tile (pure arithmetic on the load results)
    // above the MFMAs of the previous tile — right behind the loads, whose latency the MFMAs were meant to hide
    VC_DEV void pin() {
#ifndef VC_EMU
#pragma unroll
        for (int i = 0; i < NCH; ++i) asm volatile("" : "+v"(regs[i].x), "+v"(regs[i].y), "+v"(regs[i].z), "+v"(regs[i].w));
#endif
    }
    VC_DEV void store(CT* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + GEMM_THREADS * i;
            if constexpr (X3) {       // split the fp32 quad into its hi and lo bf16 planes (same image in both): two 8-byte LDS stores
                const vc_u32x4 a = regs[i];
                vc_u32x2 hi, lo;
                if constexpr (PK) gemm_unpack4(a, hi, lo); else gemm_split4(a, hi, lo);
                int off;          // 16-byte slots are swizzled, the two 8-byte halves of a slot stay in order
                if constexpr (!TR) { const int row = c / (BK / CH), q = c % (BK / CH); off = row * STRIDE + (((q >> 1) ^ ((row >> 2) & 3)) * 8) + (q & 1) * 4; }
                else if constexpr (gemm_tswz<CT, ROWS>()) { const int k = c / (ROWS / CH); off = k * TS + (((c % (ROWS / CH)) * CH) ^ ((k & 3) << 5)); }
                else off = (c / (ROWS / CH)) * TS + (c % (ROWS / CH)) * CH;
                *reinterpret_cast<vc_u32x2*>(lds + off) = hi;
                *reinterpret_cast<vc_u32x2*>(lds + PLANE + off) = lo;
                continue;
            }
            const uint32_t w[4] = {regs[i].x, regs[i].y, regs[i].z, regs[i].w};
            if constexpr (!TR) {
                int row = c / (BK / CH), kc = (c % (BK / CH)) * CH;
                if constexpr (sizeof(CT) == 2) {
                    *reinterpret_cast<vc_u32x4*>(lds + row * STRIDE + kc) = regs[i];
                } else {
                    uint32_t* d = reinterpret_cast<uint32_t*>(lds + row * STRIDE + kc);
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[j] = w[j];
                }
            } else {
                int k = c / (ROWS / CH), rc = (c % (ROWS / CH)) * CH;
                if constexpr (sizeof(CT) == 2) {
                    *reinterpret_cast<vc_u32x4*>(lds + k * TS + rc) = regs[i];      // natural [k][row] image
                } else {
                    uint32_t* d = reinterpret_cast<uint32_t*>(lds);
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[(rc + j) * STRIDE + k] = w[j];
                }
            }
        }
    }
};

// bf16 MFMA fragment (8 k-values of one row) for k-step ks of the tile; row0 = first row of the wave's 32-row block
template <bool TR, int ROWS, typename CT = vc_bf16>
VC_DEV vc_s16x8 gemm_frag_bf16(const CT* tile, int row0, int ks, int lane) {
    constexpr int GEMM_TSTRIDE = gemm_tstride<CT, ROWS>();
    if constexpr (!TR && gemm_is_x3<CT>::value) {     // unpadded 64-byte rows, chunk index swizzled by bits 2-3 of the row (row0 is a multiple of 32)
        return *reinterpret_cast<const vc_s16x8*>(tile + (row0 + (lane & 31)) * GemmCfg<CT>::STRIDE + (((ks * 2 + (lane >> 5)) ^ ((lane >> 2) & 3)) * 8));
    } else if constexpr (!TR) {
        return *reinterpret_cast<const vc_s16x8*>(tile + (row0 + (lane & 31)) * GemmCfg<CT>::STRIDE + ks * 16 + (lane >> 5) * 8);
    } else {
        const int i = lane & 15;
        const int k = ks * 16 + 8 * (lane >> 5) + (i >> 2);
        int col = row0 + ((lane >> 4) & 1) * 16 + (i & 3) * 4;
        if constexpr (gemm_tswz<CT, ROWS>()) col ^= (k & 3) << 5;          // (k + 4 shares k & 3: the second read is 4 rows further down)
        const CT* p = tile + k * GEMM_TSTRIDE + col;
        const vc_s16x4 lo = vc_ds_read_tr16(p), hi = vc_ds_read_tr16(p + 4 * GEMM_TSTRIDE);
        vc_s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
}

// The tile program.  (bid, nx, ny) = linear tile id and tile-grid extent of THIS problem, bz = k-slice — the plain kernel
// passes its own block coordinates, the grouped kernel the position inside the problem a workgroup was assigned to.
template <typename CT, typename SA, typename SB, typename TO, bool TRA, bool TRB, int WT>
VC_DEV void gemm_tile_program(const GemmParams& p, const int bid, const int nx, const int ny, const int bz) {
    constexpr int BK = GemmCfg<CT>::BK, STRIDE = GemmCfg<CT>::STRIDE;
    constexpr int GEMM_BM = 64 * WT, GEMM_BN = 64 * WT, WS = 32 * WT;     // block tile, per-wave sub-tile
    VC_DYN_SHARED(CT, lds);
    // NB: buffers are addressed as base + integer offset.  Keeping the two tile pointers in an array makes hipcc
    // lose the LDS address space (flat_load/flat_store instead of ds_read_b128/ds_write_b128: ~10x slower).
    constexpr int ATILE = gemm_tile_elems<CT, TRA, GEMM_BM>(), TILE = ATILE + gemm_tile_elems<CT, TRB, GEMM_BN>();

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (private L2s), so give each XCD a contiguous
    // run of tiles, n fastest: the blocks that share one 128-row A panel (and sweep the small B) hit the same L2.
    int tile_m, tile_n;
    {
        const int total = nx * ny;
        const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;     // bijective for any total
        if (p.n_group > 0 && p.n_group < nx) {
            // column groups (r03): when the whole B operand does not fit an XCD's 4 MiB L2 (fp32 / bf16x3 QKV weights: 6.3 MB), a row-major
            // sweep re-fetches B for every tile row (bf16x3 QKV forward: 4.4 GB fetched for 0.22 GB of operands, profiles/r03_x3_pmc.md).
            // Sweep n_group tile columns at a time over ALL tile rows instead: that slice of B stays resident, A is read nx / n_group times.
            const int gsz = p.n_group * ny, g = t / gsz, rem = t - g * gsz;
            const int w = (g + 1) * p.n_group <= nx ? p.n_group : nx - g * p.n_group;
            tile_m = rem / w; tile_n = g * p.n_group + rem - tile_m * w;
        } else {
            tile_m = t / nx; tile_n = t - tile_m * nx;
        }
    }
    const int m0 = tile_m * GEMM_BM, n0 = tile_n * GEMM_BN;
#ifndef VC_EMU
    // De-phase co-resident workgroups: every block of a launch takes the same time, so the two blocks sharing a CU would
    // run their (MFMA-bound) K-loops and their (HBM-bound) epilogues in lockstep forever.  Half of the FIRST wave of blocks
    // starts ~half a tile late; the offset then persists for the rest of the launch because successors start when
    // predecessors retire.  (Pure scheduling hint: no effect on results.)
    if (p.stagger > 0 && bz == 0) {
        const int bid0 = bid;
        if (bid0 < 2 * 256 && ((bid0 >> 3) & 1)) for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    const int kbeg = bz * p.k_per_split;
    const int kend = (kbeg + p.k_per_split < p.K) ? (kbeg + p.k_per_split) : p.K;
    const int nt = (kend - kbeg + BK - 1) / BK;

    vc_f32x16 acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Two-tile-deep register prefetch: tile t+2 is requested from HBM/L2 while tile t is on the matrix cores and tile t+1
    // waits in the other staging set, so every global load has two MFMA phases + a barrier to land (one 512-cycle MFMA
    // phase is shorter than an L2 round trip).  The loop is unrolled by two so both staging sets are statically indexed.
    GemmStager<CT, SA, TRA, GEMM_BM> sa0, sa1;
    GemmStager<CT, SB, TRB, GEMM_BN> sb0, sb1;
    const SA* Ag = (const SA*)p.A;
    const SB* Bg = (const SB*)p.B;
    const bool rowsA = p.vecA && (m0 + GEMM_BM <= p.M), rowsB = p.vecB && (n0 + GEMM_BN <= p.N);   // block-uniform

    // `fast` (compile-time): interior block with vector-aligned operands and whole k-tiles — straight-line 16-byte loads only
    auto fetch = [&](auto fast, GemmStager<CT, SA, TRA, GEMM_BM>& sa, GemmStager<CT, SB, TRB, GEMM_BN>& sb, int t) {
        const int k1 = kbeg + t * BK;
        if constexpr (decltype(fast)::value) {
            sa.load_fast(Ag, p.lda, m0, k1, tid); sb.load_fast(Bg, p.ldb, n0, k1, tid);
        } else {
            const bool kfull = k1 + BK <= kend;
            if (rowsA && kfull) sa.load_fast(Ag, p.lda, m0, k1, tid);
            else if (!TRA && p.vecA && kfull) { if constexpr (!TRA) sa.load_fast_rows(Ag, p.lda, m0, k1, p.M, tid); }     // ragged last row tile: clamped vector loads
            else sa.load(Ag, p.lda, m0, k1, p.M, kend, tid);
            if (rowsB && kfull) sb.load_fast(Bg, p.ldb, n0, k1, tid);
            else if (!TRB && p.vecB && kfull) { if constexpr (!TRB) sb.load_fast_rows(Bg, p.ldb, n0, k1, p.N, tid); }
            else sb.load(Bg, p.ldb, n0, k1, p.N, kend, tid);
        }
    };
    auto compute = [&](int cur) {
        const CT* a_tile = lds + cur * TILE;
        const CT* b_tile = a_tile + ATILE;
        if constexpr (gemm_is_x3<CT>::value) {
            // bf16x3: small terms first (lo*hi, hi*lo), then hi*hi, all into the same fp32 accumulator
            constexpr int AP = gemm_plane_elems<CT, TRA, GEMM_BM>(), BP = gemm_plane_elems<CT, TRB, GEMM_BN>();
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                vc_s16x8 ah[WT], al[WT], bh[WT], bl[WT];
#pragma unroll
                for (int i = 0; i < WT; ++i) {
                    ah[i] = gemm_frag_bf16<TRA, GEMM_BM, CT>(a_tile, wm * WS + i * 32, ks, lane);
                    al[i] = gemm_frag_bf16<TRA, GEMM_BM, CT>(a_tile + AP, wm * WS + i * 32, ks, lane);
                    bh[i] = gemm_frag_bf16<TRB, GEMM_BN, CT>(b_tile, wn * WS + i * 32, ks, lane);
                    bl[i] = gemm_frag_bf16<TRB, GEMM_BN, CT>(b_tile + BP, wn * WS + i * 32, ks, lane);
                }
#pragma unroll
                for (int i = 0; i < WT; ++i)
#pragma unroll
                    for (int j = 0; j < WT; ++j) {
                        acc[i][j] = vc_mfma_32x32x16_bf16(al[i], bh[j], acc[i][j]);
                        acc[i][j] = vc_mfma_32x32x16_bf16(ah[i], bl[j], acc[i][j]);
                        acc[i][j] = vc_mfma_32x32x16_bf16(ah[i], bh[j], acc[i][j]);
                    }
            }
        } else if constexpr (sizeof(CT) == 2) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                vc_s16x8 af[WT], bf[WT];
#pragma unroll
                for (int i = 0; i < WT; ++i) {
                    af[i] = gemm_frag_bf16<TRA, GEMM_BM>(a_tile, wm * WS + i * 32, ks, lane);
                    bf[i] = gemm_frag_bf16<TRB, GEMM_BN>(b_tile, wn * WS + i * 32, ks, lane);
                }
#pragma unroll
                for (int i = 0; i < WT; ++i)
#pragma unroll
                    for (int j = 0; j < WT; ++j) acc[i][j] = vc_mfma_32x32x16_bf16(af[i], bf[j], acc[i][j]);
            }
        } else {
            const CT* a_base = a_tile + (wm * WS + (lane & 31)) * STRIDE;
            const CT* b_base = b_tile + (wn * WS + (lane & 31)) * STRIDE;
#pragma unroll 4
            for (int ks = 0; ks < BK / 2; ++ks) {
                float af[WT], bf[WT];
#pragma unroll
                for (int i = 0; i < WT; ++i) {
                    af[i] = a_base[i * 32 * STRIDE + ks * 2 + (lane >> 5)];
                    bf[i] = b_base[i * 32 * STRIDE + ks * 2 + (lane >> 5)];
                }
#pragma unroll
                for (int i = 0; i < WT; ++i)
#pragma unroll
                    for (int j = 0; j < WT; ++j) acc[i][j] = vc_mfma_32x32x2_f32(af[i], bf[j], acc[i][j]);
            }
        }
    };

    // The k-loop exists twice.  Interior blocks of aligned problems (every block of the hot Linears) take the branch-free copy: when the
    // bounds-checked element loads shared the loop with the vector loads, the values of both paths met in the same registers and hipcc
    // waited for every prefetch right where it was issued (vmcnt(0) before the MFMAs: no overlap at all for the bf16x3 kernel, a
    // one-tile-deep prefetch for the bf16 one) — r03.
    auto k_loop = [&](auto fast) VC_INLINE_LAMBDA {
        if (nt > 0) {
            fetch(fast, sa0, sb0, 0);
            if (nt > 1) fetch(fast, sa1, sb1, 1);
            sa0.store(lds, tid); sb0.store(lds + ATILE, tid);
        }
        vc_sync();
        int t = 0;
        // steady state (tiles t + 2 and t + 3 exist): no conditionals around the prefetches, so the number of loads in flight at every
        // wait is a compile-time constant (a prefetch issued under `if (t + 2 < nt)` made hipcc wait with vmcnt(0) at the next LDS store —
        // the loads it had JUST issued included)
        if constexpr (decltype(fast)::value) {
            for (; t + 3 < nt && !VC_ABL(7); t += 2) {
                fetch(fast, sa0, sb0, t + 2);
                compute(0);
                sa1.store(lds + TILE, tid); sb1.store(lds + TILE + ATILE, tid);
                vc_sync();
                fetch(fast, sa1, sb1, t + 3);
                compute(1);
                sa0.store(lds, tid); sb0.store(lds + ATILE, tid);
                vc_sync();
            }
        }
        // tail (and the whole loop of edge blocks): invariant at the top — LDS buffer 0 holds tile t, register set 1 tile t + 1
        for (; t < nt; t += 2) {
            if (t + 2 < nt && !VC_ABL(1)) fetch(fast, sa0, sb0, t + 2);       // set 0 is free: tile t already sits in LDS buffer 0
            if (!VC_ABL(4)) compute(0);
            if (t + 1 < nt && !VC_ABL(2)) { sa1.store(lds + TILE, tid); sb1.store(lds + TILE + ATILE, tid); }
            vc_sync();
            if (t + 1 >= nt) break;
            if (t + 3 < nt && !VC_ABL(1)) fetch(fast, sa1, sb1, t + 3);
            if (!VC_ABL(4)) compute(1);
            if (t + 2 < nt && !VC_ABL(2)) { sa0.store(lds, tid); sb0.store(lds + ATILE, tid); }
            vc_sync();
        }
    };
    // bf16x3: ONE register set (the raw fp32 tile is 32 registers per operand pair; two sets spill) — tile t + 1 flies during tile t's
    // 24 MFMAs and the other workgroup on the CU covers the rest.  Invariant at the top of an iteration: LDS buffer 0 holds tile t.
    auto k_loop1 = [&](auto fast) VC_INLINE_LAMBDA {
        if (nt > 0) { fetch(fast, sa0, sb0, 0); sa0.store(lds, tid); sb0.store(lds + ATILE, tid); }
        vc_sync();
        int t = 0;
        if constexpr (decltype(fast)::value) {
            for (; t + 2 < nt; t += 2) {      // (the fences pin "loads, then MFMAs, then split + LDS stores": left alone, hipcc sinks the loads below the MFMAs, next to their use)
                fetch(fast, sa0, sb0, t + 1); vc_sched_fence();
                compute(0); vc_sched_fence();
                sa0.pin(); sb0.pin(); sa0.store(lds + TILE, tid); sb0.store(lds + TILE + ATILE, tid);
                vc_sync();
                fetch(fast, sa0, sb0, t + 2); vc_sched_fence();
                compute(1); vc_sched_fence();
                sa0.pin(); sb0.pin(); sa0.store(lds, tid); sb0.store(lds + ATILE, tid);
                vc_sync();
            }
        }
        for (; t < nt; t += 2) {
            if (t + 1 < nt) fetch(fast, sa0, sb0, t + 1);
            compute(0);
            if (t + 1 < nt) { sa0.store(lds + TILE, tid); sb0.store(lds + TILE + ATILE, tid); }
            vc_sync();
            if (t + 1 >= nt) break;
            if (t + 2 < nt) fetch(fast, sa0, sb0, t + 2);
            compute(1);
            if (t + 2 < nt) { sa0.store(lds, tid); sb0.store(lds + ATILE, tid); }
            vc_sync();
        }
    };
    const bool interior = rowsA && rowsB && (kend - kbeg) % BK == 0;
    if constexpr (gemm_is_x3<CT>::value) { if (interior) k_loop1(gemm_true()); else k_loop1(gemm_false()); }
    else { if (interior) k_loop(gemm_true()); else k_loop(gemm_false()); }

#ifdef VCAD_AB
    if (p.debug_skip & 8) { if (acc[0][0][0] == 12345.678f) ((float*)p.C)[0] = 1.f; return; }   // ablation: no epilogue
#endif
    // epilogue: D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (!p.partial && p.vecC && n0 + GEMM_BN <= p.N) {      // (a tile ragged in M only takes this path too, rows past M skipped: r04)
        // Interior block: stage the fp32 tile through LDS (free after the last barrier) and run the epilogue on whole
        // rows — residual / activation-source loads and the final stores are coalesced 16-byte (fp32) or 8-byte (bf16)
        // accesses of 512/256-byte row segments instead of 64 strided scalars per lane.
        constexpr int ES = GEMM_BN + 4;                              // fp32 row stride (pad: conflict-free column writes)
        constexpr int TPR = GEMM_BN / 4, RPP = GEMM_THREADS / TPR;   // threads per row, rows per pass
        static_assert(GEMM_BM * ES * 4 <= gemm_lds_bytes<CT, TRA, TRB, WT>(), "epilogue tile must fit the staging LDS");
        float* et = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
            for (int j = 0; j < WT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    et[(wm * WS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * ES + wn * WS + j * 32 + (lane & 31)] = acc[i][j][r];
        vc_sync();
        const int c4 = (tid % TPR) * 4, n = n0 + c4;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) quad_ld_f32(p.bias + n, b4);
#pragma unroll 4
        for (int pass = 0; pass < GEMM_BM / RPP; ++pass) {
            const int row = pass * RPP + tid / TPR, m = m0 + row;
            if (m >= p.M) break;                                     // ragged last row tile (rows only grow with `pass`)
            float v[4];
            quad_ld_f32(et + row * ES + c4, v);
            gemm_epilogue_quad<TO>(p, m, n, v, b4);
        }
        return;
    }
    if (!p.partial && m0 + GEMM_BM <= p.M && n0 + GEMM_BN <= p.N) {        // interior block, unaligned tensors: batched scalar path
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const int n = n0 + wn * WS + j * 32 + (lane & 31);
            const float bias_n = p.bias ? p.bias[n] : 0.0f;
#pragma unroll
            for (int i = 0; i < WT; ++i) gemm_epilogue_tile<TO>(p, m0 + wm * WS + i * 32 + 4 * (lane >> 5), n, acc[i][j], bias_n);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < WT; ++j) {
        const int n = n0 + wn * WS + j * 32 + (lane & 31);
        const float bias_n = (p.bias && !p.partial && n < p.N) ? p.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M && n < p.N) {
                    if (p.partial) p.partial[((long)bz * p.M + m) * p.N + n] = acc[i][j][r];
                    else gemm_epilogue_store<TO>(p, m, n, acc[i][j][r], bias_n);
                }
            }
    }
}

template <typename CT, typename SA, typename SB, typename TO, bool TRA, bool TRB, int WT>
VC_KERNEL __launch_bounds__(GEMM_THREADS, 2) void gemm_kernel(GemmParams p) {
    // (r03 experiment, not kept: mapping every tile of a k-slice of the split-K weight gradients onto ONE XCD — slice = 8 (h / 8T) + h % 8 for
    // hardware block h — so that the 16 tiles of the ViT MLP wgrad share their operand slabs through one L2: PMC fetch 695 MB per call
    // against 653 MB before, i.e. no reuse gained; the 3.1x over-fetch of that launch stands, profiles/r03_summary.md.)
    gemm_tile_program<CT, SA, SB, TO, TRA, TRB, WT>(p, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, blockIdx.z);
}

// Grouped launch: ONE grid over the tiles of many independent problems of the same signature (the decoder's 56 weight
// gradients, each a ~25 us launch on its own, deferred to the end of the decoder backward).  tile_start[g] = first linear tile of
// problem g (tile_start[n] = total); the problem descriptors live in device memory.
struct GemmGroup { const GemmParams* probs; const int* tile_start; int n; };
template <typename CT, typename SA, typename SB, typename TO, bool TRA, bool TRB, int WT>
VC_KERNEL __launch_bounds__(GEMM_THREADS, 2) void gemm_grouped_kernel(GemmGroup grp) {
    const int t = blockIdx.x;
    int g = 0, hi = grp.n - 1;                                                  // the largest g with tile_start[g] <= t: wave-uniform binary search (r06: the class-token
    while (g < hi) { const int mid = (g + hi + 1) >> 1; if (grp.tile_start[mid] <= t) g = mid; else hi = mid - 1; }   // weight gradients are 512 problems; a linear scan was one dependent scalar load per problem)
    const GemmParams p = grp.probs[g];
    constexpr int BT = 64 * WT;
    const int nx = (p.N + BT - 1) / BT, ny = (p.M + BT - 1) / BT;
    if (t - grp.tile_start[g] >= nx * ny) return;                               // padding up to a multiple of 8 tiles
    gemm_tile_program<CT, SA, SB, TO, TRA, TRB, WT>(p, t - grp.tile_start[g], nx, ny, 0);
}

// split-K reducer: sums the fp32 partial slabs in a fixed order (deterministic) and runs the epilogue.
template <typename TO>
VC_KERNEL __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmParams p, int nsplit) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    long total = (long)p.M * p.N;
    if (idx >= total) return;
    float s = 0.0f;
    for (int z = 0; z < nsplit; ++z) s += p.partial[(long)z * total + idx];
    const int n = (int)(idx % p.N);
    gemm_epilogue_store<TO>(p, (int)(idx / p.N), n, s, p.bias ? p.bias[n] : 0.0f);
}

// same, four consecutive columns per thread (16-byte slab reads, the vector epilogue): used whenever the row-wise epilogue is legal
template <typename TO>
VC_KERNEL __launch_bounds__(256) void gemm_splitk_reduce4_kernel(GemmParams p, int nsplit) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;               // quad index
    const long total = (long)p.M * p.N;
    if (q * 4 >= total) return;
    const long idx = q * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < nsplit; ++z) {
        float v[4]; quad_ld_f32(p.partial + (long)z * total + idx, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += v[k];
    }
    const int n = (int)(idx % p.N), m = (int)(idx / p.N);
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) quad_ld_f32(p.bias + n, b4);
    gemm_epilogue_quad<TO>(p, m, n, s, b4);
}
