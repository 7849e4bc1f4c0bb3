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
// Source types SA/SB may be float while CT is bf16 (converted while staging into LDS), so the fp32
// residual stream / fp32 residual-gradients feed bf16 MFMA without an extra HBM round trip.
//
// Tile: 128x128 per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA 32x32 tiles, 64 fp32
// accumulator registers), BK = 64 (bf16) / 32 (f32), LDS double-buffered, one barrier per K-tile, global
// loads for tile t+1 issued before the MFMAs of tile t (register staging).  LDS rows are padded by one
// 16-byte slot (bf16) / one dword (f32) so the ds_read_b128 / ds_read_b32 fragment reads are conflict-free.
#pragma once
#include "vc_rt.h"

enum { VC_ACT_NONE = 0, VC_ACT_GELU = 1, VC_ACT_RELU = 2, VC_ACT_TANH = 3 };

struct GemmParams {
    const void* A; const void* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    int vecA, vecB;                 // 16-byte vector loads legal for A / B (host-checked alignment)
    // split-K: gridDim.z slices of k_per_split; partials (fp32, [z][M][N]) go to `partial`, epilogue runs in the reducer
    int k_per_split; float* partial;
    // epilogue (applied in this order): v = alpha*acc (+bias[n]) (+rowadd[row(m)][n]); aux[m,n]=v; v=act(v); v*=dact(src[m,n]); v+=residual[m,n]
    float alpha;
    const float* bias;
    const float* rowadd; int rowadd_div; int rowadd_mod; long ld_rowadd;   // row = rowadd_mod ? m % div : m / div
    void* aux; long ldaux;          // type TO
    int act;
    const void* dact_src; long lddact; int dact_kind;                     // type TO
    const float* residual; long ldr;                                      // fp32, may alias C
};

VC_DEV float vc_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
VC_DEV float vc_dgelu(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

template <typename TO>
VC_DEV void gemm_epilogue_store(const GemmParams& p, int m, int n, float acc) {
    float v = p.alpha * acc;
    if (p.bias) v += p.bias[n];
    if (p.rowadd) {
        int r = p.rowadd_mod ? (m % p.rowadd_div) : (m / p.rowadd_div);
        v += p.rowadd[(long)r * p.ld_rowadd + n];
    }
    if (p.aux) vc_st(((TO*)p.aux) + (long)m * p.ldaux + n, v);
    if (p.act == VC_ACT_GELU) v = vc_gelu(v);
    else if (p.act == VC_ACT_RELU) v = fmaxf(v, 0.0f);
    else if (p.act == VC_ACT_TANH) v = tanhf(v);
    if (p.dact_src) {
        float s = vc_ld(((const TO*)p.dact_src) + (long)m * p.lddact + n);
        if (p.dact_kind == VC_ACT_GELU) v *= vc_dgelu(s);
        else if (p.dact_kind == VC_ACT_RELU) v = (s > 0.0f) ? v : 0.0f;
        else if (p.dact_kind == VC_ACT_TANH) v *= (1.0f - s * s);
    }
    if (p.residual) v += p.residual[(long)m * p.ldr + n];
    vc_st(((TO*)p.C) + (long)m * p.ldc + n, v);
}

template <typename CT> struct GemmCfg;
template <> struct GemmCfg<float> {
    static constexpr int BK = 32, CHUNK = 4, STRIDE = 33, KSTEP = 2;
};
template <> struct GemmCfg<vc_bf16> {
    static constexpr int BK = 64, CHUNK = 8, STRIDE = 72, KSTEP = 16;
};
constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_THREADS = 256;

template <typename CT> constexpr size_t gemm_lds_bytes() {
    return 2ul * (GEMM_BM + GEMM_BN) * GemmCfg<CT>::STRIDE * sizeof(CT);
}

// one staged chunk = CHUNK elements of CT = 16 bytes
template <typename CT> struct GemmChunk { CT e[GemmCfg<CT>::CHUNK]; };

template <typename CT, typename ST>
VC_DEV GemmChunk<CT> gemm_load_chunk(const ST* p, int nvalid, int vec_ok) {
    constexpr int CH = GemmCfg<CT>::CHUNK;
    GemmChunk<CT> c;
    if (nvalid >= CH && vec_ok) {
        if constexpr (sizeof(ST) == sizeof(CT)) {
            *reinterpret_cast<vc_u32x4*>(&c) = *reinterpret_cast<const vc_u32x4*>(p);
        } else {   // ST=float, CT=bf16: 8 floats -> 8 bf16
            vc_u32x4 lo = reinterpret_cast<const vc_u32x4*>(p)[0];
            vc_u32x4 hi = reinterpret_cast<const vc_u32x4*>(p)[1];
            float f[8];
            __builtin_memcpy(f, &lo, 16); __builtin_memcpy(f + 4, &hi, 16);
#pragma unroll
            for (int i = 0; i < CH; ++i) c.e[i] = vc_cvt<CT>::from_f32(f[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < CH; ++i) c.e[i] = vc_cvt<CT>::from_f32(i < nvalid ? vc_cvt<ST>::to_f32(p[i]) : 0.0f);
    }
    return c;
}

// Stage one 128 x BK operand tile.  LDS image is always [row][k] (k contiguous, padded stride).
template <typename CT, typename ST, bool TR>
struct GemmStager {
    static constexpr int BK = GemmCfg<CT>::BK, CH = GemmCfg<CT>::CHUNK, STRIDE = GemmCfg<CT>::STRIDE;
    static constexpr int NCH = 128 * BK / CH / GEMM_THREADS;      // chunks per thread (= 4)
    GemmChunk<CT> regs[NCH];

    // R = extent of the row dimension (M or N), Kend = end of this block's k-range
    VC_DEV void load(const ST* base, long ld, int r0, int k0, int R, int Kend, int vec_ok, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + GEMM_THREADS * i;
            if constexpr (!TR) {
                int row = c / (BK / CH), kc = (c % (BK / CH)) * CH;
                int nv = (r0 + row < R) ? (Kend - (k0 + kc)) : 0;
                nv = nv < 0 ? 0 : (nv > CH ? CH : nv);
                const ST* p = base + (long)(r0 + row) * ld + (k0 + kc);
                regs[i] = gemm_load_chunk<CT, ST>(nv > 0 ? p : base, nv, vec_ok);
            } else {
                int k = c / (128 / CH), rc = (c % (128 / CH)) * CH;
                int nv = (k0 + k < Kend) ? (R - (r0 + rc)) : 0;
                nv = nv < 0 ? 0 : (nv > CH ? CH : nv);
                const ST* p = base + (long)(k0 + k) * ld + (r0 + rc);
                regs[i] = gemm_load_chunk<CT, ST>(nv > 0 ? p : base, nv, vec_ok);
            }
        }
    }
    VC_DEV void store(CT* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + GEMM_THREADS * i;
            if constexpr (!TR) {
                int row = c / (BK / CH), kc = (c % (BK / CH)) * CH;
                if constexpr (sizeof(CT) == 2) {
                    *reinterpret_cast<vc_u32x4*>(lds + row * STRIDE + kc) = *reinterpret_cast<const vc_u32x4*>(&regs[i]);
                } else {
#pragma unroll
                    for (int j = 0; j < CH; ++j) lds[row * STRIDE + kc + j] = regs[i].e[j];
                }
            } else {
                int k = c / (128 / CH), rc = (c % (128 / CH)) * CH;
#pragma unroll
                for (int j = 0; j < CH; ++j) lds[(rc + j) * STRIDE + k] = regs[i].e[j];
            }
        }
    }
};

template <typename CT, typename SA, typename SB, typename TO, bool TRA, bool TRB>
VC_KERNEL __launch_bounds__(GEMM_THREADS) void gemm_kernel(GemmParams p) {
    constexpr int BK = GemmCfg<CT>::BK, STRIDE = GemmCfg<CT>::STRIDE;
    VC_DYN_SHARED(CT, lds);
    CT* As[2] = {lds, lds + (GEMM_BM + GEMM_BN) * STRIDE};
    CT* Bs[2] = {As[0] + GEMM_BM * STRIDE, As[1] + GEMM_BM * STRIDE};

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GEMM_BM, n0 = blockIdx.x * GEMM_BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = (kbeg + p.k_per_split < p.K) ? (kbeg + p.k_per_split) : p.K;
    const int nt = (kend - kbeg + BK - 1) / BK;

    vc_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    GemmStager<CT, SA, TRA> sa;
    GemmStager<CT, SB, TRB> sb;
    const SA* Ag = (const SA*)p.A;
    const SB* Bg = (const SB*)p.B;

    if (nt > 0) {
        sa.load(Ag, p.lda, m0, kbeg, p.M, kend, p.vecA, tid);
        sb.load(Bg, p.ldb, n0, kbeg, p.N, kend, p.vecB, tid);
        sa.store(As[0], tid);
        sb.store(Bs[0], tid);
    }
    vc_sync();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            sa.load(Ag, p.lda, m0, kbeg + (t + 1) * BK, p.M, kend, p.vecA, tid);
            sb.load(Bg, p.ldb, n0, kbeg + (t + 1) * BK, p.N, kend, p.vecB, tid);
        }
        const CT* a_base = As[cur] + (wm * 64 + (lane & 31)) * STRIDE;
        const CT* b_base = Bs[cur] + (wn * 64 + (lane & 31)) * STRIDE;
        if constexpr (sizeof(CT) == 2) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                vc_s16x8 af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = *reinterpret_cast<const vc_s16x8*>(a_base + i * 32 * STRIDE + ks * 16 + (lane >> 5) * 8);
                    bf[i] = *reinterpret_cast<const vc_s16x8*>(b_base + i * 32 * STRIDE + ks * 16 + (lane >> 5) * 8);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = vc_mfma_32x32x16_bf16(af[i], bf[j], acc[i][j]);
            }
        } else {
#pragma unroll 4
            for (int ks = 0; ks < BK / 2; ++ks) {
                float af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = a_base[i * 32 * STRIDE + ks * 2 + (lane >> 5)];
                    bf[i] = b_base[i * 32 * STRIDE + ks * 2 + (lane >> 5)];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = vc_mfma_32x32x2_f32(af[i], bf[j], acc[i][j]);
            }
        }
        if (t + 1 < nt) {
            sa.store(As[cur ^ 1], tid);
            sb.store(Bs[cur ^ 1], tid);
        }
        vc_sync();
    }

    // epilogue: D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                int n = n0 + wn * 64 + j * 32 + (lane & 31);
                if (m < p.M && n < p.N) {
                    if (p.partial) p.partial[((long)blockIdx.z * p.M + m) * p.N + n] = acc[i][j][r];
                    else gemm_epilogue_store<TO>(p, m, n, acc[i][j][r]);
                }
            }
}

// split-K reducer: sums the fp32 partial slabs in a fixed order (deterministic) and runs the epilogue.
template <typename TO>
VC_KERNEL __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmParams p, int nsplit) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    long total = (long)p.M * p.N;
    if (idx >= total) return;
    float s = 0.0f;
    for (int z = 0; z < nsplit; ++z) s += p.partial[(long)z * total + idx];
    gemm_epilogue_store<TO>(p, (int)(idx / p.N), (int)(idx % p.N), s);
}
