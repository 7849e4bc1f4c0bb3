// norm.h — LayerNorm family (one wave64 per row, wave-shuffle reductions, 16-byte loads), the fused
// patchify+LayerNorm(1024) that feeds the patch-embed GEMM, and the LN(512)+cls+pos "embed finalize".
// Statistics are two-pass in registers (mean, then centred variance) like ATen's, eps inside the sqrt.
#pragma once
#include "vc_rt.h"

// ---- row access helpers: a wave owns one row of C = 64*VPL elements; lane l owns VPL/4 groups of 4
// consecutive elements at columns g*256 + l*4 + (0..3)  (16-byte loads for fp32, 8-byte for bf16).
// (explicit 16-byte / 8-byte vector accesses: rows are 16-byte aligned everywhere on the hot path)
template <typename T> VC_DEV void quad_load(const T* p, float* v);
template <> VC_DEV void quad_load<float>(const float* p, float* v) {
    const vc_u32x4 q = *reinterpret_cast<const vc_u32x4*>(p);
    v[0] = vc_bits_f32(q.x); v[1] = vc_bits_f32(q.y); v[2] = vc_bits_f32(q.z); v[3] = vc_bits_f32(q.w);
}
template <> VC_DEV void quad_load<vc_bf16>(const vc_bf16* p, float* v) {
    const vc_u32x2 q = *reinterpret_cast<const vc_u32x2*>(p);
    v[0] = vc_lo16_f32(q.x); v[1] = vc_hi16_f32(q.x); v[2] = vc_lo16_f32(q.y); v[3] = vc_hi16_f32(q.y);
}
template <> VC_DEV void quad_load<vc_pk>(const vc_pk* p, float* v) {
    const vc_u32x4 q = *reinterpret_cast<const vc_u32x4*>(p);
    v[0] = vc_bits_f32(q.x & 0xFFFF0000u) + vc_bits_f32(q.x << 16); v[1] = vc_bits_f32(q.y & 0xFFFF0000u) + vc_bits_f32(q.y << 16);
    v[2] = vc_bits_f32(q.z & 0xFFFF0000u) + vc_bits_f32(q.z << 16); v[3] = vc_bits_f32(q.w & 0xFFFF0000u) + vc_bits_f32(q.w << 16);
}
template <typename T> VC_DEV void quad_store(T* p, const float* v);
template <> VC_DEV void quad_store<float>(float* p, const float* v) {
    vc_u32x4 q; q.x = vc_f32_bits(v[0]); q.y = vc_f32_bits(v[1]); q.z = vc_f32_bits(v[2]); q.w = vc_f32_bits(v[3]);
    *reinterpret_cast<vc_u32x4*>(p) = q;
}
template <> VC_DEV void quad_store<vc_bf16>(vc_bf16* p, const float* v) {
    vc_u32x2 q; q.x = vc_pack_bf16x2(v[0], v[1]); q.y = vc_pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<vc_u32x2*>(p) = q;
}
template <> VC_DEV void quad_store<vc_pk>(vc_pk* p, const float* v) {           // pre-split bf16x3 operand words (gemm.h): the GEMM that consumes the tensor unpacks
    vc_u32x4 q; q.x = vc_pk_pack(v[0]); q.y = vc_pk_pack(v[1]); q.z = vc_pk_pack(v[2]); q.w = vc_pk_pack(v[3]);
    *reinterpret_cast<vc_u32x4*>(p) = q;
}
template <typename T, int VPL>
VC_DEV void row_load(const T* p, float (&v)[VPL], int lane) {
#pragma unroll
    for (int g = 0; g < VPL / 4; ++g) quad_load<T>(p + g * 256 + lane * 4, &v[g * 4]);
}
template <typename T, int VPL>
VC_DEV void row_store(T* p, const float (&v)[VPL], int lane) {
#pragma unroll
    for (int g = 0; g < VPL / 4; ++g) quad_store<T>(p + g * 256 + lane * 4, &v[g * 4]);
}
template <int VPL>
VC_DEV void row_stats(const float (&v)[VPL], float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) s += v[i];
    mean = vc_wave_sum(s) * (1.0f / (64 * VPL));
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { float d = v[i] - mean; q += d * d; }
    rstd = rsqrtf(vc_wave_sum(q) * (1.0f / (64 * VPL)) + eps);
}

// patch p of frame n: vector index e = p1*32 + p2 (c = 1) <- frame[n][ph*32+p1][pw*32+p2]
// ('b c (h p1) (w p2) -> b (h w) (p1 p2 c)', reference model/trajectory_model.py:54-65 via vit-pytorch)
// image n = b*T + t lives at frames + b*bstride + t*img*img (so the [:, :-1] view of the loader batch needs no copy)
// u8 = 1: the frames are the loader's uint8 grayscale pixels and the torchvision ToTensor + Normalize(0.5, 0.5) of the reference
// (main.py:103-108; data_loader.py:441-447) happens here, in the same fp32 operations and order ((u / 255 - 0.5) / 0.5, correctly
// rounded division), so the patch vectors are bit-identical to the fp32 path while PCIe and HBM carry 1 byte per pixel instead of 4.
VC_DEV void quad_load_u8norm(const uint8_t* p, float* v) {
    const uint32_t q = *reinterpret_cast<const uint32_t*>(p);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ((float)((q >> (8 * k)) & 0xFFu) / 255.0f - 0.5f) / 0.5f;
}
// u8 = 2: the frames are the STORED pixels of the dataset — uint8 RGB, interleaved [H][W][3] (pkl `frames uint8 [N,224,224,3]`, reference
// data_loader/sequence_retriver.py:25-33) — and PIL's `Image.convert('L')` (torchvision Grayscale, reference data_loader.py:441-447; the integer
// ITU-R 601-2 form L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16, bit-exact against PIL: tests/test_data_cpu.py) happens here too, followed by
// the same normalisation — the host does no per-pixel work at all.  4 pixels = 12 bytes = three aligned dwords.
VC_DEV void quad_load_rgb8norm(const uint8_t* p, float* v) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
    const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];                         // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
    const uint32_t r[4] = {w0 & 0xFFu, w0 >> 24, (w1 >> 16) & 0xFFu, (w2 >> 8) & 0xFFu};
    const uint32_t g[4] = {(w0 >> 8) & 0xFFu, w1 & 0xFFu, w1 >> 24, (w2 >> 16) & 0xFFu};
    const uint32_t b[4] = {(w0 >> 16) & 0xFFu, (w1 >> 8) & 0xFFu, w2 & 0xFFu, w2 >> 24};
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ((float)((r[k] * 19595u + g[k] * 38470u + b[k] * 7471u + 0x8000u) >> 16) / 255.0f - 0.5f) / 0.5f;
}
template <int VPL>
VC_DEV void patch_load(const void* frames_, int u8, long row, float (&v)[VPL], int lane, int img, int patch, int T, long bstride) {
    const int g_ = img / patch;
    const long n = row / (g_ * g_);
    const int pp = (int)(row % (g_ * g_)), ph = pp / g_, pw = pp % g_;
    const long off = (n / T) * bstride + (n % T) * (long)img * img + (long)(ph * patch) * img + pw * patch;
#pragma unroll
    for (int g = 0; g < VPL / 4; ++g) {
        int e = g * 256 + lane * 4;
        int p1 = e / patch, p2 = e % patch;
        if (u8 == 2) quad_load_rgb8norm((const uint8_t*)frames_ + 3 * (off + (long)p1 * img + p2), &v[g * 4]);      // (bstride / offsets in pixels)
        else if (u8) quad_load_u8norm((const uint8_t*)frames_ + off + (long)p1 * img + p2, &v[g * 4]);
        else quad_load<float>((const float*)frames_ + off + (long)p1 * img + p2, &v[g * 4]);
    }
}

struct LnFwdParams {
    const void* x; long ldx;          // input rows (type TX), or frames when PATCH
    const float* gamma; const float* beta;
    float* y32; long ldy32;           // optional fp32 output
    void* yt; long ldyt;              // optional T output
    float* stats;                     // optional [rows][2] = mean, rstd
    long rows; float eps;
    int img, patch; int u8;           // PATCH mode (u8: uint8 pixels, normalised on load)
    // EMBED mode: input row r = n*P + p  ->  output row n*(P+1) + p + 1, plus pos[p+1]; extra rows write cls+pos[0]
    const float* pos; const float* cls; int P;
    vc_drop drop;                     // EMBED mode: emb_dropout on the finished token rows (idx = out_row * C + col)
    // r05 (plain mode): the residual add of the block in front of this LayerNorm.  `add` (type TY, [rows][ldadd]) is the branch output the preceding
    // Linear wrote (bias and dropout applied, 16-bit); x + add is the new fp32 residual stream, written to sum32 and normalised.  The add used to sit
    // in that Linear's epilogue, where the persistent GEMM's eight lock-stepped waves pay for the fp32 side stream with idle matrix cores
    // (0.17 busy); here it rides on a pass that is HBM-bound anyway.
    const void* add; long ldadd; float* sum32; long ldsum;
    vc_drop add_drop;                 // dropout of the branch (element index row * C + col, the index the Linear's own epilogue would use): x + mask * add
};

// MODE 0 plain, 1 PATCH (x = frames), 2 EMBED (see above; grid covers N*(P+1) output rows)
template <typename TX, typename TY, int VPL, int MODE>
VC_KERNEL __launch_bounds__(256) void ln_fwd_kernel(LnFwdParams p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    constexpr int C = 64 * VPL;
    if (row >= p.rows) return;
    float v[VPL];
    long in_row = row, out_row = row;
    if constexpr (MODE == 2) {
        const long n = row / (p.P + 1); const int t = (int)(row % (p.P + 1));
        if (t == 0) {                                   // cls token + pos[0]
            float c[VPL], q[VPL];
            row_load<float, VPL>(p.cls, c, lane); row_load<float, VPL>(p.pos, q, lane);
#pragma unroll
            for (int i = 0; i < VPL; ++i) c[i] += q[i];
            if (p.drop.key) {
#pragma unroll
                for (int q4 = 0; q4 < VPL / 4; ++q4) { float dm[4]; vc_drop_mul4(p.drop, (uint32_t)(row * C + q4 * 256 + lane * 4), dm); for (int k = 0; k < 4; ++k) c[q4 * 4 + k] *= dm[k]; }
            }
            row_store<float, VPL>(p.y32 + row * p.ldy32, c, lane);
            return;
        }
        in_row = n * p.P + (t - 1);
    }
    if constexpr (MODE == 1) patch_load<VPL>(p.x, p.u8, row, v, lane, p.img, p.patch, p.P, p.ldx);   // P = T, ldx = batch stride
    else row_load<TX, VPL>((const TX*)p.x + in_row * p.ldx, v, lane);
    if constexpr (MODE == 0) if (p.add) {
        float a[VPL];
        row_load<TY, VPL>((const TY*)p.add + row * p.ldadd, a, lane);
        if (p.add_drop.key) {
#pragma unroll
            for (int q4 = 0; q4 < VPL / 4; ++q4) { float dm[4]; vc_drop_mul4(p.add_drop, (uint32_t)(row * C + q4 * 256 + lane * 4), dm); for (int k = 0; k < 4; ++k) a[q4 * 4 + k] *= dm[k]; }
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[i] += a[i];
        if (p.sum32) row_store<float, VPL>(p.sum32 + row * p.ldsum, v, lane);
    }
    float mean, rstd;
    row_stats<VPL>(v, p.eps, mean, rstd);
    if (p.gamma) {
        float g[VPL], b[VPL];
        row_load<float, VPL>(p.gamma, g, lane); row_load<float, VPL>(p.beta, b, lane);
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[i] = (v[i] - mean) * rstd * g[i] + b[i];
    } else {                                            // no affine: the caller folded gamma / beta into the Linear behind this norm (pe_fold_kernel)
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[i] = (v[i] - mean) * rstd;
    }
    if constexpr (MODE == 2) {
        float q[VPL];
        row_load<float, VPL>(p.pos + (long)(row % (p.P + 1)) * C, q, lane);
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[i] += q[i];
        if (p.drop.key) {
#pragma unroll
            for (int q4 = 0; q4 < VPL / 4; ++q4) { float dm[4]; vc_drop_mul4(p.drop, (uint32_t)(row * C + q4 * 256 + lane * 4), dm); for (int k = 0; k < 4; ++k) v[q4 * 4 + k] *= dm[k]; }
        }
    }
    if (p.y32) row_store<float, VPL>(p.y32 + out_row * p.ldy32, v, lane);
    if (p.yt) row_store<TY, VPL>((TY*)p.yt + out_row * p.ldyt, v, lane);
    if (p.stats && lane == 0) { p.stats[in_row * 2] = mean; p.stats[in_row * 2 + 1] = rstd; }
}

struct LnBwdParams {
    const void* dy; long lddy;        // type TD
    const void* x; long ldx;          // type TX (fp32 residual stream / pre-LN sum), or frames when PATCH
    const float* stats;               // [rows][2]
    const float* gamma;
    const float* add_in; long ldadd;  // optional fp32 tensor added to dx (the residual-path gradient)
    int add_period;                   // r06: > 1 = add_in is defined on rows that are multiples of it only and ZERO elsewhere (the class-token-only last ViT layer: the stream gradient
                                      // exists on class rows; r05 cleared the other 49 of 50 rows with a 210 MB memset that this pass then read back)
    float* dx32; long lddx32;         // optional fp32 dx output (may alias add_in)
    void* dxt; long lddxt;            // optional T dx output; with drop.key != 0 it is dx * dropout-mask(row * C + c): the masked
    vc_drop drop;                     //   gradient the next (dropped) Linear's wgrad / dgrad consume, fused here instead of a separate pass
    float* partial;                   // [gridDim.x][2][C] dgamma / dbeta partial sums (fixed order => deterministic)
    long rows;
    int img, patch; int P; int u8;    // PATCH / EMBED mapping (EMBED: dy row = n*(P+1)+p+1 for x row n*P+p); u8 as in LnFwdParams
    int dsum;                         // also reduce the emitted gradient (dxt if written, else dx32) over rows: third partial row = the bias
                                      // gradient of the Linear that consumes it (saves that Linear's own column-sum pass over the tensor)
    vc_drop drop32;                   // r06: dropout mask (row * C + c) applied to the dx32 OUTPUT only — the ViT's emb_dropout rides on the first layer's norm backward
                                      // instead of a pass of its own over the fp32 gradient stream (not combined with dxt / dsum)
};

template <typename TD, typename TX, typename TY, int VPL, int MODE>
VC_KERNEL __launch_bounds__(256) void ln_bwd_kernel(LnBwdParams p) {
    constexpr int C = 64 * VPL;
    VC_SHARED float red[4][C];             // one partial row at a time (three rows at once cost the kernel a quarter of its occupancy)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[VPL], db[VPL], ds[VPL], g[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { dg[i] = 0.f; db[i] = 0.f; ds[i] = 0.f; }
    row_load<float, VPL>(p.gamma, g, lane);
    for (long row = (long)blockIdx.x * 4 + wave; row < p.rows; row += (long)gridDim.x * 4) {
        float x[VPL], dy[VPL];
        long dy_row = row;
        if constexpr (MODE == 2) dy_row = (row / p.P) * (p.P + 1) + (row % p.P) + 1;
        if constexpr (MODE == 1) patch_load<VPL>(p.x, p.u8, row, x, lane, p.img, p.patch, p.P, p.ldx);
        else row_load<TX, VPL>((const TX*)p.x + row * p.ldx, x, lane);
        row_load<TD, VPL>((const TD*)p.dy + dy_row * p.lddy, dy, lane);
        const float mean = p.stats[row * 2], rstd = p.stats[row * 2 + 1];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            x[i] = (x[i] - mean) * rstd;               // xhat
            dg[i] += dy[i] * x[i]; db[i] += dy[i];
            dy[i] *= g[i];
            c1 += dy[i]; c2 += dy[i] * x[i];
        }
        if (p.dx32 || p.dxt) {
            c1 = vc_wave_sum(c1) * (1.0f / C); c2 = vc_wave_sum(c2) * (1.0f / C);
            float dx[VPL];
#pragma unroll
            for (int i = 0; i < VPL; ++i) dx[i] = rstd * (dy[i] - c1 - x[i] * c2);
            if (p.add_in && (p.add_period <= 1 || row % p.add_period == 0)) {
                float a[VPL];
                row_load<float, VPL>(p.add_in + row * p.ldadd, a, lane);
#pragma unroll
                for (int i = 0; i < VPL; ++i) dx[i] += a[i];
            }
            if (p.dx32) {
                if (p.drop32.key) {
#pragma unroll
                    for (int q4 = 0; q4 < VPL / 4; ++q4) { float dm[4]; vc_drop_mul4(p.drop32, (uint32_t)(row * C + q4 * 256 + lane * 4), dm); for (int k = 0; k < 4; ++k) dx[q4 * 4 + k] *= dm[k]; }
                }
                row_store<float, VPL>(p.dx32 + row * p.lddx32, dx, lane);
            }
            if (p.dxt) {
                if (p.drop.key) {
#pragma unroll
                    for (int q4 = 0; q4 < VPL / 4; ++q4) { float dm[4]; vc_drop_mul4(p.drop, (uint32_t)(row * C + q4 * 256 + lane * 4), dm); for (int k = 0; k < 4; ++k) dx[q4 * 4 + k] *= dm[k]; }
                }
                row_store<TY, VPL>((TY*)p.dxt + row * p.lddxt, dx, lane);
            }
            if (p.dsum) {
#pragma unroll
                for (int i = 0; i < VPL; ++i) ds[i] += dx[i];                      // (fp32 values, before the store's rounding)
            }
        }
    }
    if (p.partial) {
        const int NR = p.dsum ? 3 : 2;
        auto flush = [&](const float (&v)[VPL], int part) {
            row_store<float, VPL>(&red[wave][0], v, lane);
            vc_sync();
            for (int i = threadIdx.x; i < C; i += 256)
                p.partial[((long)blockIdx.x * NR + part) * C + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
            vc_sync();
        };
        flush(dg, 0); flush(db, 1);
        if (p.dsum) flush(ds, 2);
    }
}

// ---- column sums: out[c] (=|+=) sum_r x[r*ld + c].  Deterministic 512-way tree: a block is 64 columns x 4 row quarters of a 512-row
// chunk (each thread walks <= 128 rows with 8 loads in flight; the quarters are added 0..3 through LDS), partial rows are reduced by the
// next pass: 104 000 token rows -> 204 -> 1 in two launches, the <= 512 partial rows of a LayerNorm backward in one.  (r01's 128-way
// tree of 256-column blocks needed three and two: ~210 launches of ~6.5 us per train step.)
constexpr int COLSUM_ROWS = 512;
struct ColsumParams {
    const void* x; long ld; long rows; int cols; long batch_stride_x;
    float* out; long ld_out_rows; long batch_stride_out; int accumulate;    // out row g = block-row index (partials) or 0 (final)
    int rows_per_block;
    int seg; float *out1, *out2;        // seg > 0 (single-level, unbatched): columns [0, seg) go to out, [seg, 2 seg) to out1, [2 seg, 3 seg) to out2 — the
                                        // LayerNorm backward's dgamma / dbeta / bias-gradient partial rows in ONE launch instead of three
};
template <typename TX>
VC_KERNEL __launch_bounds__(256) void colsum_pass_kernel(ColsumParams p) {
    VC_SHARED float red[4][64];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool live = c < p.cols;
    const TX* x = (const TX*)p.x + (long)blockIdx.z * p.batch_stride_x;
    const long b0 = (long)blockIdx.y * p.rows_per_block;
    long b1 = b0 + p.rows_per_block; if (b1 > p.rows) b1 = p.rows;
    const long per = ((b1 - b0) + 3) / 4;
    long r = b0 + q * per, r1 = r + per; if (r1 > b1) r1 = b1;
    float s = 0.f;
    if (live) {
        for (; r + 8 <= r1; r += 8) {             // 8 independent loads in flight per thread (a plain loop waits for each)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = vc_ld(x + (r + u) * p.ld + c);
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; r < r1; ++r) s += vc_ld(x + r * p.ld + c);
    }
    red[q][cl] = s;
    vc_sync();
    if (q == 0 && live) {
        const float t = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        float* o = p.out + (long)blockIdx.z * p.batch_stride_out + (long)blockIdx.y * p.ld_out_rows + c;
        if (p.seg > 0) { const int sg = c / p.seg; o = (sg == 0 ? p.out : (sg == 1 ? p.out1 : p.out2)) + (c - sg * p.seg); }
        *o = p.accumulate ? (*o + t) : t;
    }
}

// ---- elementwise: dpre = d * (1 - y^2)   (tanh backward; y fp32)
template <typename TY>
VC_KERNEL __launch_bounds__(256) void dtanh_kernel(const float* d, const float* y, float* out32, TY* outt, long n) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float yy = y[i];
    float v = d[i] * (1.0f - yy * yy);
    if (out32) out32[i] = v;
    if (outt) vc_st(outt + i, v);
}

// ---- out[r][c] = in[r][c] * dropmask(r * cols + c): the masked gradient that feeds a dropped GEMM output's wgrad / dgrad
template <typename TY>
VC_KERNEL __launch_bounds__(256) void dropout_mul_kernel(const float* in, long ld_in, TY* out, long ld_out, long rows, int cols, vc_drop d) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= rows * cols) return;
    const long r = i / cols; const int c = (int)(i % cols);
    float v[4];
    quad_load<float>(in + r * ld_in + c, v);
    { float dm[4]; vc_drop_mul4(d, (uint32_t)i, dm);                              // i is a multiple of 4
      for (int k = 0; k < 4; ++k) v[k] *= dm[k]; }
    quad_store<TY>(out + r * ld_out + c, v);
}

// zero the first `width_bytes` (a multiple of 16) of each of `rows` rows spaced `ld_bytes` apart (16-byte stores; a strided memset)
VC_KERNEL __launch_bounds__(256) void zero_cols_kernel(char* p, long ld_bytes, long rows, int width_bytes) {
    const int per_row = width_bytes / 16;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * per_row) return;
    vc_u32x4 z; z.x = z.y = z.z = z.w = 0u;
    *reinterpret_cast<vc_u32x4*>(p + (i / per_row) * ld_bytes + (i % per_row) * 16) = z;
}

// ---- act = tanh(a W^T + b + ts[t])  with K = act_dim (7): too skinny for MFMA, one thread per output
// (reference model/autoregressive_transformer.py:112,176-178)
template <typename TY>
VC_KERNEL __launch_bounds__(256) void embed_action_kernel(const float* a, const float* W, const float* b, const float* ts,
                                                          float* y32, TY* yt, long M, int H, int K, int T) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * H) return;
    long m = i / H; int n = (int)(i % H);
    float s = b[n] + (ts ? ts[(m % T) * (long)H + n] : 0.0f);
    for (int k = 0; k < K; ++k) s += a[m * K + k] * W[n * K + k];
    s = tanhf(s);
    y32[i] = s;
    if (yt) vc_st(yt + i, s);
}

// ---- mem[m][n] = tanh(src[m / T][n])   (memory = tanh(CAD embedding) repeated over time, reference :163,175 when no projection)
template <typename TS>
VC_KERNEL __launch_bounds__(256) void bcast_tanh_kernel(const TS* src, float* out, long M, int H, int T) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * H) return;
    out[i] = tanhf(vc_ld(src + (i / H / T) * (long)H + (i % H)));
}
VC_KERNEL __launch_bounds__(256) void add_inplace_kernel(float* a, const float* b, long n) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] += b[i];
}

// y = alpha * x (y may be x): the gradient scale of the fp16 build (engine.hip: grad_scale) — 16 bytes per thread, grid-stride
VC_KERNEL __launch_bounds__(256) void scale_kernel(const float* x, float* y, long n, float alpha) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float v[4]; quad_load<float>(x + 4 * i, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= alpha;
        quad_store<float>(y + 4 * i, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[(n4 << 2) + threadIdx.x] = alpha * x[(n4 << 2) + threadIdx.x];
}
// the same for a source or destination that is only 4-byte aligned (a caller's dlogits may be an offset view): one element per thread
VC_KERNEL __launch_bounds__(256) void scale_unaligned_kernel(const float* x, float* y, long n, float alpha) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = alpha * x[i];
}

// ---- gradient wire format of the data-parallel exchange (engine.hip: vcad_wire_*): a bucket of the flat fp32 gradient buffer travels over xGMI in
// the library's 16-bit storage format (bf16 here, IEEE half in the -DVC_H16 build): half the bytes of the reference's fp32 DDP buckets.
// bf16 has fp32's exponent range: no scale.  fp16 does not: the bucket is multiplied by a power of two derived from its (all-reduced) maximum
// magnitude so that the SUM over `world` ranks stays below 32 768; the scale is recomputed from the same device scalar on the way back.
VC_DEV float wire_scale(const float* amax, int world) {
    if (!amax) return 1.0f;
    const float a = amax[0] * (float)world;
    if (!(a > 0.0f) || !(a <= 3.0e38f)) return 1.0f;              // all-zero bucket, or a non-finite gradient (which must stay non-finite on the wire)
    float e = floorf(log2f(32768.0f / a));
    e = e < -60.0f ? -60.0f : (e > 60.0f ? 60.0f : e);
    return exp2f(e);
}
VC_KERNEL __launch_bounds__(256) void wire_amax_stage1_kernel(const float* g, long n, float* partial) {
    VC_SHARED float red[256];
    float m = 0.f;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 4 <= n) { float v[4]; quad_load<float>(g + i, v); for (int j = 0; j < 4; ++j) { const float a = fabsf(v[j]); m = a > m || a != a ? a : m; } }
        else for (int j = 0; j < 4; ++j) if (i + j < n) { const float a = fabsf(g[i + j]); m = a > m || a != a ? a : m; }
    }
    red[threadIdx.x] = m;
    vc_sync();
    for (int k = 128; k >= 1; k >>= 1) { if ((int)threadIdx.x < k) { const float a = red[threadIdx.x + k], b = red[threadIdx.x]; red[threadIdx.x] = (a > b || a != a) ? a : b; } vc_sync(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
VC_KERNEL __launch_bounds__(256) void wire_amax_stage2_kernel(const float* partial, int nblk, float* out) {
    VC_SHARED float red[256];
    float m = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) { const float a = partial[i]; m = (a > m || a != a) ? a : m; }
    red[threadIdx.x] = m;
    vc_sync();
    for (int k = 128; k >= 1; k >>= 1) { if ((int)threadIdx.x < k) { const float a = red[threadIdx.x + k], b = red[threadIdx.x]; red[threadIdx.x] = (a > b || a != a) ? a : b; } vc_sync(); }
    if (threadIdx.x == 0) out[0] = red[0] != red[0] ? 3.4e38f : red[0];          // (NaN does not survive an all-reduce(MAX) on every backend: send "huge" instead)
}
VC_KERNEL __launch_bounds__(256) void wire_pack_kernel(const float* g, vc_bf16* w, long n, const float* amax, int world) {
    const float sc = wire_scale(amax, world);
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float v[4]; quad_load<float>(g + 4 * i, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= sc;
        quad_store<vc_bf16>(w + 4 * i, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) vc_st(w + (n4 << 2) + threadIdx.x, sc * g[(n4 << 2) + threadIdx.x]);
}
VC_KERNEL __launch_bounds__(256) void wire_unpack_kernel(const vc_bf16* w, float* g, long n, const float* amax, int world) {
    const float inv = 1.0f / wire_scale(amax, world);
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float v[4]; quad_load<vc_bf16>(w + 4 * i, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= inv;
        quad_store<float>(g + 4 * i, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) g[(n4 << 2) + threadIdx.x] = inv * vc_ld(w + (n4 << 2) + threadIdx.x);
}

// ---- fp32 -> T cast (weight shadows) and generic fill
template <typename TY>
VC_KERNEL __launch_bounds__(256) void cast_kernel(const float* x, TY* y, long n) {
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    for (int j = 0; j < 4; ++j) if (i + j < n) vc_st(y + i + j, x[i + j]);
}

// ---- g = dropout(act(z)) and dz = (dz * dropmask) * act'(z), bf16, 8 elements (16 bytes) per thread: the MLP activation of the ViT as its own
// pass (r02).  Fused into the register-staged GEMM's epilogue these cost 200 / 235 us per call (that kernel overlaps 54 GFLOP of MFMA, ~60 us
// of erf / exp / hash VALU work and 320 MB of traffic poorly); the plain GEMM on the persistent kernel plus this pass is 68 + ~60 us.
// Element index = row * cols + col of the compact tensor = the index the fused epilogue hashes, so the masks are the same bits.
VC_DEV void vc_unpack8(const vc_u32x4& q, float (&v)[8]) {
    const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = vc_lo16_f32(u[k]); v[2 * k + 1] = vc_hi16_f32(u[k]); }
}
VC_DEV vc_u32x4 vc_pack8(const float (&v)[8]) {
    vc_u32x4 q; q.x = vc_pack_bf16x2(v[0], v[1]); q.y = vc_pack_bf16x2(v[2], v[3]); q.z = vc_pack_bf16x2(v[4], v[5]); q.w = vc_pack_bf16x2(v[6], v[7]);
    return q;
}
VC_KERNEL __launch_bounds__(256) void act_fwd_bf16_kernel(const vc_bf16* z, vc_bf16* g, long n8, int act, vc_drop d) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float v[8];
    vc_unpack8(*reinterpret_cast<const vc_u32x4*>(z + i * 8), v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = vc_apply_act(v[k], act);
    if (d.key) {
        float m0[4], m1[4];
        vc_drop_mul4(d, (uint32_t)(i * 8), m0); vc_drop_mul4(d, (uint32_t)(i * 8 + 4), m1);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] *= m0[k]; v[4 + k] *= m1[k]; }
    }
    *reinterpret_cast<vc_u32x4*>(g + i * 8) = vc_pack8(v);
}
VC_KERNEL __launch_bounds__(256) void dact_bwd_bf16_kernel(vc_bf16* dz, const vc_bf16* z, long n8, int kind, vc_drop d) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float v[8], s[8];
    vc_unpack8(*reinterpret_cast<const vc_u32x4*>(dz + i * 8), v);
    vc_unpack8(*reinterpret_cast<const vc_u32x4*>(z + i * 8), s);
    if (d.key) {
        float m0[4], m1[4];
        vc_drop_mul4(d, (uint32_t)(i * 8), m0); vc_drop_mul4(d, (uint32_t)(i * 8 + 4), m1);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] *= m0[k]; v[4 + k] *= m1[k]; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = vc_apply_dact(v[k], s[k], kind);
    *reinterpret_cast<vc_u32x4*>(dz + i * 8) = vc_pack8(v);
}

// The same pass over whole rows, also reducing its output over rows (the bias gradient of the Linear whose pre-activation this is): thread =
// one 8-column chunk, 256 / (cols / 8) rows per block step, grid-stride over rows; partial[block][cols] is summed by a column-sum pass over
// <= 2048 rows instead of one over the whole tensor.  Needs 256 % (cols / 8) == 0.
VC_KERNEL __launch_bounds__(256) void dact_bwd_bf16_rows_kernel(vc_bf16* dz, const vc_bf16* z, long rows, int cols, int kind, vc_drop d, float* partial) {
    VC_SHARED float red[256 * 8];
    const int c8n = cols / 8, rpb = 256 / c8n, r_in = threadIdx.x / c8n, c8 = threadIdx.x % c8n;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0f;
    for (long row = (long)blockIdx.x * rpb + r_in; row < rows; row += (long)gridDim.x * rpb) {
        const long i = row * c8n + c8;
        float v[8], s[8];
        vc_unpack8(*reinterpret_cast<const vc_u32x4*>(dz + i * 8), v);
        vc_unpack8(*reinterpret_cast<const vc_u32x4*>(z + i * 8), s);
        if (d.key) {
            float m0[4], m1[4];
            vc_drop_mul4(d, (uint32_t)(i * 8), m0); vc_drop_mul4(d, (uint32_t)(i * 8 + 4), m1);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] *= m0[k]; v[4 + k] *= m1[k]; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = vc_apply_dact(v[k], s[k], kind); acc[k] += v[k]; }
        *reinterpret_cast<vc_u32x4*>(dz + i * 8) = vc_pack8(v);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[r_in * cols + c8 * 8 + k] = acc[k];
    vc_sync();
    for (int c = threadIdx.x; c < cols; c += 256) {
        float t = 0.0f;
        for (int r = 0; r < rpb; ++r) t += red[r * cols + c];
        partial[(long)blockIdx.x * cols + c] = t;
    }
}

// ---- LayerNorm affine folded into the Linear behind it (r06: the patch embedding, Rearrange -> LayerNorm(1024) -> Linear(1024, 512); vit-pytorch
// to_patch_embedding, ctor call reference model/trajectory_model.py:54-65).  With xh = (x - mean) rstd:  Linear(gamma xh + beta) = xh (W gamma)^T + (b + W beta),
// so the engine stores xh (16-bit), multiplies by Wf = W . diag(gamma) and adds bf = b + W beta.  In the backward the weight gradient of the folded Linear,
// dWf = dY^T xh, IS the sum the LayerNorm's parameter gradients need:  dgamma_k = sum_d dWf[d][k] W[d][k],  dbeta_k = sum_d S[d] W[d][k]  (S = colsum dY = db),
// dW[d][k] = dWf[d][k] gamma_k + S[d] beta_k — exact, and the dgrad to the normalised patches (107 GFLOP, 200 us at the benchmark shape) and the LayerNorm
// backward pass over them (119 us, 620 MB) that r01-r05 ran only to reduce those 2 048 numbers are not needed: the frames take no gradient.
// W [D][K] fp32 master, gamma / beta [K]; Wf [D][K] 16-bit, bf [D] fp32.  One workgroup per output row d.
VC_KERNEL __launch_bounds__(256) void pe_fold_kernel(const float* W, const float* b, const float* gamma, const float* beta, vc_bf16* Wf, float* bf, int K) {
    VC_SHARED float red[4];
    const int d = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    for (int k = tid; k < K; k += 256) {
        const float w = W[(long)d * K + k];
        vc_st(Wf + (long)d * K + k, w * gamma[k]);
        acc += w * beta[k];
    }
    acc = vc_wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    vc_sync();
    if (tid == 0) bf[d] = b[d] + ((red[0] + red[1]) + (red[2] + red[3]));
}
// dWf [D][K] fp32 (weight gradient of the folded Linear), S [D] (its bias gradient) -> dW [D][K] and, per chunk of rows, the partial sums of dgamma | dbeta:
// partial[chunk][2 K] (the caller's column-sum pass adds the chunks in order: deterministic).  Thread = column k (coalesced along k), blockIdx.y = chunk of
// D / gridDim.y rows, eight rows' loads in flight.  (A first version walked all D rows in one thread: 4 workgroups, 209 us of dependent round trips.)
VC_KERNEL __launch_bounds__(256) void pe_fold_bwd_kernel(const float* __restrict__ dWf, const float* __restrict__ S, const float* __restrict__ W,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ dW, float* __restrict__ partial, int D, int K) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const int per = (D + gridDim.y - 1) / gridDim.y, d0 = blockIdx.y * per, d1 = d0 + per < D ? d0 + per : D;
    const float g = gamma[k], be = beta[k];
    float ag = 0.f, ab = 0.f;
    int d = d0;
    for (; d + 8 <= d1; d += 8) {
        float dwf[8], w[8], sd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { dwf[u] = dWf[(long)(d + u) * K + k]; w[u] = W[(long)(d + u) * K + k]; sd[u] = S[d + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { dW[(long)(d + u) * K + k] = dwf[u] * g + sd[u] * be; ag += dwf[u] * w[u]; ab += sd[u] * w[u]; }
    }
    for (; d < d1; ++d) {
        const float dwf = dWf[(long)d * K + k], w = W[(long)d * K + k], sd = S[d];
        dW[(long)d * K + k] = dwf * g + sd * be; ag += dwf * w; ab += sd * w;
    }
    partial[(long)blockIdx.y * 2 * K + k] = ag; partial[(long)blockIdx.y * 2 * K + K + k] = ab;
}

// ---- dst[c][r] = src[r][c] (bf16): the transposed weight shadows of the frame ViT (engine.hip: wT); 32x32 tiles through LDS
VC_KERNEL __launch_bounds__(256) void transpose_bf16_kernel(const vc_bf16* src, vc_bf16* dst, int rows, int cols) {
    VC_SHARED uint16_t tile[32 * 33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j * 33 + tx] = (r < rows && c < cols) ? src[(long)r * cols + c].bits : (uint16_t)0;
    }
    vc_sync();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (r < rows && c < cols) dst[(long)c * rows + r].bits = tile[tx * 33 + j];
    }
}

// ---- the same for up to 32 matrices in ONE grid (r06): the frame ViT's 24 transposed weight shadows were 24 launches of 5-45 us on the critical path between the stem's
// backward and the ViT's (0.19 ms per step for 126 MB of traffic).  Jobs travel by value in the kernel arguments; 64 x 64 tiles, whole 128-byte row pieces both ways.
struct TransposeBatch { int n; int tile_start[33]; long src_off[32], dst_off[32]; int rows[32], cols[32]; };
VC_KERNEL __launch_bounds__(256) void transpose_bf16_batched_kernel(const vc_bf16* S, vc_bf16* D, TransposeBatch tb) {
    VC_SHARED uint16_t tile[64 * 65];
    int j = 0;
    while (j + 1 < tb.n && tb.tile_start[j + 1] <= (int)blockIdx.x) ++j;
    const int rows = tb.rows[j], cols = tb.cols[j], t = (int)blockIdx.x - tb.tile_start[j], tcx = (cols + 63) / 64;
    const int r0 = (t / tcx) * 64, c0 = (t % tcx) * 64;
    const vc_bf16* src = S + tb.src_off[j]; vc_bf16* dst = D + tb.dst_off[j];
    if (r0 + 64 <= rows && c0 + 64 <= cols && !((rows | cols) & 1) && !(((uintptr_t)src | (uintptr_t)dst) & 3)) {
        // interior tile (every tile of the model's weights): 4-byte accesses on both sides — a lane moves two adjacent elements, a wave-instruction whole 128-byte row pieces
        // (the 2-byte form below ran the 126 MB of shadows at 1.3 TB/s)
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int i = ty; i < 64; i += 8) {
            const uint32_t w = *reinterpret_cast<const uint32_t*>(src + (long)(r0 + i) * cols + c0 + 2 * tx);
            tile[i * 65 + 2 * tx] = (uint16_t)(w & 0xffffu); tile[i * 65 + 2 * tx + 1] = (uint16_t)(w >> 16);
        }
        vc_sync();
        for (int i = ty; i < 64; i += 8) {
            const uint32_t w = (uint32_t)tile[(2 * tx) * 65 + i] | ((uint32_t)tile[(2 * tx + 1) * 65 + i] << 16);
            *reinterpret_cast<uint32_t*>(dst + (long)(c0 + i) * rows + r0 + 2 * tx) = w;
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i * 65 + tx] = (r < rows && c < cols) ? src[(long)r * cols + c].bits : (uint16_t)0;
    }
    vc_sync();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < rows && c < cols) dst[(long)c * rows + r].bits = tile[tx * 65 + i];
    }
}

// ---- grouped column sums (the decoder's deferred bias gradients): one grid over the 256-column strips of many jobs.
// pass 0: partial[job.part_off + chunk * cols + c] = sum of rows [128 chunk, 128 chunk + 128);  pass 1: out[c] = sum over chunks
// (fixed order: deterministic).  Jobs live in device memory; strip_start[j] = first strip of job j.
struct ColsumJob { const void* x; long ld; int rows, cols; float* out; int is_bf16; int strip_start; long part_off; };
VC_KERNEL __launch_bounds__(256) void colsum_grouped_kernel(const ColsumJob* jobs, int njobs, float* partial, int pass) {
    int j = 0;
    while (j + 1 < njobs && jobs[j + 1].strip_start <= (int)blockIdx.x) ++j;
    const ColsumJob job = jobs[j];
    const int c = ((int)blockIdx.x - job.strip_start) * 256 + threadIdx.x;
    if (c >= job.cols) return;
    const int nchunk = (job.rows + 127) / 128;
    if (pass == 0) {
        const int chunk = blockIdx.y;
        if (chunk >= nchunk) return;
        long r0 = (long)chunk * 128, r1 = r0 + 128; if (r1 > job.rows) r1 = job.rows;
        float s = 0.f;
        long r = r0;
        if (job.is_bf16) {
            const vc_bf16* x = (const vc_bf16*)job.x;
            for (; r + 8 <= r1; r += 8) { float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = vc_ld(x + (r + u) * job.ld + c);
                s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])); }
            for (; r < r1; ++r) s += vc_ld(x + r * job.ld + c);
        } else {
            const float* x = (const float*)job.x;
            for (; r + 8 <= r1; r += 8) { float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = x[(r + u) * job.ld + c];
                s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])); }
            for (; r < r1; ++r) s += x[r * job.ld + c];
        }
        partial[job.part_off + (long)chunk * job.cols + c] = s;
    } else {
        float s = 0.f;
        for (int k = 0; k < nchunk; ++k) s += partial[job.part_off + (long)k * job.cols + c];
        job.out[c] = s;
    }
}
