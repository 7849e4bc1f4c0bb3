// gemm_mx8.h — the fp8 mode of the big ViT Linears (BASELINE configs[4]): OCP e4m3 operands with one E8M0
// power-of-two scale per 32 consecutive k-values ("MXFP8"), multiplied by the gfx950 block-scaled MFMA
// (v_mfma_scale_f32_32x32x64_f8f6f4: fp32 accumulate, twice the bf16 MFMA rate, half the operand bytes through the L2 -> LDS path
// that bounds the bf16 kernels).  Block scaling is LOCAL — a quantiser needs no tensor-wide amax reduction, so activations are
// quantised in one pass by mx8_quant_kernel and weights whenever their bf16 shadow is refreshed.
//
//   C[m,n] = epilogue( sum_k A8[m,k] 2^(sa[m,k/32]-127) * B8[n,k] 2^(sb[n,k/32]-127) )        A8: [M,K] e4m3, B8: [N,K] e4m3 (a Linear weight)
//
// Kernel: 128 x 128 tile, 256 threads (4 waves as 2 x 2, 64 x 64 each = 2 x 2 accumulator tiles), one k-tile = 128 k = 128 BYTES per
// row (the same 128-byte rows as the bf16 kernels' 64 k: same DMA pieces, same XOR swizzle, two ds_read_b128 per fragment), LDS
// double buffer fed by global_load_lds_dwordx4, two workgroups per CU.  The scale bytes are one dword per (row, k-tile), loaded
// straight to registers.  Epilogue: gemm.h's column-per-lane gemm_epilogue_tile (bias, GELU + pre-activation output, dropout,
// residual ...), so every fused forward epilogue of the ViT is available.
#pragma once
#include "ops.h"

constexpr int MX_BM = 128, MX_BN = 128, MX_BK = 128, MX_THREADS = 256;
constexpr int MX_TILE_BYTES = MX_BM * MX_BK;                                   // 16 KiB per operand tile
constexpr size_t MX_LDS_BYTES = 2ul * 2 * MX_TILE_BYTES;                       // double buffer x (A + B) = 64 KiB

// (struct Mx8Params: ops.h — GemmParams whose A / B point at the e4m3 bytes, plus the E8M0 scale matrices [rows][K / 32])

// ---- quantiser: x [rows, cols] (fp32 or bf16, row stride ld) -> q [rows, cols] e4m3 + scales [rows, cols/32] (E8M0)
// scale exponent = floor(log2(amax of the block)) - 8 (e4m3's largest binade), so the block's largest value lands in [256, 512) and is
// clamped to 448 (the OCP MX rule); an all-zero block gets scale 2^-127.  One lane = 8 consecutive elements, 4 lanes = one block.
template <typename TX>
VC_KERNEL __launch_bounds__(256) void mx8_quant_kernel(const TX* x, long ld, uint8_t* q, uint8_t* sc, long rows, int cols) {
    const long lanes_per_row = cols / 8;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = gid < rows * lanes_per_row;
    const long r = live ? gid / lanes_per_row : 0; const int c = live ? (int)(gid % lanes_per_row) * 8 : 0;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = live ? vc_ld(x + r * ld + c + j) : 0.0f;
    float am = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf(v[j]));
    am = fmaxf(am, vc_shfl_xor(am, 1)); am = fmaxf(am, vc_shfl_xor(am, 2));      // the 4 lanes of a block are adjacent (cols % 32 == 0)
    int e = -127;
    if (am > 0.f) { int ex; (void)frexpf(am, &ex); e = ex - 1 - 8; e = e < -127 ? -127 : (e > 127 ? 127 : e); }
    const float inv = ldexpf(1.0f, -e);
    if (live) {
        vc_u32x2 o;
        o.x = vc_cvt_pk_e4m3(v[0] * inv, v[1] * inv) | (vc_cvt_pk_e4m3(v[2] * inv, v[3] * inv) << 16);
        o.y = vc_cvt_pk_e4m3(v[4] * inv, v[5] * inv) | (vc_cvt_pk_e4m3(v[6] * inv, v[7] * inv) << 16);
        *reinterpret_cast<vc_u32x2*>(q + r * cols + c) = o;
        if ((c & 31) == 0) sc[r * (cols / 32) + (c >> 5)] = (uint8_t)(e + 127);
    }
}

// The operand of one MFMA (64 k = two 32-k scale blocks) for lane l = (row r = l & 31, half h = l >> 5), as probed on the hardware
// (tools/probe_mx8_layout.hip): registers 0-3 hold k = 16 h .. 16 h + 15 of scale block 0, registers 4-7 hold k = 32 + 16 h .. of scale
// block 1; the scale of block b of row r is taken from lane r + 32 b.  In 16-byte slots of the 128-byte row: ks*4 + h and ks*4 + 2 + h.
VC_DEV vc_i32x8 mx8_frag(const uint8_t* tile, int row0, int ks, int lane) {
    const int row = row0 + (lane & 31);
    const int s0 = ks * 4 + (lane >> 5), sw = (row >> 1) & 7;                     // 16-byte slots, XOR-swizzled like the bf16 k-contiguous image
    const vc_u32x4 lo = *reinterpret_cast<const vc_u32x4*>(tile + row * MX_BK + ((s0 ^ sw) << 4));
    const vc_u32x4 hi = *reinterpret_cast<const vc_u32x4*>(tile + row * MX_BK + (((s0 + 2) ^ sw) << 4));
    vc_i32x8 r;
    r[0] = (int)lo.x; r[1] = (int)lo.y; r[2] = (int)lo.z; r[3] = (int)lo.w; r[4] = (int)hi.x; r[5] = (int)hi.y; r[6] = (int)hi.z; r[7] = (int)hi.w;
    return r;
}

template <typename TO>
VC_KERNEL __launch_bounds__(MX_THREADS, 2) void gemm_mx8_kernel(Mx8Params q) {
    const GemmParams& p = q.g;
    VC_DYN_SHARED(uint8_t, lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = vc_uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: consecutive tiles of one XCD share the A panel (tn fastest)
    const int tiles_n = p.N / MX_BN, ntile = (int)gridDim.x;
    const int b = blockIdx.x, xcd = b & 7, nx = (ntile + 7 - xcd) >> 3;
    const int q8 = ntile >> 3, r8 = ntile & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    (void)nx;
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const uint8_t* Ag = (const uint8_t*)p.A; const uint8_t* Bg = (const uint8_t*)p.B;
    // DMA pieces: 16 per operand tile (8 rows x 128 B), 8 per wave; per-lane source offsets (swizzle applied through the SOURCE address)
    uint32_t offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pc = wave * 4 + i, row = pc * 8 + (lane >> 3), slot = (lane & 7) ^ ((row >> 1) & 7);
        int ra = tm * MX_BM + row; ra = ra < p.M ? ra : p.M - 1;
        offA[i] = (uint32_t)((long)ra * p.lda + slot * 16);
        offB[i] = (uint32_t)((long)(tn * MX_BN + row) * p.ldb + slot * 16);
    }
    auto issue = [&](int kt, int buf) {
        uint8_t* ta = lds + buf * 2 * MX_TILE_BYTES; uint8_t* tb = ta + MX_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) { vc_dma16(Ag + (long)kt * MX_BK + offA[i], ta + (wave * 4 + i) * 1024); vc_dma16(Bg + (long)kt * MX_BK + offB[i], tb + (wave * 4 + i) * 1024); }
    };
    vc_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;
    const int ktiles = p.K / MX_BK;
    // this lane's scale rows: A rows tm*128 + wm*64 + i*32 + (lane & 31), B rows tn*128 + wn*64 + jn*32 + (lane & 31)
    long sra[2], srb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int ra = tm * MX_BM + wm * 64 + i * 32 + (lane & 31); ra = ra < p.M ? ra : p.M - 1;
        sra[i] = (long)ra * q.ldsa; srb[i] = (long)(tn * MX_BN + wn * 64 + i * 32 + (lane & 31)) * q.ldsb;
    }
    issue(0, 0);
    for (int kt = 0; kt < ktiles; ++kt) {
        const int buf = kt & 1;
        // scales of this k-tile: 4 bytes per row, shifted so that byte 0 / 2 are this lane-half's blocks of k-steps 0 / 1
        int sa[2], sb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            sa[i] = (int)(*reinterpret_cast<const uint32_t*>(q.sa + sra[i] + kt * 4) >> (8 * (lane >> 5)));
            sb[i] = (int)(*reinterpret_cast<const uint32_t*>(q.sb + srb[i] + kt * 4) >> (8 * (lane >> 5)));
        }
        vc_wait_vmcnt<0>();                     // this k-tile's DMA (and the scale loads) have landed
        vc_sync();                              // ... for every wave; everyone is done reading the other buffer
        if (kt + 1 < ktiles) issue(kt + 1, buf ^ 1);
        const uint8_t* ta = lds + buf * 2 * MX_TILE_BYTES; const uint8_t* tb = ta + MX_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            vc_i32x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { af[i] = mx8_frag(ta, wm * 64 + i * 32, ks, lane); bf[i] = mx8_frag(tb, wn * 64 + i * 32, ks, lane); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    if (ks == 0) acc[i][jn] = vc_mfma_mx8_32x32x64<0, 0>(af[i], bf[jn], acc[i][jn], sa[i], sb[jn]);
                    else acc[i][jn] = vc_mfma_mx8_32x32x64<2, 2>(af[i], bf[jn], acc[i][jn], sa[i], sb[jn]);
                }
        }
    }
    // ---- epilogue: column-per-lane (lane = column n, 16 rows per accumulator tile)
    const bool interior = (tm + 1) * MX_BM <= p.M;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int n = tn * MX_BN + wn * 64 + jn * 32 + (lane & 31);
            const int mbase = tm * MX_BM + wm * 64 + i * 32 + 4 * (lane >> 5);
            const float bn = p.bias ? p.bias[n] : 0.0f;
            if (interior) gemm_epilogue_tile<TO>(p, mbase, n, acc[i][jn], bn);
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { const int m = mbase + (r & 3) + 8 * (r >> 2); if (m < p.M) gemm_epilogue_store<TO>(p, m, n, acc[i][jn][r], bn); }
            }
        }
}
