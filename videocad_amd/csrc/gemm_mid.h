// gemm_mid.h — DMA-ring GEMM for the 2 000-3 000-row Linears of the decoder (and the CAD ViT): one output tile per workgroup,
// a SIX-stage LDS ring fed by global_load_lds_dwordx4.
//
// Why a third GEMM: a 2048 x 1024 x 1024 Linear is 4.3 GFLOP — ~4 us of matrix-core time — but ran 21-37 us on the register-staged
// kernel (profiles/r02b: 190 such launches, ~3.8 ms of an 8.3 ms decoder phase).  Its operands were written a few microseconds
// earlier by another kernel, i.e. they sit in ANOTHER XCD's L2 or in the Infinity Cache: every k-tile of every workgroup is a cold
// miss, and with a two-tile register prefetch each of the 16 k-iterations waits ~half a fabric round trip.  The persistent kernel
// (gemm_dma.h) does not apply: 64 of its 256 x 128 tiles leave three quarters of the chip idle.  Here a workgroup owns a 128 x 64
// (k-contiguous B: forward, W^T) or 64 x 128 (row-contiguous B: dgrad through W) tile — 256 tiles for N = 1024 — and keeps FIVE
// 24 KiB stages (120 KiB) in flight while the sixth is on the matrix cores: the fabric latency is paid once per workgroup, not per
// k-tile.  LDS images, swizzles and fragment reads are gemm_dma.h's; the epilogue is gemm.h's row-wise fused epilogue on the fp32
// tile staged through the (by then idle) ring, so every fused variant (bias, residual, ReLU / dropout, activation derivative, pre-
// activation output, row-broadcast add) is the same code as in the other kernels.
//
// Preconditions (dispatcher): bf16 A (k-contiguous) and B, 16-byte aligned rows, K % 64 == 0, N % BN == 0, vecC.
#pragma once
#include "gemm_dma.h"

constexpr int GM_THREADS = 256, GM_BK = 64, GM_STAGES = 6;
template <int BM, int BN> struct GmTile {
    static constexpr int A_ELEMS = BM * GM_BK, B_ELEMS = BN * GM_BK, STAGE_ELEMS = A_ELEMS + B_ELEMS;
    static constexpr int NPA = A_ELEMS * 2 / 1024 / 4, NPB = B_ELEMS * 2 / 1024 / 4;       // 1 KiB pieces per wave per stage
    static constexpr int PW = NPA + NPB;
    static constexpr size_t RING_BYTES = (size_t)GM_STAGES * STAGE_ELEMS * 2;
    static constexpr int ES = BN + 4;                                                       // fp32 row stride of the epilogue tile
    static constexpr size_t LDS_BYTES = RING_BYTES > (size_t)BM * ES * 4 ? RING_BYTES : (size_t)BM * ES * 4;
    static constexpr int MI = BM / 64, NJ = BN / 64;                                        // 32 x 32 accumulator tiles per wave (2 x 2 waves)
};

template <int N> VC_DEV void gm_wait_stages(int younger) {      // at most `younger` stages (N pieces each) of this wave may still be in flight
    if (younger >= 4) vc_wait_vmcnt<4 * N>();
    else if (younger == 3) vc_wait_vmcnt<3 * N>();
    else if (younger == 2) vc_wait_vmcnt<2 * N>();
    else if (younger == 1) vc_wait_vmcnt<N>();
    else vc_wait_vmcnt<0>();
}

template <typename TO, bool TRB, int BM, int BN>
VC_KERNEL __launch_bounds__(GM_THREADS, 1) void gemm_mid_kernel(GemmParams p) {
    if (p.batch > 1) {                                       // batched launch: blockIdx.y picks the problem (wave-uniform pointer arithmetic)
        const long bz = blockIdx.y;
        p.A = (const unsigned char*)p.A + bz * p.bsa * 2; p.B = (const unsigned char*)p.B + bz * p.bsb * 2;
        p.C = (unsigned char*)p.C + bz * p.bsc * (long)sizeof(TO);
    }
    using TL = GmTile<BM, BN>;
    constexpr int MI = TL::MI, NJ = TL::NJ, NPA = TL::NPA, NPB = TL::NPB, ES = TL::ES;
    VC_DYN_SHARED(vc_bf16, lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = vc_uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order (as gemm.h): an XCD gets a contiguous run of tiles, n fastest — the workgroups sharing an A panel share an L2
    const int nx = p.N / BN, ny = VC_CEIL_DIV(p.M, BM);
    int tile_m, tile_n;
    {
        const int total = nx * ny, bid = blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tile_m = t / nx; tile_n = t - tile_m * nx;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const unsigned char* Ag = (const unsigned char*)p.A;
    const unsigned char* Bg = (const unsigned char*)p.B;
    uint32_t offA[NPA], offB[NPB];
    gd_offsets<false, BM, NPA>(offA, p.lda, m0, p.M, wave * NPA, lane);
    gd_offsets<TRB, BN, NPB>(offB, p.ldb, n0, p.N, wave * NPB, lane);
    const long kstepA = (long)GM_BK * 2, kstepB = TRB ? (long)GM_BK * p.ldb * 2 : (long)GM_BK * 2;
    const int nt = p.K / GM_BK;
    auto issue = [&](int kt) {
        vc_bf16* st = lds + (kt % GM_STAGES) * TL::STAGE_ELEMS;
        gd_issue<NPA>(Ag + kt * kstepA, offA, st, wave * NPA);
        gd_issue<NPB>(Bg + kt * kstepB, offB, st + TL::A_ELEMS, wave * NPB);
    };
    for (int s0 = 0; s0 < GM_STAGES - 1 && s0 < nt; ++s0) issue(s0);

    vc_f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    for (int kt = 0; kt < nt; ++kt) {
        // stage kt must have landed; the stages issued after it (kt+1 .. min(kt + STAGES - 2, nt - 1)) stay in flight
        const int younger = (nt - 1 - kt) < (GM_STAGES - 2) ? (nt - 1 - kt) : (GM_STAGES - 2);
        gm_wait_stages<TL::PW>(younger);
        vc_barrier_raw();                                   // everyone's pieces landed; everyone is done reading the slot of stage kt - 1
        if (kt + GM_STAGES - 1 < nt) issue(kt + GM_STAGES - 1);
        const vc_bf16* a_tile = lds + (kt % GM_STAGES) * TL::STAGE_ELEMS;
        const vc_bf16* b_tile = a_tile + TL::A_ELEMS;
#pragma unroll
        for (int ks = 0; ks < GM_BK / 16; ++ks) {
            vc_s16x8 af[MI], bf[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = gd_frag<false, BM>(a_tile, wm * (BM / 2) + i * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = gd_frag<TRB, BN>(b_tile, wn * (BN / 2) + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = vc_mfma_32x32x16_bf16(af[i], bf[j], acc[i][j]);
        }
    }
    vc_wait_vmcnt<0>();
    vc_sync();                                              // the ring is idle: it becomes the fp32 tile of the row-wise epilogue
    float* et = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                et[(wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * ES + wn * (BN / 2) + j * 32 + (lane & 31)] = acc[i][j][r];
    vc_sync();
    constexpr int TPR = BN / 4, RPP = GM_THREADS / TPR;     // threads per row, rows per pass
    const int c4 = (tid % TPR) * 4, n = n0 + c4;
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) quad_ld_f32(p.bias + n, b4);
#pragma unroll 4
    for (int pass = 0; pass < BM / RPP; ++pass) {
        const int row = pass * RPP + tid / TPR, m = m0 + row;
        if (m < p.M) {
            float v[4];
            quad_ld_f32(et + row * ES + c4, v);
            gemm_epilogue_quad<TO>(p, m, n, v, b4);
        }
    }
}
