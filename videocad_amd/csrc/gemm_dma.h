// Persistent, DMA-fed bf16 GEMM for the large Linear layers (gfx950).
//
// Why a second main loop: with register staging hipcc drains vmcnt(0) before every LDS store, so a global load only ever has
// one k-tile (~0.2 us of MFMA) to land while an L2 round trip under load is ~1 us: the K-loop of gemm_kernel is latency-bound
// and every tile also pays an un-overlapped prologue.  Here the operands go HBM/L2 -> LDS by `global_load_lds_dwordx4`
// (no staging registers, no ds_write pass) into a 3-stage ring that is waited on with *counted* vmcnt and raw barriers, and the
// workgroup is persistent: it walks a list of (tile, k-slice) items as ONE flat stream of k-tiles, so the loads of the next
// tile are already in flight while the current tile's epilogue runs straight out of the accumulator registers.
//
// Shape: 512 threads = 8 waves (4 along M x 2 along N), tile 256x128, BK = 64, one workgroup per CU (ring = 3 x 48 KiB).
// LDS images are UNPADDED (the DMA writes lane-linear 1 KiB pieces) and XOR-swizzled through the per-lane SOURCE address:
//   k-contiguous operand  [rows][64]  : 16-byte slot s of row r is stored at slot s ^ ((r >> 1) & 7)  (ds_read_b128, conflict-free)
//   row-contiguous operand [64][COLS] : slot s of k-row k is stored at slot s ^ ((k & 3) << 2)          (ds_read_b64_tr_b16)
// The MFMA is issued with the operands swapped (D = B_frag x A_frag) so that each lane ends up with 4 CONSECUTIVE output
// columns per accumulator quad: the epilogue needs no LDS transpose and uses 8/16-byte accesses.
//
// The per-column bias of a tile rides along as one more (512-byte) DMA into a spare LDS corner; the per-element side input of a
// fused epilogue (residual stream, or the activation-derivative source) is fetched with ordinary loads issued BEFORE the tile's
// last MFMA phase, so its latency hides behind 16 MFMAs and the k-loops themselves contain no register-destination load at all
// (hipcc answers any such load with vmcnt(0), which would drain the DMA ring).
//
// Preconditions (checked by the dispatcher, ops_gemm.hip): bf16 operands, 16-byte aligned rows, N % 128 == 0, K % 64 == 0,
// row-contiguous operands with extent % 8 == 0, no row-broadcast add.  Anything else runs on gemm_kernel.
#pragma once
#include "gemm.h"

constexpr int GD_THREADS = 512, GD_BM = 256, GD_BN = 128, GD_BK = 64, GD_STAGES = 3;
constexpr int GD_A_ELEMS = GD_BM * GD_BK, GD_B_ELEMS = GD_BN * GD_BK, GD_STAGE_ELEMS = GD_A_ELEMS + GD_B_ELEMS;
constexpr size_t GD_RING_BYTES = (size_t)GD_STAGES * GD_STAGE_ELEMS * 2;         // 147456
constexpr size_t GD_LDS_BYTES = GD_RING_BYTES + 2 * GD_BN * sizeof(float);        // + two bias rows (double-buffered by item parity)
constexpr int GD_PIECES_A = GD_A_ELEMS * 2 / 1024, GD_PIECES_B = GD_B_ELEMS * 2 / 1024;   // 32, 16 (1 KiB each)
constexpr int GD_PW = (GD_PIECES_A + GD_PIECES_B) / 8;                          // DMA instructions per wave per stage (6)
constexpr int GD_NS = 16;                                                       // epilogue store instructions per wave per tile (2x2x4 quads)

// issue this wave's share of one operand tile: NP pieces starting at piece `p0`
template <bool TR, int EXT, int NP>
VC_DEV void gd_issue(const vc_bf16* base, long ld, int r0, int k1, int R, vc_bf16* tile, int p0, int lane) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int pc = p0 + i;
        const vc_bf16* src;
        if constexpr (!TR) {                       // piece = 8 rows x 128 B
            const int row = pc * 8 + (lane >> 3);
            const int slot = (lane & 7) ^ ((row >> 1) & 7);
            int rg = r0 + row; rg = rg < R ? rg : R - 1;                       // tail rows: re-read the last row (never stored)
            src = base + (long)rg * ld + k1 + slot * 8;
        } else {                                   // piece = 1024/(2*EXT) k-rows of EXT elements
            constexpr int SPR = EXT / 8, KPP = 64 / SPR;                        // slots per k-row, k-rows per piece
            const int k = pc * KPP + lane / SPR;
            const int slot = (lane % SPR) ^ ((k & 3) << 2);
            int c = r0 + slot * 8; c = c + 8 <= R ? c : R - 8;
            src = base + (long)(k1 + k) * ld + c;
        }
        vc_dma16(src, tile + pc * 512);
    }
}

// 8 k-values of one operand row for k-step ks (16 k) of the stage
template <bool TR, int EXT>
VC_DEV vc_s16x8 gd_frag(const vc_bf16* tile, int row0, int ks, int lane) {
    if constexpr (!TR) {
        const int row = row0 + (lane & 31);
        const int slot = (ks * 2 + (lane >> 5)) ^ ((row >> 1) & 7);
        return *reinterpret_cast<const vc_s16x8*>(tile + row * 64 + slot * 8);
    } else {
        const int i = lane & 15;
        const int k = ks * 16 + 8 * (lane >> 5) + (i >> 2);
        const int col = row0 + ((lane >> 4) & 1) * 16 + (i & 3) * 4;
        const vc_bf16* p = tile + k * EXT + (((col >> 3) ^ ((k & 3) << 2)) << 3) + (col & 7);
        const vc_s16x4 lo = vc_ds_read_tr16(p), hi = vc_ds_read_tr16(p + 4 * EXT);     // k+4 has the same (k & 3)
        vc_s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
}

struct GdCursor { int item, kt, ntc, seq; };      // item index, k-tile within it, k-tiles of that item, items started so far

// wait until at most n VMEM operations of this wave are outstanding (n wave-uniform; rounded DOWN to an encodable immediate)
VC_DEV void gd_wait_le(int n) {
    if (n >= GD_PW + 2 * GD_NS) vc_wait_vmcnt<GD_PW + 2 * GD_NS>();
    else if (n >= 2 * GD_NS) vc_wait_vmcnt<2 * GD_NS>();
    else if (n >= GD_PW + GD_NS) vc_wait_vmcnt<GD_PW + GD_NS>();
    else if (n >= GD_NS) vc_wait_vmcnt<GD_NS>();
    else if (n >= GD_PW) vc_wait_vmcnt<GD_PW>();
    else vc_wait_vmcnt<0>();
}

// fused epilogue on one accumulator quad (row m, columns n..n+3) with the side input already in registers
template <typename TO>
VC_DEV void gd_epilogue_quad(const GemmParams& p, int m, int n, float (&v)[4], const float (&b4)[4], const vc_u32x4& side) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = p.alpha * v[k] + b4[k];
    if (p.aux) quad_st<TO>(((TO*)p.aux) + (long)m * p.ldaux + n, v);
    if (p.act) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = vc_apply_act(v[k], p.act);
    }
    if (p.drop.key) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= vc_drop_mul(p.drop, (long)m * p.N + n + k);
    }
    if (p.dact_src) {
        float s4[4];
        if constexpr (sizeof(TO) == 2) {
            s4[0] = vc_bits_f32(side.x << 16); s4[1] = vc_bits_f32(side.x & 0xffff0000u);
            s4[2] = vc_bits_f32(side.y << 16); s4[3] = vc_bits_f32(side.y & 0xffff0000u);
        } else {
            s4[0] = vc_bits_f32(side.x); s4[1] = vc_bits_f32(side.y); s4[2] = vc_bits_f32(side.z); s4[3] = vc_bits_f32(side.w);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = vc_apply_dact(v[k], s4[k], p.dact_kind);
    }
    if (p.residual) {
        v[0] += vc_bits_f32(side.x); v[1] += vc_bits_f32(side.y); v[2] += vc_bits_f32(side.z); v[3] += vc_bits_f32(side.w);
    }
    quad_st<TO>(((TO*)p.C) + (long)m * p.ldc + n, v);
}

template <typename TO, bool TRA, bool TRB>
VC_KERNEL __launch_bounds__(GD_THREADS, 1) void gemm_dma_kernel(GemmParams p, int tiles_n, int tiles_mn, int nsplit, int total) {
    VC_DYN_SHARED(vc_bf16, lds);
    float* bias_lds = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(lds) + GD_RING_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const vc_bf16* Ag = (const vc_bf16*)p.A;
    const vc_bf16* Bg = (const vc_bf16*)p.B;

    // this workgroup's items: XCD x (= block id & 7, the hardware's round-robin) owns one contiguous chunk of the item list
    // (items ordered k-slice major, then tile_m, tile_n fastest), and its workgroups sweep that chunk interleaved — at any
    // moment one XCD's L2 serves neighbouring tiles that share A panels / the same k-slice.
    const int G = gridDim.x, b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int nbx = (G + 7 - xcd) >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int cs = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cn = xcd < r8 ? q8 + 1 : q8;
    const int first = cs + j, last = cs + cn;                    // items first, first + nbx, ... < last
    const int nt = p.k_per_split / GD_BK, ktiles = p.K / GD_BK;  // k-tiles per slice (the last slice may be shorter)
    const bool use_bias = p.bias && !p.partial;
    const bool use_side = (p.residual || p.dact_src) && !p.partial;

    auto issue = [&](const GdCursor& c, int slot) {
        const int z = c.item / tiles_mn, rem = c.item - z * tiles_mn;
        const int tm = rem / tiles_n, tn = rem - tm * tiles_n;
        const int k1 = z * p.k_per_split + c.kt * GD_BK;
        vc_bf16* st = lds + slot * GD_STAGE_ELEMS;
        // the tile's 128 bias values: issued AHEAD of the item's first stage, so the wait that retires that stage covers them
        if (use_bias && c.kt == 0 && wave == 0 && lane < GD_BN / 4) vc_dma16(p.bias + tn * GD_BN + lane * 4, bias_lds + (c.seq & 1) * GD_BN);
        gd_issue<TRA, GD_BM, GD_PIECES_A / 8>(Ag, p.lda, tm * GD_BM, k1, p.M, st, wave * (GD_PIECES_A / 8), lane);
        gd_issue<TRB, GD_BN, GD_PIECES_B / 8>(Bg, p.ldb, tn * GD_BN, k1, p.N, st + GD_A_ELEMS, wave * (GD_PIECES_B / 8), lane);
    };
    auto slice_tiles = [&](int item) { const int rest = ktiles - (item / tiles_mn) * nt; return rest < nt ? rest : nt; };
    auto advance = [&](GdCursor& c) {
        if (++c.kt == c.ntc) { c.kt = 0; c.item += nbx; ++c.seq; if (c.item < last) c.ntc = slice_tiles(c.item); }
    };

    GdCursor pf{first, 0, first < last ? slice_tiles(first) : 1, 0}, cp = pf;
    int ahead = 0;                                               // stages issued and not yet consumed
    for (; ahead < GD_STAGES - 1 && pf.item < last; ++ahead) { issue(pf, ahead); advance(pf); }

    vc_f32x16 acc[2][2];
    vc_u32x4 side[2][2][4];
    int slot = 0, young_prev = 0, young_cur = 0;                 // stores issued in the previous / current iteration (lower bounds)
    while (cp.item < last) {
        // stage `slot` must have landed.  VMEM retires in issue order on gfx9, so "at most N outstanding" with N = the number
        // of operations issued AFTER this stage's DMA — the younger stage's pieces and the epilogue stores of the last two
        // iterations — is exact: neither the prefetch nor the stores are waited for.  N must never over-count (that would let
        // the stage itself still be in flight), so only guaranteed stores (interior tile, the C quads) are counted.
        gd_wait_le((ahead >= 2 ? GD_PW : 0) + young_prev + young_cur);
        vc_barrier_raw();                                        // everyone's pieces landed; everyone is done reading slot-1
        young_prev = young_cur; young_cur = 0;

        const bool fin = cp.kt == cp.ntc - 1;
        const int z = cp.item / tiles_mn, rem = cp.item - z * tiles_mn;
        const int tm = rem / tiles_n, tn = rem - tm * tiles_n;
        if (fin && use_side) {
            // side input of the fused epilogue: requested now, consumed after this phase's 16 MFMAs
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int m = tm * GD_BM + wm * 64 + i * 32 + (lane & 31); m = m < p.M ? m : p.M - 1;
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = tn * GD_BN + wn * 64 + jn * 32 + 8 * q + 4 * (lane >> 5);
                        if (p.residual) side[i][jn][q] = *reinterpret_cast<const vc_u32x4*>(p.residual + (long)m * p.ldr + n);
                        else if constexpr (sizeof(TO) == 2) {
                            const vc_u32x2 t = *reinterpret_cast<const vc_u32x2*>(((const TO*)p.dact_src) + (long)m * p.lddact + n);
                            side[i][jn][q].x = t.x; side[i][jn][q].y = t.y;
                        } else side[i][jn][q] = *reinterpret_cast<const vc_u32x4*>(((const TO*)p.dact_src) + (long)m * p.lddact + n);
                    }
            }
        }
        if (pf.item < last) { issue(pf, slot == 0 ? GD_STAGES - 1 : slot - 1); advance(pf); } else --ahead;

        if (cp.kt == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;
        }
        const vc_bf16* a_tile = lds + slot * GD_STAGE_ELEMS;
        const vc_bf16* b_tile = a_tile + GD_A_ELEMS;
#pragma unroll
        for (int ks = 0; ks < GD_BK / 16; ++ks) {
            vc_s16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = gd_frag<TRA, GD_BM>(a_tile, wm * 64 + i * 32, ks, lane);
                bf[i] = gd_frag<TRB, GD_BN>(b_tile, wn * 64 + i * 32, ks, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) acc[i][jn] = vc_mfma_32x32x16_bf16(bf[jn], af[i], acc[i][jn]);   // swapped: D[n][m]
        }

        if (fin) {
            // epilogue from registers: lane holds row m = ..+(lane&31), columns n = ..+8*q+4*(lane>>5)+{0..3} for q = 0..3
            const float* brow = bias_lds + (cp.seq & 1) * GD_BN + wn * 64 + 4 * (lane >> 5);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = tm * GD_BM + wm * 64 + i * 32 + (lane & 31);
                if (m < p.M) {
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int n = tn * GD_BN + wn * 64 + jn * 32 + 8 * q + 4 * (lane >> 5);
                            float v[4] = {acc[i][jn][4 * q], acc[i][jn][4 * q + 1], acc[i][jn][4 * q + 2], acc[i][jn][4 * q + 3]};
                            if (p.partial) {
                                quad_st<float>(p.partial + ((long)z * p.M + m) * p.N + n, v);
                            } else {
                                float b4[4] = {0.f, 0.f, 0.f, 0.f};
                                if (use_bias) quad_ld_f32(brow + jn * 32 + 8 * q, b4);
                                gd_epilogue_quad<TO>(p, m, n, v, b4, side[i][jn][q]);
                            }
                        }
                }
            }
            if ((tm + 1) * GD_BM <= p.M && !(p.debug_skip & 32)) young_cur = GD_NS;
        }
        advance(cp);
        slot = slot == GD_STAGES - 1 ? 0 : slot + 1;
    }
}
