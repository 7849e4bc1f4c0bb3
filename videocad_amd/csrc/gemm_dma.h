// Persistent, DMA-fed bf16 GEMM for the large Linear layers (gfx950).
//
// Why a second main loop: with register staging hipcc drains vmcnt(0) before every LDS store, so a global load only ever has
// one k-tile (~0.2 us of MFMA) to land while an L2 round trip under load is ~1 us: the K-loop of gemm_kernel is latency-bound
// and every tile also pays an un-overlapped prologue.  Here the operands go HBM/L2 -> LDS by `global_load_lds_dwordx4`
// (no staging registers, no ds_write pass) into a 3-stage ring that is waited on with *counted* vmcnt and raw barriers, and the
// workgroup is persistent: it walks a list of (tile, k-slice) items as ONE flat stream of k-tiles, so the loads of the next
// tile are already in flight while the current tile's epilogue runs straight out of the accumulator registers.
//
// Shapes (GdTile<BN>): 512 threads = 8 waves (4 along M x 2 along N), BK = 64, one workgroup per CU;
//   tile 256 x 128, ring = 3 x 48 KiB — the fused residual / activation-derivative epilogues;
//   tile 256 x 256, ring = 2 x 64 KiB (r02) — plain epilogues (bias, k-slice slabs): half the operand bytes per MFMA.
// LDS images are UNPADDED (the DMA writes lane-linear 1 KiB pieces) and XOR-swizzled through the per-lane SOURCE address:
//   k-contiguous operand  [rows][64]  : 16-byte slot s of row r is stored at slot s ^ ((r >> 1) & 7)  (ds_read_b128, conflict-free)
//   row-contiguous operand [64][COLS] : slot s of k-row k is stored at slot s ^ ((k & 3) << 2)          (ds_read_b64_tr_b16)
// Epilogue forms: "row per lane" — the MFMA is issued with the operands swapped (D = B_frag x A_frag) so that a lane ends up with
// one output row and 4 CONSECUTIVE columns per accumulator quad (8/16-byte accesses, no LDS transpose); "column per lane" (r02,
// k-contiguous B on the wide tile) — unswapped MFMA over a B image whose rows were permuted by the DMA's source addresses, so a lane
// owns NJ ADJACENT columns of 16 rows and a store instruction covers 2 rows x 256-512 contiguous bytes (see the kernel).
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
constexpr size_t GD_LDS_BYTES = GD_RING_BYTES + 3 * GD_BN * sizeof(float);        // + three bias rows (the prefetch runs up to two items ahead)
constexpr int GD_PIECES_A = GD_A_ELEMS * 2 / 1024, GD_PIECES_B = GD_B_ELEMS * 2 / 1024;   // 32, 16 (1 KiB each)
constexpr int GD_PW = (GD_PIECES_A + GD_PIECES_B) / 4;                          // DMA instructions per issuing wave per stage (12)
constexpr int GD_NS = 16;                                                       // epilogue store instructions per wave per tile (2x2x4 quads)
// The same kernel with a 256 x 256 tile (GdTile<256>): what bounds the 256 x 128 kernel on every large shape is the L2 -> LDS DMA
// rate of a CU (~20 B/clk measured: 48 KiB per k-tile against 1 024 MFMA cycles caps it at ~40 % — 998 TF/s at 8192^3, r02
// profiles), not MFMA issue, so the lever is fewer operand bytes per MFMA: 64 KiB per 2 048 MFMA cycles.  Two 64 KiB stages fill
// the LDS (one stage in flight while the other is consumed); each wave owns 64 x 128 outputs (128 accumulator registers), which
// leaves no room for the 128 side-input registers a fused residual / activation-derivative epilogue would need: plain epilogues
// (bias, k-slice slabs) only — the QKV forward, every dgrad through W^T, the split-K weight gradients.
// Ablation builds (tools/gemm_ablate.sh, never the product library): GD_ABLATE bit 1 = no MFMA (fragments still read), 2 = no fragment
// reads (MFMA on stale registers), 4 = no operand DMA.  Results are garbage by construction; only the time is looked at.
#ifndef GD_ABLATE
#define GD_ABLATE 0
#endif
// BK (r05): k-tile depth of a ring stage: 64 in every product instantiation.  The A/B build also carries the 256-wide weight-gradient kernel on 32-deep stages in a
// FOUR-stage ring (three stages = 96 KiB in flight per CU instead of one 64 KiB stage): the hypothesis was that a token reduction, which streams both operands from HBM
// with nothing to re-use, runs at the pace of the bytes in flight.  It does not: 172 -> 211 us per launch in the model (profiles/r05_wgrad_bk32_ab.txt) — twice the
// barriers, waits and DMA-issue turns per byte cost more than the deeper queue buys.  Only the row-contiguous (tr-read) layouts can take it without new LDS images
// ([k][columns]: a 32-deep stage is the first 32 k-rows of the 64-deep one).  The ring bookkeeping below is written for any stage count.
template <int BN, int BK = 64> struct GdTile {
    static_assert(BK == 64 || (BK == 32 && BN == 256), "32-deep stages: 256-wide tile only");
    static constexpr int STAGES = BN == 128 ? 3 : (BK == 32 ? 4 : 2);
    static constexpr int NJ = BN / 64;                                          // 32-column accumulator tiles per wave (2 or 4)
    static constexpr int A_ELEMS = GD_BM * BK, B_ELEMS = BN * BK, STAGE_ELEMS = A_ELEMS + B_ELEMS;
    static constexpr size_t RING_BYTES = (size_t)STAGES * STAGE_ELEMS * 2;
    static constexpr int NBIAS = STAGES < 3 ? 3 : STAGES;                        // bias rows kept in LDS: the prefetch cursor runs up to STAGES - 1 items ahead of the consumer
    static constexpr size_t LDS_BYTES = RING_BYTES + NBIAS * BN * sizeof(float) + 8 * sizeof(int);     // + bias rows + the ticket ring of the dynamic claiming
    static constexpr int PIECES_A = A_ELEMS * 2 / 1024, PIECES_B = B_ELEMS * 2 / 1024;
    static constexpr int PW = (PIECES_A + PIECES_B) / 4;                        // 12 / 16 / 8
    static constexpr int NS = 2 * NJ * 4;                                       // 16 / 32
};

// Per-lane byte offsets of this wave's NP pieces of one operand tile (computed once per item; the k position of a stage is a
// wave-uniform base added by the scalar unit, so issuing a stage costs ~3 instructions per piece).
// PERM = NJ > 0 (B operand of the column-per-lane epilogue): inside every panel of NJ * 32 rows, LDS row rho holds operand row
// NJ * (rho % 32) + rho / 32, so the fragment of accumulator tile jn (LDS rows 32 jn .. 32 jn + 31, conflict-free as before)
// carries output columns NJ c + jn for lane c: a lane ends up with NJ ADJACENT output columns.
template <bool TR, int EXT, int NP, int PERM = 0>
VC_DEV void gd_offsets(uint32_t (&off)[NP], long ld, int r0, int R, int p0, int lane) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int pc = p0 + i;
        if constexpr (!TR) {                       // piece = 8 rows x 128 B
            const int row = pc * 8 + (lane >> 3);
            const int slot = (lane & 7) ^ ((row >> 1) & 7);
            int grow = row;
            if constexpr (PERM > 0) { const int w = row % (PERM * 32); grow = row - w + PERM * (w % 32) + w / 32; }
            int rg = r0 + grow; rg = rg < R ? rg : R - 1;                      // tail rows: re-read the last row (never stored)
            off[i] = (uint32_t)(((long)rg * ld + slot * 8) * 2);
        } else {                                   // piece = 1024/(2*EXT) k-rows of EXT elements
            constexpr int SPR = EXT / 8, KPP = 64 / SPR;                        // slots per k-row, k-rows per piece
            const int k = pc * KPP + lane / SPR;
            const int slot = (lane % SPR) ^ ((k & 3) << 2);
            int c = r0 + slot * 8; c = c + 8 <= R ? c : R - 8;
            off[i] = (uint32_t)(((long)k * ld + c) * 2);
        }
    }
}
template <int NP>
VC_DEV void gd_issue(const unsigned char* ubase, const uint32_t (&off)[NP], vc_bf16* tile, int p0) {
#pragma unroll
    for (int i = 0; i < NP; ++i) vc_dma16(ubase + off[i], tile + (p0 + i) * 512);
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

// NJ (2 or 4) adjacent columns of one row: one 4 / 8 / 16-byte access
template <typename T, int NJ> VC_DEV void gd_vec_st(T* q, const float (&v)[NJ]) {
    if constexpr (sizeof(T) == 4 && NJ == 4) { vc_u32x4 t; t.x = vc_f32_bits(v[0]); t.y = vc_f32_bits(v[1]); t.z = vc_f32_bits(v[2]); t.w = vc_f32_bits(v[3]); *reinterpret_cast<vc_u32x4*>(q) = t; }
    else if constexpr (sizeof(T) == 4) { vc_u32x2 t; t.x = vc_f32_bits(v[0]); t.y = vc_f32_bits(v[1]); *reinterpret_cast<vc_u32x2*>(q) = t; }
    else if constexpr (NJ == 4) { vc_u32x2 t; t.x = vc_pack_bf16x2(v[0], v[1]); t.y = vc_pack_bf16x2(v[2], v[3]); *reinterpret_cast<vc_u32x2*>(q) = t; }
    else *reinterpret_cast<uint32_t*>(q) = vc_pack_bf16x2(v[0], v[1]);
}
template <int NJ> VC_DEV void gd_vec_ld_f32(const float* q, float (&v)[NJ]) {
    if constexpr (NJ == 4) { const vc_u32x4 t = *reinterpret_cast<const vc_u32x4*>(q); v[0] = vc_bits_f32(t.x); v[1] = vc_bits_f32(t.y); v[2] = vc_bits_f32(t.z); v[3] = vc_bits_f32(t.w); }
    else { const vc_u32x2 t = *reinterpret_cast<const vc_u32x2*>(q); v[0] = vc_bits_f32(t.x); v[1] = vc_bits_f32(t.y); }
}

template <int I, int N, typename F> VC_DEV void gd_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); gd_static_for<I + 1, N>(f); }
}
// scheduling directive for the region it ends: NM MFMAs, each followed by its share of ND LDS reads
template <int NM, int ND> VC_DEV void gd_interleave() {
#ifndef VC_EMU
    gd_static_for<0, NM>([](auto m) {
        constexpr int M = decltype(m)::value, n = (ND * (M + 1)) / NM - (ND * M) / NM;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (n > 0) __builtin_amdgcn_sched_group_barrier(0x100, n, 0);
    });
#endif
}

struct GdCursor { int item, kt, ntc, seq, z, tn, row0, mend, li, pb; };   // item, k-tile within it, its k-tile count, items started, (k-slice, tile column, first row, row bound), index in the workgroup's static list, problem (GdBatch)
// Mini tiles (r06).  800 tiles on 256 CUs are 3.125 rounds: 32 workgroups walk a fourth tile while 224 idle (the N = 512 Linears of the ViT at the benchmark
// shape: a fifth of 30 launches per step).  With h > 0 the rows from tile row tm0 on are cut into pieces of h (64 / 128) rows instead: the same tile program on a
// tile of which only the first h rows exist — the A rows beyond are clamped re-reads of the piece's last row (one cache line per DMA piece), the waves that
// own them skip their fragment reads, MFMAs and stores but keep issuing their share of the ring and meeting the barriers.  A mini costs about a third of a full
// tile (its B tile and eight k-tiles of two to four waves), so the launch ends after 3 rounds + one mini per workgroup instead of 4 rounds.
// Item ids: [0, nfull) = full tiles (k-slice, tile row, tile column), [nfull, nfull + nmini) = minis (piece, tile column); every XCD's list is its chunk of the full
// tiles followed by its chunk of the minis.  h = 0: no minis (nfull = total).
struct GdMini { int tm0, h, nfull, nmini; };
// Several weight-gradient problems in ONE launch (r06; tr-read layouts, k-slice slabs only).  A ViT layer's three small weight gradients — net.4 and net.0 (512 x 512)
// and to_out (512 x 1 024), all token reductions over the same 102 400 rows — ran as three launches of 256 items with 25-50 k-tiles each: at ~98 us apiece for 50 us of
// k-tiles, mostly ramp-up, a 256 KiB slab per item and tail.  Batched they are 16 tiles x 16 k-slices = 256 items of 100 k-tiles: one ramp, a quarter of the slab
// traffic.  Item ids run k-slice major over ALL problems' tiles (an XCD's chunk still shares one k-slice's operands); problem i's operands, extents and slab base
// replace GemmParams' per item.  n <= 1: the single problem of GemmParams.
constexpr int GD_MAXB = 4;
struct GdBatch { int n, tiles_all; const void* A[GD_MAXB]; const void* B[GD_MAXB]; float* part[GD_MAXB]; int M[GD_MAXB], N[GD_MAXB]; long lda[GD_MAXB], ldb[GD_MAXB];
                 int t0[GD_MAXB] /* first tile of problem i within a k-slice (unused: INT_MAX) */, tn[GD_MAXB] /* its tile columns */; };
// slab sums of a batched launch in one grid: problem i owns quads [q0[i], q0[i + 1]) of the flat quad index; plain fp32 output (C[m][n], ld = ldc)
struct GdBatchReduce { int n; long q0[GD_MAXB + 1]; const float* part[GD_MAXB]; float* C[GD_MAXB]; int N[GD_MAXB]; long MN[GD_MAXB], ldc[GD_MAXB]; };
// (wave-uniform index into a by-value kernel argument: a select chain — a dynamic index would make hipcc copy the struct to scratch)
template <typename T> VC_DEV T gd_sel(const T (&a)[GD_MAXB], int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3])); }
VC_KERNEL __launch_bounds__(256) void gemm_splitk_reduce4_batched_kernel(GdBatchReduce r, int nsplit) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= r.q0[r.n]) return;
    const int pb = (q >= r.q0[1]) + (r.n > 2 && q >= r.q0[2]) + (r.n > 3 && q >= r.q0[3]);       // (per lane: a block may straddle two problems)
    const long idx = (q - (pb == 0 ? r.q0[0] : (pb == 1 ? r.q0[1] : (pb == 2 ? r.q0[2] : r.q0[3])))) * 4;
    const float* part = pb == 0 ? r.part[0] : (pb == 1 ? r.part[1] : (pb == 2 ? r.part[2] : r.part[3]));
    float* C = pb == 0 ? r.C[0] : (pb == 1 ? r.C[1] : (pb == 2 ? r.C[2] : r.C[3]));
    const long MN = pb == 0 ? r.MN[0] : (pb == 1 ? r.MN[1] : (pb == 2 ? r.MN[2] : r.MN[3])), ldc = pb == 0 ? r.ldc[0] : (pb == 1 ? r.ldc[1] : (pb == 2 ? r.ldc[2] : r.ldc[3]));
    const int N = pb == 0 ? r.N[0] : (pb == 1 ? r.N[1] : (pb == 2 ? r.N[2] : r.N[3]));
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < nsplit; ++z) {           // fixed order: deterministic (the order of gemm_splitk_reduce4_kernel)
        float v[4]; quad_ld_f32(part + (long)z * MN + idx, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += v[k];
    }
    const long m = idx / N; const int n = (int)(idx - m * N);
    quad_st<float>(C + m * ldc + n, s);
}

// wait until at most n VMEM operations of this wave are outstanding (n wave-uniform; rounded DOWN to an encodable immediate)
template <int PW, int NS>
VC_DEV void gd_wait_le(int n) {
    if constexpr (PW + 2 * NS <= 63) { if (n >= PW + 2 * NS) { vc_wait_vmcnt<PW + 2 * NS>(); return; } }
    if constexpr (2 * NS <= 63) { if (n >= 2 * NS) { vc_wait_vmcnt<2 * NS>(); return; } }
    if (n >= PW + NS) vc_wait_vmcnt<PW + NS>();
    else if (n >= NS) vc_wait_vmcnt<NS>();
    else if (n >= PW) vc_wait_vmcnt<PW>();
    else vc_wait_vmcnt<0>();
}

// (Non-temporal C stores were tried to keep the write stream from evicting the B panel: 3x SLOWER — each lane's 8/16-byte
// piece of a row then reaches memory as its own partial write instead of merging in L2.  Plain stores it is.)
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
        { float dm[4]; vc_drop_mul4(p.drop, (uint32_t)((long)m * p.N + n), dm);
          for (int k = 0; k < 4; ++k) v[k] *= dm[k]; }
    }
    if (p.dact_src) {
        float s4[4];
        if constexpr (sizeof(TO) == 2) {
            s4[0] = vc_lo16_f32(side.x); s4[1] = vc_hi16_f32(side.x);
            s4[2] = vc_lo16_f32(side.y); s4[3] = vc_hi16_f32(side.y);
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

// NW = 4 (256-wide tile only): FOUR waves, one per SIMD, 128 x 128 outputs each (256 accumulator registers).  With eight waves the LDS
// port is the co-limiter of the main loop — per k-tile 192 KiB of fragment reads + 64 KiB of DMA writes against 2 048 MFMA cycles at
// 128 B/clk (ablations: profiles/r02_gemm_mainloop_ablation.txt); a 128 x 128 register tile needs 8 fragments per 16 MFMAs instead of 6
// per 8 (128 KiB of reads per k-tile), and a wave hides its own fragment latency behind 16 back-to-back MFMAs.  All four waves issue
// their quarter of every stage (no second wave on the SIMD to take turns with).
template <typename TO, bool TRA, bool TRB, int BN, bool COLW, int NW = 8, int BK = 64>
VC_KERNEL __launch_bounds__(NW * 64, 1) void gemm_dma_kernel(GemmParams p, int tiles_n, int tiles_mn, int nsplit, int total, int xn, int* claim, GdMini mn, GdBatch bt) {
    using TL = GdTile<BN, BK>;
    static_assert(NW == 8 || (NW == 4 && BN == 256 && BK == 64), "four-wave form: 256-wide tile, 64-deep stages only");
    static_assert(BK == 64 || (TRA && TRB), "32-deep stages: the row-contiguous (tr-read) layouts only");
    constexpr int NJ = TL::NJ, STAGES = TL::STAGES, STAGE_ELEMS = TL::STAGE_ELEMS, PIECES_B = TL::PIECES_B, HALF_N = BN / 2, NBIAS = TL::NBIAS;
    constexpr int GD_A_ELEMS_K = TL::A_ELEMS;
    constexpr int MI = NW == 8 ? 2 : 4, WR = MI * 32;                           // 32-row accumulator tiles / rows per wave
    // COL: column-per-lane accumulators (MFMA issued A x B; a lane owns NJ ADJACENT output columns of 16 rows).  r01's epilogue was
    // row-per-lane (swapped MFMA): every store / side-load instruction touched 32-64 different cache lines with 8-16 bytes each and
    // ran at ~7 B/clk per CU — a third of the QKV forward.  Here an instruction covers 2 rows x 32 lanes x NJ adjacent columns
    // (128-512 contiguous bytes per row).  Needs the B rows permuted on their way into LDS (gd_offsets<PERM>), which the DMA's
    // per-lane source address gives for free for a k-contiguous B; the tr-read B layouts (wgrad, dgrad through W) keep the r01 form.
    constexpr bool COL = !TRB && COLW;
    constexpr int NS_ITEM = (COL ? 32 : TL::NS) * (MI / 2);                      // epilogue stores per wave per interior item
    VC_DYN_SHARED(vc_bf16, lds);
    float* bias_lds = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(lds) + TL::RING_BYTES);
    int* const tk = reinterpret_cast<int*>(bias_lds + NBIAS * BN);        // dynamic claiming: item of sequence number s at tk[s & 7]
    const int tid = threadIdx.x, lane = tid & 63, wave = vc_uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned char* Ag = (const unsigned char*)p.A;      // operand bases / k-steps of the item the PREFETCH cursor is on (retarget moves them in a batched launch)
    const unsigned char* Bg = (const unsigned char*)p.B;
    const bool batched = bt.n > 1;

    // this workgroup's items: XCD x (= block id & 7, the hardware's round-robin) owns one contiguous chunk of the item list
    // (items ordered k-slice major, then tile_m, tile_n fastest), and its workgroups sweep that chunk interleaved — at any
    // moment one XCD's L2 serves neighbouring tiles that share A panels / the same k-slice.
    // xn > 1 (forward GEMMs with a weight matrix larger than an XCD's L2 share, e.g. the QKV projection: 3 MiB of weights against a
    // 4 MiB L2 that also streams A and C): the XCDs form an (8 / xn) x xn grid, XCD (gy, gx) owns tile rows gy and tile COLUMNS gx
    // only, so the 1 / xn of the weights it touches stays L2-resident instead of being re-fetched by every item (r01 counters: 1 075
    // MB fetched for 110 MB of operands); the price — A is read by xn XCDs — is paid out of the Infinity Cache.
    const int G = gridDim.x, b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int nbx = (G + 7 - xcd) >> 3;
    int tm0 = 0, tn0 = 0, tnc = tiles_n, tmn = tiles_mn, first, last;
    // chunk of XCD x out of n items dealt as evenly as possible (the first n % 8 XCDs take one more)
    auto xcd_chunk = [](int n, int x, int& cs, int& cn) { const int q8 = n >> 3, r8 = n & 7; cs = x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8; cn = x < r8 ? q8 + 1 : q8; };
    int csf = 0, cnf = 0, csm = 0, cnm = 0;                      // this XCD's chunks of the full tiles and of the minis (static lists, xn == 1)
    if (xn > 1) {
        const int xm = 8 / xn, gx = xcd % xn, gy = xcd / xn, tiles_m = tiles_mn / tiles_n;
        tnc = tiles_n / xn; tn0 = gx * tnc;
        const int qm = tiles_m / xm, rm = tiles_m % xm;
        tm0 = gy < rm ? gy * (qm + 1) : rm * (qm + 1) + (gy - rm) * qm;
        tmn = (gy < rm ? qm + 1 : qm) * tnc;                       // tiles of this XCD per k-slice
        first = j; last = tmn * nsplit;                          // (XCD-local ids; never with minis)
    } else {
        xcd_chunk(mn.nfull, xcd, csf, cnf); xcd_chunk(mn.nmini, xcd, csm, cnm);
        last = total;                                            // item ids are global; `last` = "no more items"
    }
    // static lists: list position li = j, j + nbx, ... -> item id
    auto resolve = [&](int li) -> int {
        if (xn > 1) return li < last ? li : last;
        if (li < cnf) return csf + li;
        if (li < cnf + cnm) return mn.nfull + csm + (li - cnf);
        return last;
    };
    first = resolve(j);
    // Dynamic item claiming (r03; `claim` = 9 zeroed ints: one ticket counter per XCD + a finish counter): the static lists above assume all
    // gridDim.x workgroups run at once — one per CU.  When other kernels hold CUs (RCCL's all-reduce under the backward), the workgroups that
    // could not start run as a second round after the others and the launch takes up to 2x.  With tickets, workgroup order does not matter:
    // whoever runs takes the XCD's next item (same sweep order, so the L2 locality of the static lists stays), late starters find the
    // counters exhausted; an XCD that runs dry steals from the next one (not with XCD column groups: their item lists are XCD-specific).
    // Tickets come from a SCALAR atomic (lgkmcnt; the DMA ring's vmcnt accounting is untouched), drawn by wave 0 while it would wait for the
    // ring anyway, one item ahead of the prefetch cursor; the last workgroup to finish re-zeroes the counters for the next launch.
    const bool dyn = claim != nullptr;
    int steal_x = xcd;                                            // XCD whose counter wave 0 draws from (moves on when that one is exhausted)
    auto claim_item = [&]() -> int {                              // wave 0 only
        if (xn > 1) { const int t = vc_wave_ticket(claim + steal_x); return t < last ? t : last; }
        for (int tries = 0; tries < 8; ++tries) {
            int f0, fn, m0, mc; xcd_chunk(mn.nfull, steal_x, f0, fn); xcd_chunk(mn.nmini, steal_x, m0, mc);
            const int t = vc_wave_ticket(claim + steal_x);       // ticket t = position in that XCD's list (its full tiles, then its minis)
            if (t < fn) return f0 + t;
            if (t < fn + mc) return mn.nfull + m0 + (t - fn);
            steal_x = (steal_x + 1) & 7;
        }
        return last;
    };
    int claimed = 0;                                              // sequence numbers whose items are in tk[] (all waves count alike; wave 0 writes)
    auto claim_next = [&]() {
        if (wave == 0) { const int it = claim_item(); if (lane == 0) tk[claimed & 7] = it; }
        ++claimed;
    };
    // Tickets are drawn as late as possible — a workgroup that hoards items at the start leaves others idle when there are about as many items as
    // workgroups: the first item here, item s + 1 at the ktile_begin in front of the prefetch cursor's LAST k-tile of item s (the prefetch that
    // follows that barrier is the one that moves on and reads tk[s + 1]).
    if (dyn) { claim_next(); vc_sync(); first = tk[0]; }
    const int nt = p.k_per_split / BK, ktiles = p.K / BK;        // k-tiles per slice (the last slice may be shorter)
    const bool use_bias = p.bias && !p.partial;
    const bool use_side = NJ == 2 && (p.residual || p.dact_src) && !p.partial;   // (the 256-wide tile has no registers for side inputs)
    // The two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) take turns issuing a WHOLE stage (12 pieces per wave):
    // the group whose turn it is stalls ~0.3 us in the address path while the other group is already on the matrix cores,
    // then runs its own MFMAs while the first group waits at the next barrier — issue time no longer adds to MFMA time.
    constexpr int NPA = TL::PIECES_A / 4, NPB = PIECES_B / 4;
    const int grp = NW == 8 ? wave >> 2 : 0, wq = wave & 3;
    auto my_turn = [&](int tn) { return NW == 4 || grp == tn; };
    // bytes one k-tile advances the (wave-uniform) operand base
    long kstepA = TRA ? (long)BK * p.lda * 2 : (long)BK * 2, kstepB = TRB ? (long)BK * p.ldb * 2 : (long)BK * 2;

    auto locate = [&](GdCursor& c) {            // integer divisions: once per item, never per k-tile
        if (c.item >= mn.nfull) {               // mini: (piece, tile column), mn.h rows from row mn.tm0 * GD_BM + piece * mn.h (single k-slice launches only)
            const int m_ = c.item - mn.nfull, pc = m_ / tnc;
            c.z = 0; c.tn = m_ - pc * tnc; c.row0 = mn.tm0 * GD_BM + pc * mn.h; c.mend = c.row0 + mn.h < p.M ? c.row0 + mn.h : p.M;
            c.ntc = ktiles < nt ? ktiles : nt;
            return;
        }
        if (batched) {                          // (k-slice, problem, tile row, tile column)
            c.z = c.item / bt.tiles_all; const int rem = c.item - c.z * bt.tiles_all;
            c.pb = (rem >= bt.t0[1]) + (rem >= bt.t0[2]) + (rem >= bt.t0[3]);
            const int r2 = rem - gd_sel(bt.t0, c.pb), tnp = gd_sel(bt.tn, c.pb), rm_ = r2 / tnp;
            c.row0 = rm_ * GD_BM; c.mend = gd_sel(bt.M, c.pb); c.tn = r2 - rm_ * tnp;
            const int rest = ktiles - c.z * nt; c.ntc = rest < nt ? rest : nt;
            return;
        }
        c.z = c.item / tmn; const int rem = c.item - c.z * tmn;
        const int rm_ = rem / tnc;
        c.row0 = (tm0 + rm_) * GD_BM; c.mend = p.M; c.tn = tn0 + rem - rm_ * tnc;
        const int rest = ktiles - c.z * nt; c.ntc = rest < nt ? rest : nt;
    };
    auto advance = [&](GdCursor& c) -> bool {   // true when the cursor moved on to a new item
        if (++c.kt < c.ntc) return false;
        c.kt = 0; ++c.seq;
        if (dyn) c.item = tk[c.seq & 7];                                  // (claimed ahead; visible since the last barrier)
        else { c.li += nbx; c.item = resolve(c.li); }
        if (c.item < last) locate(c);
        return true;
    };
    uint32_t offA[NPA], offB[NPB];
    auto retarget = [&](const GdCursor& c) {
        if (batched) {
            const long lda = gd_sel(bt.lda, c.pb), ldb = gd_sel(bt.ldb, c.pb);
            Ag = (const unsigned char*)gd_sel(bt.A, c.pb); Bg = (const unsigned char*)gd_sel(bt.B, c.pb);
            kstepA = TRA ? (long)BK * lda * 2 : (long)BK * 2; kstepB = TRB ? (long)BK * ldb * 2 : (long)BK * 2;
            gd_offsets<TRA, GD_BM, NPA>(offA, lda, c.row0, c.mend, wq * NPA, lane);
            gd_offsets<TRB, BN, NPB, COL ? NJ : 0>(offB, ldb, c.tn * BN, gd_sel(bt.N, c.pb), wq * NPB, lane);
            return;
        }
        gd_offsets<TRA, GD_BM, NPA>(offA, p.lda, c.row0, c.mend, wq * NPA, lane);
        gd_offsets<TRB, BN, NPB, COL ? NJ : 0>(offB, p.ldb, c.tn * BN, p.N, wq * NPB, lane);
    };
    auto issue = [&](const GdCursor& c, int slot) {
        const long kt_abs = (long)c.z * nt + c.kt;
        vc_bf16* st = lds + slot * STAGE_ELEMS;
        // the tile's bias values: issued AHEAD of the item's first stage, so the wait that retires that stage covers them
        if (use_bias && c.kt == 0 && wq == 0 && lane < BN / 4) vc_dma16(p.bias + c.tn * BN + lane * 4, bias_lds + (c.seq % NBIAS) * BN);
        if constexpr (GD_ABLATE & 4) return;
        gd_issue<NPA>(Ag + kt_abs * kstepA, offA, st, wq * NPA);
        gd_issue<NPB>(Bg + kt_abs * kstepB, offB, st + GD_A_ELEMS_K, wq * NPB);
    };

    GdCursor pf{first, 0, 1, 0, 0, 0, 0, 0, j, 0};
    if (first < last) { locate(pf); retarget(pf); }
    GdCursor cp = pf;
    int turn = 0;                                                // parity of the stage being consumed == the group that issued it
    for (int s0 = 0; s0 < STAGES - 1 && pf.item < last; ++s0) {
        if (dyn && pf.kt == pf.ntc - 1 && claimed < pf.seq + 2) { claim_next(); vc_sync(); }      // (single-k-tile items: the cursor moves on right here)
        if (my_turn(s0 & 1)) issue(pf, s0);
        if (advance(pf) && pf.item < last) retarget(pf);
    }

    vc_f32x16 acc[MI][NJ];
    // VMEM operations this wave issued in the last k-tile iterations (lower bounds): e1 / e2 / e3 = epilogue stores of iterations t-1 / t-2 / t-3, d1 / d2 = DMA
    // pieces of iterations t-1 / t-2; d_now = pieces issued in the current iteration (recorded by ktile_prefetch)
    int slot = 0, e1 = 0, e2 = 0, e3 = 0, d1 = 0, d2 = 0, young_cur = 0, d_now = 0;
    bool wave_active = true;                                     // this wave owns rows of the item being consumed (false: a mini tile's surplus waves)

    // One k-tile: retire stage `slot`, re-arm the slot freed by the previous k-tile, feed the matrix cores.
    // Every instruction here is paid 300+ times per launch by every wave (a wave issues one instruction per ~4 cycles, a
    // taken branch costs ~5 of those), so the body is kept to: wait, barrier, (every other k-tile) 12 DMA issues, 16 ds_read, 16 MFMA.
    auto ktile_begin = [&]() {
        // (wave 0 draws the ticket in front of the cursor's last k-tile of the item — or one k-tile earlier when that is a k-tile whose stage wave 0's
        // own group waits for below: the atomic's round trip then runs under the ring wait instead of holding the barrier up)
        if (dyn && pf.item < last && claimed < pf.seq + 2 && (pf.kt == pf.ntc - 1 || (pf.kt == pf.ntc - 2 && (NW == 4 || turn == 0)))) claim_next();
        // stage `slot` must have landed.  VMEM retires in issue order on gfx9, so "at most N outstanding" with N = the number
        // of operations issued AFTER this stage's DMA — the younger stage's pieces and the epilogue stores of the last two
        // k-tiles — is exact: neither the prefetch nor the stores are waited for.  N must never over-count (the stage itself
        // could still be in flight), so only guaranteed stores (interior tile, the C quads) are counted.
        // (a wave only ever waits for the stages its own group issued: on its turn the stage being consumed is its oldest DMA
        // and nothing younger of its own is in flight yet — the stage after next is issued below, after the barrier)
        // (3 stages: the stage was issued two k-tiles ago, after that k-tile's barrier and before its epilogue — the stores of the
        // last two epilogues are younger; 2 stages: issued one k-tile ago — only the last epilogue's stores are)
        // (general form, r05: stage t was issued in iteration t - (STAGES - 1), in front of that iteration's epilogue; everything this wave issued since is younger —
        //  the stores of iterations t - STAGES + 1 .. t - 1 and the DMA pieces of iterations t - STAGES + 2 .. t - 1.  With two or three stages the own-group pieces in
        //  that window are zero — the r01 / r02 counts; the four-stage ring of the 32-deep weight-gradient form has ONE younger own stage in flight.)
        e3 = e2; e2 = e1; e1 = young_cur; d2 = d1; d1 = d_now; young_cur = 0; d_now = 0;
        if constexpr (NW == 4) {                                 // every wave issued a quarter of the stage one k-tile ago; only the last epilogue's stores are younger
            if (e1) vc_wait_vmcnt<(NS_ITEM > 63 ? 63 : NS_ITEM)>(); else vc_wait_vmcnt<0>();     // (6-bit counter: 63 rounds DOWN, which is safe)
        } else if (grp == turn) gd_wait_le<TL::PW, NS_ITEM>(STAGES == 2 ? e1 : (STAGES == 3 ? d1 + e1 + e2 : d1 + d2 + e1 + e2 + e3));
        vc_barrier_raw();                                        // everyone's pieces landed; everyone is done reading slot-1
    };
    // (Issuing the next stage's pieces BETWEEN the MFMAs instead of in one burst after the barrier was measured: +1-3 % on the
    // k-contiguous layout, -25..-40 % on the ds_read_b64_tr_b16 layouts, whose 32 LDS reads per k-tile then collide with the
    // DMA's LDS writes.  The LDS port — 48 KiB of DMA writes at 64-85 B/clk plus 128 KiB of fragment reads at 256 B/clk per
    // 1024-cycle MFMA phase — is what this tile shape saturates first; see DESIGN.md.)
    auto ktile_prefetch = [&]() {
        if (pf.item < last) {
            // stage s is issued by group s & 1: during k-tile t (turn = t & 1) that is stage t + STAGES - 1
            if (my_turn((STAGES & 1) ? turn : turn ^ 1)) { issue(pf, slot == 0 ? STAGES - 1 : slot - 1); d_now = TL::PW; }      // the slot freed by the previous k-tile (stage t + STAGES - 1: group (t + STAGES - 1) & 1)
            if (advance(pf) && pf.item < last) retarget(pf);
        }
        turn ^= 1;
    };
    // Fragments are software-pipelined inside the k-tile: the 6 / 4 ds_reads of k-step ks + 1 are issued BEFORE the 8 / 4 MFMAs of k-step ks
    // (r02's order — reads, wait, MFMAs, per k-step — left the matrix pipe idle for one LDS round trip per k-step: the ISA had
    // `s_waitcnt lgkmcnt(0)` in front of every MFMA group; profiles/r03_gemm_fragpipe_ab.txt).
    auto load_frags = [&](const vc_bf16* a_tile, const vc_bf16* b_tile, int ks, vc_s16x8 (&af)[MI], vc_s16x8 (&bf)[NJ]) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if constexpr (GD_ABLATE & 2) { for (int e = 0; e < 8; ++e) af[i][e] = (short)(lane + i); asm volatile("" : "+v"(af[i])); }
            else af[i] = gd_frag<TRA, GD_BM>(a_tile, wm * WR + i * 32, ks, lane);
        }
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) {
            if constexpr (GD_ABLATE & 2) { for (int e = 0; e < 8; ++e) bf[jn][e] = (short)(lane + jn); asm volatile("" : "+v"(bf[jn])); }
            else bf[jn] = gd_frag<TRB, BN>(b_tile, wn * HALF_N + jn * 32, ks, lane);
        }
    };
    auto mfma_step = [&](const vc_s16x8 (&af)[MI], const vc_s16x8 (&bf)[NJ]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) {
                if constexpr (GD_ABLATE & 1) { asm volatile("" :: "v"(af[i]), "v"(bf[jn])); continue; }
                if constexpr (COL) acc[i][jn] = vc_mfma_32x32x16_bf16(af[i], bf[jn], acc[i][jn]);      // D[m][n]: lane = column
                else acc[i][jn] = vc_mfma_32x32x16_bf16(bf[jn], af[i], acc[i][jn]);                    // swapped, D[n][m]: lane = row
            }
    };
    auto ktile_mfma = [&]() {
        if (!wave_active) { slot = slot == STAGES - 1 ? 0 : slot + 1; return; }
        const vc_bf16* a_tile = lds + slot * STAGE_ELEMS;
        const vc_bf16* b_tile = a_tile + GD_A_ELEMS_K;
        vc_s16x8 af[2][MI], bf[2][NJ];
        load_frags(a_tile, b_tile, 0, af[0], bf[0]);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            if (ks + 1 < BK / 16) load_frags(a_tile, b_tile, ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
            mfma_step(af[ks & 1], bf[ks & 1]);
            // issue order of this region: one MFMA, then its share of the next k-step's LDS reads — a burst of reads from all eight waves
            // fills the LDS queue and the waves then sit in ds_read issue while the matrix pipe drains (r02 ablation: MFMA + reads = sum of both)
            if (ks + 1 < BK / 16) gd_interleave<MI * NJ, MI * (TRA ? 2 : 1) + NJ * (TRB ? 2 : 1)>();
            vc_sched_fence();
        }
        slot = slot == STAGES - 1 ? 0 : slot + 1;
    };


    // plain epilogue = scale-free, bias at most: the QKV / dgrad / split-K launches (most of the FLOPs) skip the generic code
    const bool plain = !p.partial && !p.aux && !p.act && !p.drop.key && !p.dact_src && !p.residual && p.alpha == 1.0f;
    // accumulators start at the tile's bias (acc_init) — on the 256-wide tile only: the 128-wide tile keeps r02's loop
    // (its launches are the fused residual epilogues, bound by their side-input / output streams; a straight-line "dropout + residual" epilogue
    // was tried there and gained nothing, profiles/r03_gemm_epilogue_fastpath_ab.txt) and its fast paths add the bias instead
    constexpr bool FOLD = BN == 256;
    const bool fold_bias = FOLD && use_bias && plain;

    for (; cp.item < last; ++cp.seq, cp.item = dyn ? tk[cp.seq & 7] : resolve(cp.li += nbx)) {
        if (cp.seq) locate(cp);
        const int z = cp.z, tn = cp.tn, row0 = cp.row0, mend = cp.mend;        // (row0 / mend: the item's first row and row bound — a mini tile covers fewer than GD_BM rows)
        wave_active = wm * WR < mend - row0;
        // accumulators start at the tile's bias when the epilogue is plain (then the epilogue is convert + store: no per-element add); the bias row
        // was DMA'd ahead of the item's first stage, so it is readable once that stage's wait + barrier (ktile_begin) are through
        auto acc_init = [&]() {
            if (fold_bias) {
                if constexpr (COL) {
                    float b[NJ]; gd_vec_ld_f32<NJ>(bias_lds + (cp.seq % NBIAS) * BN + wn * HALF_N + NJ * (lane & 31), b);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][jn][r] = b[jn];
                } else {
                    const float* br = bias_lds + (cp.seq % NBIAS) * BN + wn * HALF_N + 4 * (lane >> 5);
#pragma unroll
                    for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float b4[4]; quad_ld_f32(br + jn * 32 + 8 * q, b4);
#pragma unroll
                            for (int i = 0; i < MI; ++i)
#pragma unroll
                                for (int k = 0; k < 4; ++k) acc[i][jn][4 * q + k] = b4[k];
                        }
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;
        };
        if constexpr (FOLD) {
            if (cp.ntc > 1) {
                ktile_begin(); acc_init(); ktile_prefetch(); ktile_mfma();
                for (int kt = 1; kt < cp.ntc - 1; ++kt) { ktile_begin(); ktile_prefetch(); ktile_mfma(); }
            }
        } else {
            acc_init();
#ifdef VCAD_AB
#include "gemm_dma_ab.h"       // (A/B build only: the r04 spread-epilogue experiment; a fragment that uses this loop's locals)
#endif
            for (int kt = 0; kt < cp.ntc - 1; ++kt) { ktile_begin(); ktile_prefetch(); ktile_mfma(); }
        }

        // ---- last k-tile of the item: the epilogue's side input is requested before the MFMA phase that hides its latency
        ktile_begin();
        if constexpr (FOLD) if (cp.ntc == 1) acc_init();
        // ---------------------------------------------------------------- column-per-lane form (k-contiguous B)
        if constexpr (COL) {
            const int cl = lane & 31;
            const int nb = tn * BN + wn * HALF_N + NJ * cl;                        // first of this lane's NJ adjacent columns
            const int mb = row0 + wm * WR + 4 * (lane >> 5);                 // row of (i = 0, r = 0); row(i, r) = mb + 32 i + (r & 3) + 8 (r >> 2)
            vc_u32x2 sd[MI][16];                                                    // side input of (i, r): NJ = 2 columns (fp32 pair / packed bf16 pair in .x)
            if constexpr (NJ == 2) if (use_side) {
                if (p.residual) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            int m = mb + i * 32 + (r & 3) + 8 * (r >> 2); m = m < mend ? m : mend - 1;
                            sd[i][r] = *reinterpret_cast<const vc_u32x2*>(p.residual + (long)m * p.ldr + nb);
                        }
                } else if constexpr (sizeof(TO) == 2) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            int m = mb + i * 32 + (r & 3) + 8 * (r >> 2); m = m < mend ? m : mend - 1;
                            sd[i][r].x = *reinterpret_cast<const uint32_t*>(((const TO*)p.dact_src) + (long)m * p.lddact + nb);
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            int m = mb + i * 32 + (r & 3) + 8 * (r >> 2); m = m < mend ? m : mend - 1;
                            sd[i][r] = *reinterpret_cast<const vc_u32x2*>(((const TO*)p.dact_src) + (long)m * p.lddact + nb);
                        }
                }
            }
            ktile_prefetch();
            ktile_mfma();
            if (VC_ABL(64)) { if (acc[0][0][0] == 12345.678f) ((float*)p.C)[0] = acc[1][1][3] + acc[0][1][5] + acc[1][0][7]; continue; }
            float bv[NJ];
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) bv[jn] = 0.0f;
            if (use_bias) gd_vec_ld_f32<NJ>(bias_lds + (cp.seq % NBIAS) * BN + wn * HALF_N + NJ * cl, bv);
            // ---- plain / k-slice-slab epilogues (most of the FLOPs): straight-line code.  The generic loop below decides partial / plain /
            // fused, the row bound and a 64-bit row address PER ROW — ~35 instructions and 4 branches per store, ~1 100 per wave and item,
            // a quarter of the QKV forward's time (profiles/r03_gemm_epilogue_fastpath_ab.txt).  Here: the mode and "interior tile" are
            // decided once (wave-uniform), row offsets are compile-time multiples of the (scalar) leading dimension.
            if (p.partial || plain) {
                const bool interior = row0 + GD_BM <= mend;
                auto rows = [&](auto INTERIOR, auto PART) {
                    using T = typename std::conditional<decltype(PART)::value, float, TO>::type;
                    const long ld = decltype(PART)::value ? (long)p.N : p.ldc;
                    T* q0 = (decltype(PART)::value ? (T*)(p.partial + (long)z * p.M * p.N) : (T*)p.C) + (long)mb * ld + nb;
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
                            if (decltype(INTERIOR)::value || mb + ro < mend) {
                                float v[NJ];
#pragma unroll
                                for (int jn = 0; jn < NJ; ++jn) v[jn] = FOLD ? acc[i][jn][r] : acc[i][jn][r] + bv[jn];      // (256-wide: the bias is already in, acc_init; slabs: bv = 0)
                                gd_vec_st<T, NJ>(q0 + (long)ro * ld, v);
                            }
                        }
                };
                if (p.partial) { if (interior) rows(gemm_true{}, gemm_true{}); else rows(gemm_false{}, gemm_true{}); }
                else { if (interior) rows(gemm_true{}, gemm_false{}); else rows(gemm_false{}, gemm_false{}); }
                if (interior && !VC_ABL(32)) young_cur = NS_ITEM;
                continue;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (m < mend) {
                        float v[NJ];
#pragma unroll
                        for (int jn = 0; jn < NJ; ++jn) v[jn] = acc[i][jn][r];
                        if constexpr (NJ == 4) {          // (fused epilogues only from here on: the plain / slab stores returned above)
                            // 256-wide tile: the fused epilogues that need NO per-element side input (the MLP's first Linear: bias,
                            // pre-activation output, GELU, dropout) on the lane's four adjacent columns — 8 / 16-byte row pieces
                            float b4[4] = {bv[0], bv[1], bv[2], bv[3]};
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = p.alpha * v[k] + b4[k];
                            if (p.aux) gd_vec_st<TO, 4>(((TO*)p.aux) + (long)m * p.ldaux + nb, v);
                            if (p.act) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) v[k] = vc_apply_act(v[k], p.act);
                            }
                            if (p.drop.key) {
                                float dm[4]; vc_drop_mul4(p.drop, (uint32_t)((long)m * p.N + nb), dm);        // nb % 4 == 0
#pragma unroll
                                for (int k = 0; k < 4; ++k) v[k] *= dm[k];
                            }
                            gd_vec_st<TO, 4>(((TO*)p.C) + (long)m * p.ldc + nb, v);
                        } else if constexpr (NJ == 2) {
                            // fused epilogue on the lane's two adjacent columns: alpha, bias, pre-activation output, activation, dropout,
                            // activation derivative, residual (order of gemm.h: gemm_epilogue_quad)
                            v[0] = p.alpha * v[0] + bv[0]; v[1] = p.alpha * v[1] + bv[1];
                            if (p.aux) gd_vec_st<TO, 2>(((TO*)p.aux) + (long)m * p.ldaux + nb, v);
                            if (p.act) { v[0] = vc_apply_act(v[0], p.act); v[1] = vc_apply_act(v[1], p.act); }
                            if (p.drop.key) {                                      // nb is even: both columns share one hash (vc_drop_mul's pairing)
                                const uint32_t h = vc_drop_hash(p.drop, (uint32_t)(((long)m * p.N + nb) >> 1));
                                v[0] *= vc_drop_keep_lo(p.drop, h) ? p.drop.scale : 0.0f; v[1] *= vc_drop_keep_hi(p.drop, h) ? p.drop.scale : 0.0f;
                            }
                            if (p.dact_src) {
                                float s0, s1;
                                if constexpr (sizeof(TO) == 2) { s0 = vc_lo16_f32(sd[i][r].x); s1 = vc_hi16_f32(sd[i][r].x); }
                                else { s0 = vc_bits_f32(sd[i][r].x); s1 = vc_bits_f32(sd[i][r].y); }
                                v[0] = vc_apply_dact(v[0], s0, p.dact_kind); v[1] = vc_apply_dact(v[1], s1, p.dact_kind);
                            }
                            if (p.residual) { v[0] += vc_bits_f32(sd[i][r].x); v[1] += vc_bits_f32(sd[i][r].y); }
                            gd_vec_st<TO, 2>(((TO*)p.C) + (long)m * p.ldc + nb, v);
                        }
                    }
                }
            if (row0 + GD_BM <= mend && !VC_ABL(32)) young_cur = NS_ITEM;
            continue;
        }
        // ---------------------------------------------------------------- row-per-lane form (tr-read B layouts), as in r01
        vc_u32x4 side[MI][2][4];
        if constexpr (NJ == 2) if (use_side) {
            // the residual / dact choice is hoisted around the whole unrolled batch (a per-load select makes hipcc branch and
            // drain around every load)
            int mrow[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) { const int m = row0 + wm * WR + i * 32 + (lane & 31); mrow[i] = m < mend ? m : mend - 1; }
            const int ncol = tn * BN + wn * HALF_N + 4 * (lane >> 5);
            if (p.residual) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            side[i][jn][q] = *reinterpret_cast<const vc_u32x4*>(p.residual + (long)mrow[i] * p.ldr + ncol + jn * 32 + 8 * q);
            } else if constexpr (sizeof(TO) == 2) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const vc_u32x2 t = *reinterpret_cast<const vc_u32x2*>(((const TO*)p.dact_src) + (long)mrow[i] * p.lddact + ncol + jn * 32 + 8 * q);
                            side[i][jn][q].x = t.x; side[i][jn][q].y = t.y;
                        }
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            side[i][jn][q] = *reinterpret_cast<const vc_u32x4*>(((const TO*)p.dact_src) + (long)mrow[i] * p.lddact + ncol + jn * 32 + 8 * q);
            }
        }
        ktile_prefetch();
        const float* brow = bias_lds + (cp.seq % NBIAS) * BN + wn * HALF_N + 4 * (lane >> 5);
        ktile_mfma();

        // ---- epilogue from registers: lane holds row m = ..+(lane&31), columns n = ..+8*q+4*(lane>>5)+{0..3} for q = 0..3
        if (VC_ABL(64)) { if (acc[0][0][0] == 12345.678f) ((float*)p.C)[0] = acc[1][1][3] + acc[0][1][5] + acc[1][0][7]; continue; }
        if (p.partial || plain) {      // straight-line plain / slab stores (see the column-per-lane form above)
            auto quads = [&](auto PART) {
                using T = typename std::conditional<decltype(PART)::value, float, TO>::type;
                const int Mz = batched ? gd_sel(bt.M, cp.pb) : p.M, Nz = batched ? gd_sel(bt.N, cp.pb) : p.N;      // (batched launches: the problem's own extents and slab base)
                float* const slab = batched ? gd_sel(bt.part, cp.pb) : p.partial;
                const long ld = decltype(PART)::value ? (long)Nz : p.ldc;
                const int m0 = row0 + wm * WR + (lane & 31), n0 = tn * BN + wn * HALF_N + 4 * (lane >> 5);
                T* q0 = (decltype(PART)::value ? (T*)(slab + (long)z * Mz * Nz) : (T*)p.C) + (long)m0 * ld + n0;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if (m0 + i * 32 < mend) {
#pragma unroll
                        for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                float v[4] = {acc[i][jn][4 * q], acc[i][jn][4 * q + 1], acc[i][jn][4 * q + 2], acc[i][jn][4 * q + 3]};
                                if constexpr (!FOLD && !decltype(PART)::value) if (use_bias) {
                                    float b4[4]; quad_ld_f32(brow + jn * 32 + 8 * q, b4);
#pragma unroll
                                    for (int k = 0; k < 4; ++k) v[k] += b4[k];
                                }
                                quad_st<T>(q0 + (long)(i * 32) * ld + jn * 32 + 8 * q, v);
                            }
                    }
                }
            };
            if (p.partial) quads(gemm_true{}); else quads(gemm_false{});
            if (row0 + GD_BM <= mend && !VC_ABL(32)) young_cur = NS_ITEM;
            continue;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = row0 + wm * WR + i * 32 + (lane & 31);
            if (m < mend) {
#pragma unroll
                for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = tn * BN + wn * HALF_N + jn * 32 + 8 * q + 4 * (lane >> 5);
                        float v[4] = {acc[i][jn][4 * q], acc[i][jn][4 * q + 1], acc[i][jn][4 * q + 2], acc[i][jn][4 * q + 3]};
                        float b4[4] = {0.f, 0.f, 0.f, 0.f};
                        if (use_bias) quad_ld_f32(brow + jn * 32 + 8 * q, b4);
                        if constexpr (NJ == 2) gd_epilogue_quad<TO>(p, m, n, v, b4, side[i][jn][q]);
                    }
            }
        }
        if (row0 + GD_BM <= mend && !VC_ABL(32)) young_cur = NS_ITEM;
    }
    vc_wait_vmcnt<0>();            // no DMA may still be writing this workgroup's LDS when it is handed to the next one
    if (dyn && wave == 0) {        // the last workgroup through re-zeroes the counters (every other one has drawn its last ticket by then)
        const int done = vc_wave_ticket(claim + 8);
        if (done == (int)gridDim.x - 1 && lane < 9) claim[lane] = 0;
    }
}
