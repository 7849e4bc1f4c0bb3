// Ping-pong variant of the persistent DMA-fed bf16 GEMM (gemm_dma.h): same tile (256x128, BK = 64), same 3-stage LDS ring fed by
// global_load_lds_dwordx4 with counted vmcnt, same item stream — but the eight waves no longer move in lockstep.
//
// What r01's counters said about gemm_dma_kernel (profiles/r01_gemm_pmc.md): MFMA utilisation 0.25-0.37 with 46-70 % of the wave
// cycles parked at the per-k-tile barrier; DMA issue, fragment reads, MFMA and epilogue time ADD because every wave is in the same
// phase at the same time, and the epilogue's row-per-lane stores (64 different cache lines per instruction) hold the CU's
// address path while the next tile's DMA waits behind them.  Two changes:
//
//  1. Two wave groups one half-period apart.  Waves 0-3 and 4-7 (one of each per SIMD) alternate between a LOAD segment — the 16
//     ds_read_b128 (32 ds_read_b64_tr_b16) fragment reads of a whole k-tile into 64 registers — and a COMPUTE segment — the k-tile's
//     16 MFMAs back to back from those registers.  While one group is on the matrix pipe of its SIMD the other group's waves on
//     the same SIMDs read LDS, issue DMA, wait for DMA, or run their epilogue.  Segments are separated by raw workgroup barriers
//     (two per k-tile); group 1 runs one segment behind group 0.
//       segment 2g   : every wave issues its share (6 of 48 pieces) of stage g+2;   group 0 LOAD(g)     group 1 COMPUTE(g-1)
//       segment 2g+1 : every wave waits for its share of stage g+1 before the barrier; group 0 COMPUTE(g)  group 1 LOAD(g)
//     Slot (g+2) % 3 was last read in segment 2g-1, stage g+1 is first read in segment 2g+2: a DMA piece has >= 3.5 segments
//     (~1 800 MFMA cycles) to land, one more than in the lockstep kernel.
//  2. Line-coalesced epilogue.  The MFMA is issued A x B (not swapped): a lane holds ONE output column and 16 rows, so every store
//     instruction touches 2 rows x 32 consecutive columns (two full 128-byte lines for fp32) instead of 64 scattered 8/16-byte
//     pieces; stores go out as `global_store v_lane_offset, v_data, s[row base]` (uniform row bases: 3 SALU instructions per row).
//     The item's bias is requested one segment early by loads hipcc does not see (vc_hload_*: a compiler-visible load would be
//     answered with vmcnt(0) and drain the ring) and retired with a counted wait; a group's epilogue runs while the other computes.
// vmcnt bookkeeping: VMEM operations retire in issue order, so "my share of stage s has landed" == "at most N operations
// outstanding", N = the operations this wave issued after that share.  Each wave counts what it issues (`issued`) and remembers the
// count right after each of the last three stage issues; N is rounded DOWN to an encodable immediate.
//
// Epilogue: alpha, bias, k-slice slabs — the plain GEMMs that carry most of the FLOPs (QKV forward, every dgrad through W^T, every
// split-K wgrad).  Per-element side inputs (residual, dropout, activation / derivative) need 64 more live registers per lane and keep
// the lockstep kernel: with them this kernel spilled, and a spilled destination of a hidden load is silently wrong.
#pragma once
#include "gemm_dma.h"

constexpr int GP_NPA = GD_PIECES_A / 8, GP_NPB = GD_PIECES_B / 8;     // DMA pieces per wave per stage: 4 + 2
constexpr int GP_NP = GP_NPA + GP_NPB;
constexpr int GP_NST = 64;                                            // epilogue stores per wave per interior item (4 tiles x 16 rows)
constexpr size_t GP_LDS_BYTES = GD_RING_BYTES;

// wait until at most n of this wave's VMEM operations are outstanding (n >= 0, wave-uniform; rounded DOWN to an immediate)
VC_DEV void gp_wait_le(int n) {
    if (n >= 63) vc_wait_vmcnt<63>();
    else if (n >= 48) vc_wait_vmcnt<48>();
    else if (n >= 32) vc_wait_vmcnt<32>();
    else if (n >= 24) vc_wait_vmcnt<24>();
    else if (n >= 3 * GP_NP) vc_wait_vmcnt<3 * GP_NP>();
    else if (n >= 2 * GP_NP) vc_wait_vmcnt<2 * GP_NP>();
    else if (n >= GP_NP + 2) vc_wait_vmcnt<GP_NP + 2>();
    else if (n >= GP_NP) vc_wait_vmcnt<GP_NP>();
    else vc_wait_vmcnt<0>();
}
// the same for the two hidden bias loads of an item (the wait must name the destination registers)
VC_DEV void gp_hwait2_le(int n, uint32_t& r0, uint32_t& r1) {
    if (n >= 2 * GP_NP) vc_hwait2<2 * GP_NP>(r0, r1);
    else if (n >= GP_NP) vc_hwait2<GP_NP>(r0, r1);
    else vc_hwait2<0>(r0, r1);
}

struct GpItem { int z, tm, tn; bool interior; };

// row r of an accumulator tile sits at tile row (r & 3) + 8 * (r >> 2) (+ 4 * (lane >> 5), folded into the lane offset)
#define GP_ROW(r) (((r) & 3) + 8 * ((r) >> 2))

template <typename TO, bool TRA, bool TRB>
VC_KERNEL __launch_bounds__(GD_THREADS, 1) void gemm_pp_kernel(GemmParams p, int tiles_n, int tiles_mn, int nsplit, int total) {
    VC_DYN_SHARED(vc_bf16, lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = vc_uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    const unsigned char* Ag = (const unsigned char*)p.A;
    const unsigned char* Bg = (const unsigned char*)p.B;

    // item list of this workgroup: identical to gemm_dma_kernel (XCD-contiguous chunks, interleaved sweep)
    const int G = gridDim.x, b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int nbx = (G + 7 - xcd) >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int cs = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cn = xcd < r8 ? q8 + 1 : q8;
    const int first = cs + j, last = cs + cn;
    const int nt = p.k_per_split / GD_BK, ktiles = p.K / GD_BK;
    const bool use_bias = p.bias && !p.partial;
    const long kstepA = TRA ? (long)GD_BK * p.lda * 2 : (long)GD_BK * 2, kstepB = TRB ? (long)GD_BK * p.ldb * 2 : (long)GD_BK * 2;

    auto locate = [&](GdCursor& c) VC_INLINE_LAMBDA {
        c.z = c.item / tiles_mn; const int rem = c.item - c.z * tiles_mn;
        c.tm = rem / tiles_n; c.tn = rem - c.tm * tiles_n;
        const int rest = ktiles - c.z * nt; c.ntc = rest < nt ? rest : nt;
    };
    auto advance = [&](GdCursor& c) VC_INLINE_LAMBDA -> bool {
        if (++c.kt < c.ntc) return false;
        c.kt = 0; c.item += nbx; ++c.seq;
        if (c.item < last) locate(c);
        return true;
    };
    uint32_t offA[GP_NPA], offB[GP_NPB];
    auto retarget = [&](const GdCursor& c) VC_INLINE_LAMBDA {
        gd_offsets<TRA, GD_BM, GP_NPA>(offA, p.lda, c.tm * GD_BM, p.M, wave * GP_NPA, lane);
        gd_offsets<TRB, GD_BN, GP_NPB>(offB, p.ldb, c.tn * GD_BN, p.N, wave * GP_NPB, lane);
    };

    // ---- VMEM bookkeeping of this wave
    int issued = 0, mk0 = 0, mk1 = 0;          // mk0 / mk1: `issued` right after the newest / second-newest stage issue
    GdCursor pf{first, 0, 1, 0, 0, 0, 0};
    if (first < last) { locate(pf); retarget(pf); }
    GdCursor cp = pf;
    int pf_slot = 0;
    auto issue_stage = [&]() VC_INLINE_LAMBDA {                  // called once per stage index by every wave, also when the stream is exhausted
        if (pf.item < last) {
            const long kt_abs = (long)pf.z * nt + pf.kt;
            vc_bf16* st = lds + pf_slot * GD_STAGE_ELEMS;
            gd_issue<GP_NPA>(Ag + kt_abs * kstepA, offA, st, wave * GP_NPA);
            gd_issue<GP_NPB>(Bg + kt_abs * kstepB, offB, st + GD_A_ELEMS, wave * GP_NPB);
            issued += GP_NP;
            if (advance(pf) && pf.item < last) retarget(pf);
        }
        pf_slot = pf_slot == GD_STAGES - 1 ? 0 : pf_slot + 1;
        mk1 = mk0; mk0 = issued;
    };

    // ---- per-item state.  Everything that touches the register arrays below is written in line inside the one segment loop (no
    // lambdas: with the arrays captured by reference hipcc left the accumulators and the side registers in scratch memory).
    vc_f32x16 acc[2][2];
    vc_s16x8 fa[GD_BK / 16][2], fb[GD_BK / 16][2];
    uint32_t bias2[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;
    int side_mark = 0; bool side_hidden = false;
    GpItem ep{0, 0, 0, false}; bool ep_pending = false;
    const int nloc = wn * 64 + (lane & 31);                        // column of tile jn = 0 inside the 128-wide item
    const int mloc = wm * 64 + 4 * (lane >> 5);                    // row of (i = 0, r = 0) inside the 256-row item

    // total k-tiles of this workgroup's stream
    int Gt = 0;
    { GdCursor c{first, 0, 1, 0, 0, 0, 0};
      for (; c.item < last; c.item += nbx) { locate(c); Gt += c.ntc; } }

    // ---- prologue: stages 0 and 1 in flight, stage 0 landed
    issue_stage(); issue_stage();
    gp_wait_le(issued - mk1);
    vc_barrier_raw(); vc_sched_fence();
    // One loop over the 2 Gt + 2 segments for both groups: in segment sg a wave is in its LOAD role when (sg + grp) is even.
    //   group 0: LOAD(g) in segment 2g, COMPUTE(g) in 2g+1;   group 1: LOAD(g) in 2g+1, COMPUTE(g) in 2g+2.
    // A LOAD-role segment starts with the epilogue of the item the group finished in its previous COMPUTE segment.
    int slot = 0, nl = 0, nc = 0;                                   // ring slot of the next k-tile to load; k-tiles loaded / computed so far
    for (int sg = 0; sg <= 2 * Gt + 1; ++sg) {
        const bool even = !(sg & 1);
        if (even) issue_stage();                                    // every wave: its share of stage sg/2 + 2
        if (((sg + grp) & 1) == 0) {
            // the previous k-tile's fragments are dead from here on (its COMPUTE segment is over): free their 64 registers for the epilogue
#pragma unroll
            for (int ks = 0; ks < GD_BK / 16; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) { vc_undef(fa[ks][i]); vc_undef(fb[ks][i]); }
            // ================================================================ epilogue
            if (ep_pending) {
                // Addressing: ONE uniform 64-bit base per item (SGPR pair) + a uniform 32-bit row/column offset per store (SALU)
                // + ONE 32-bit lane offset (VGPR) -> `global_store v_off, v_data, s[base]`; no per-store 64-bit VALU arithmetic.
                const int mu = ep.tm * GD_BM + wm * 64, nu = ep.tn * GD_BN + wn * 64;       // wave-uniform origin of this wave's 64 x 64
                const int lrow = 4 * (lane >> 5), lcol = lane & 31;
                if (p.partial) {                                    // k-slice: raw fp32 slab, reduced + finished by the split-K kernel
                    const uint32_t rs = (uint32_t)vc_uniform(p.N * 4);
                    char* sb = (char*)(uintptr_t)vc_uniform64((uint64_t)(uintptr_t)(p.partial + (long)ep.z * p.M * p.N + (long)mu * p.N + nu));
                    const uint32_t voff = (uint32_t)lrow * rs + (uint32_t)lcol * 4;
                    const int mlim = ep.interior ? 0x7fffffff : p.M - mu - lrow;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                if (i * 32 + GP_ROW(r) < mlim) vc_hstore_b32(sb + (uint32_t)((i * 32 + GP_ROW(r)) * rs + jn * 32 * 4), voff, vc_f32_bits(acc[i][jn][r]));
                } else {
                    if (ep.interior && side_hidden && use_bias) gp_hwait2_le(issued - side_mark, bias2[0], bias2[1]);   // retire the hidden bias loads
                    const uint32_t crs = (uint32_t)vc_uniform((int)(p.ldc * (long)sizeof(TO)));
                    char* cb = (char*)(uintptr_t)vc_uniform64((uint64_t)(uintptr_t)((TO*)p.C + (long)mu * p.ldc + nu));
                    const uint32_t cvoff = (uint32_t)lrow * crs + (uint32_t)lcol * (uint32_t)sizeof(TO);
                    const int mlim = ep.interior ? 0x7fffffff : p.M - mu - lrow;
                    if (!ep.interior) {                             // ragged last row of items: compiler-visible side loads (drains the ring once per launch)
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn) bias2[jn] = use_bias ? vc_f32_bits(p.bias[nu + jn * 32 + lcol]) : 0u;
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn) {
                            vc_sched_fence();                      // one tile at a time
                            const float bn = use_bias ? vc_bits_f32(bias2[jn]) : 0.0f;
                            float v[16];
#pragma unroll
                            for (int r = 0; r < 16; ++r) v[r] = p.alpha * acc[i][jn][r] + bn;
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                if (i * 32 + GP_ROW(r) < mlim) {
                                    char* rb = cb + (uint32_t)((i * 32 + GP_ROW(r)) * crs + jn * 32 * (int)sizeof(TO));
                                    if constexpr (sizeof(TO) == 2) vc_hstore_b16(rb, cvoff, (uint32_t)vc_f32_to_bf16(v[r]).bits);
                                    else vc_hstore_b32(rb, cvoff, vc_f32_bits(v[r]));
                                }
                        }
                }
                if (ep.interior) issued += GP_NST;
                ep_pending = false;
            }
            // ================================================================ LOAD(nl)
            vc_sched_fence();
            if (nl < Gt) {
                if (cp.kt == cp.ntc - 1 && !p.partial && (cp.tm + 1) * GD_BM <= p.M) {
                    // last k-tile of an interior item: request the epilogue's side inputs + bias now (hidden loads; row bases =
                    // ONE 64-bit uniform base per item forced into SGPRs + 32-bit uniform offsets — a base hipcc computes with
                    // 64-bit VALU multiplies lands in VGPRs and the "s" operand of the hidden loads does not assemble)
                    const int m0 = cp.tm * GD_BM, n0 = cp.tn * GD_BN;
                    if (use_bias) {
                        const uint64_t bb = vc_uniform64((uint64_t)(uintptr_t)(p.bias + n0 + wn * 64));
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn) bias2[jn] = vc_hload_b32((const void*)(uintptr_t)(bb + (uint32_t)(jn * 32 * 4)), (uint32_t)((lane & 31) * 4));
                        issued += 2;
                    }
                    side_mark = issued; side_hidden = true;
                }
                vc_sched_fence();
                const vc_bf16* a_tile = lds + slot * GD_STAGE_ELEMS;
                const vc_bf16* b_tile = a_tile + GD_A_ELEMS;
#pragma unroll
                for (int ks = 0; ks < GD_BK / 16; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        fa[ks][i] = gd_frag<TRA, GD_BM>(a_tile, wm * 64 + i * 32, ks, lane);
                        fb[ks][i] = gd_frag<TRB, GD_BN>(b_tile, wn * 64 + i * 32, ks, lane);
                    }
                ++nl; slot = slot == GD_STAGES - 1 ? 0 : slot + 1;
            }
        } else if (nc < nl) {
            // ================================================================ COMPUTE(nc)
            if (cp.kt == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;
            }
            vc_setprio<1>();
#pragma unroll
            for (int ks = 0; ks < GD_BK / 16; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) acc[i][jn] = vc_mfma_32x32x16_bf16(fa[ks][i], fb[ks][jn], acc[i][jn]);
            vc_setprio<0>();
            if (cp.kt == cp.ntc - 1) { ep = GpItem{cp.z, cp.tm, cp.tn, (cp.tm + 1) * GD_BM <= p.M}; ep_pending = true; }
            if (++cp.kt == cp.ntc) { cp.kt = 0; cp.item += nbx; ++cp.seq; if (cp.item < last) locate(cp); }
            ++nc;
        }
        if (!even) gp_wait_le(issued - mk1);                        // stage (sg+1)/2 must have landed before the barrier
        vc_barrier_raw(); vc_sched_fence();
    }
    vc_wait_vmcnt<0>();            // no DMA may still be writing this workgroup's LDS when it is handed to the next one
}
