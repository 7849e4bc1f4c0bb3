// ops_gemm_dma.hip — instantiations and launch of the persistent DMA-fed GEMM (gemm_dma.h); own translation unit so that it
// compiles in parallel with the register-staged kernels of ops_gemm.hip.
#include "ops.h"
#include "gemm_dma.h"

// persistent DMA-fed kernel (gemm_dma.h): `total` work items = 256x128 tiles x k-slices
template <typename TO, bool TRA, bool TRB, int BN, bool COLW, int NW = 8, int BK = 64>
static int gemm_launch_dma(GemmCall c, int nsplit, vc_stream_t s) {
#ifndef VC_EMU
    static unsigned attr_set = 0;
    if (!(attr_set & vc_device_bit())) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_dma_kernel<TO, TRA, TRB, BN, COLW, NW, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GdTile<BN, BK>::LDS_BYTES);
        if (e != hipSuccess) { vc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return VC_ERR_LAUNCH; }
        attr_set |= vc_device_bit();
    }
#endif
    ProfScope ps(c.role ? c.role - 1 : (TRA ? VC_CAT_GEMM_WGRAD : (TRB ? VC_CAT_GEMM_DGRAD : VC_CAT_GEMM_FWD)), 2.0 * c.p.M * c.p.N * c.p.K,
                 (double)c.p.M * c.p.K * 2 + (double)c.p.N * c.p.K * 2 + (double)c.p.M * c.p.N * sizeof(TO), s, VC_TAG_GEMM_DMA);
    const int tiles_n = c.p.N / BN, tiles_mn = VC_CEIL_DIV(c.p.M, GD_BM) * tiles_n, total = tiles_mn * nsplit;
    // one workgroup per CU (147 KiB of LDS each).  VC_GF_RESERVE (r06, data-parallel runs): leave 8 x n CUs to the kernels of other streams — a persistent workgroup
    // that finds its CU taken (RCCL's kernels during the exchange) waits in the dispatcher until another one exits, and while it waits NOTHING of the process's other
    // streams is dispatched either: with 8 of 256 CUs pinned the whole train step ran +24 % (tickets) / +31 % (static lists), not +3 % (profiles/r06_step_hog_*.txt)
    const int cus = 256 - 8 * (int)((c.flags >> VC_GF_RESERVE_SHIFT) & 15u);
    const int grid = total < cus ? total : cus;
    {
    // XCD column groups (see the kernel): only for k-contiguous forward-layout GEMMs whose weight matrix would not stay in one XCD's L2
    int xn = 1;
    const int g_xn = (int)((c.flags >> VC_GF_XCD_COLS_SHIFT) & 15u) == 1 ? 0 : ((int)((c.flags >> VC_GF_XCD_COLS_SHIFT) & 15u) ? (int)((c.flags >> VC_GF_XCD_COLS_SHIFT) & 15u) : -1);
    if (!TRA && !TRB && g_xn != 0 && grid >= 8) {
        const double b_bytes = (double)c.p.N * c.p.K * 2, a_bytes = (double)c.p.M * c.p.K * 2, slice = (double)BN * c.p.K * 2;
        const int tiles_m = tiles_mn / tiles_n;
        if (g_xn > 0) { xn = g_xn; if (tiles_n % xn || tiles_m < 8 / xn) xn = 1; }         // forced (tests, A/B): any grid
        else if (grid == cus && b_bytes >= 2.0e6) {
            xn = 2; while (xn < 8 && b_bytes / xn > 1.6e6) xn *= 2;
            // worth it only if the extra A reads (xn XCD columns) stay well below the weight re-fetches they remove, and every XCD keeps rows to sweep
            if (tiles_n % xn || a_bytes * xn > 0.5 * (double)tiles_mn * slice || tiles_m < 8 * 8 / xn) xn = 1;
        }
    }
    // Mini tiles (gemm_dma.h GdMini): when the item count leaves a mostly empty last round — 800 tiles on 256 CUs: 32 workgroups walk a fourth tile while 224 idle —
    // the rows of that round are cut into 64-row pieces of the same tile program instead, one per workgroup at about a third of a full tile's time.  Single k-slice,
    // row-major A (forward / dgrad layouts), no XCD column groups.  VC_GF_MINI_ALWAYS (tests): the last full tile row and everything behind it, whatever the grid.
    GdMini mn = {0, 0, total, 0};
    const int mini_mode = (c.flags & VC_GF_MINI_NEVER) ? 0 : ((c.flags & VC_GF_MINI_ALWAYS) ? 1 : -1);
    if (!TRA && nsplit == 1 && xn == 1 && mini_mode != 0) {
        const int tiles_m = tiles_mn / tiles_n, MH = 64;
        int tm0 = -1;
        if (mini_mode == 1) tm0 = c.p.M / GD_BM > 0 ? c.p.M / GD_BM - 1 : 0;
        else if (grid == cus && tiles_mn > cus) {
            const int full_rounds = tiles_mn / cus, cand = (int)((long)full_rounds * cus / tiles_n);       // tile rows that fill whole rounds
            const int nm = VC_CEIL_DIV(c.p.M - cand * GD_BM, MH) * tiles_n;
            // cost in rounds: a mini ~ 0.35 of a full tile (its B tile, the k-loop of one wave pair)
            const double plain = (double)VC_CEIL_DIV(tiles_mn, cus), with = (double)(cand * tiles_n) / cus + 0.35 * VC_CEIL_DIV(nm, cus);
            if (cand >= 1 && cand < tiles_m && with < plain - 0.3) tm0 = cand;
        }
        if (tm0 >= 0 && tm0 < tiles_m) { mn.tm0 = tm0; mn.h = MH; mn.nfull = tm0 * tiles_n; mn.nmini = VC_CEIL_DIV(c.p.M - tm0 * GD_BM, MH) * tiles_n; }
    }
    const int total_items = mn.nfull + mn.nmini;
    VC_LAUNCH((gemm_dma_kernel<TO, TRA, TRB, BN, COLW, NW, BK>), dim3(total_items < cus ? total_items : cus), dim3(NW * 64), (GdTile<BN, BK>::LDS_BYTES), s, c.p, tiles_n, tiles_mn, nsplit, total_items, xn, c.claim, mn, GdBatch());
    }
    if (nsplit > 1) {
        long tot = (long)c.p.M * c.p.N;
        if (c.p.vecC && c.p.N % 4 == 0) VC_LAUNCH((gemm_splitk_reduce4_kernel<TO>), dim3((unsigned)VC_CEIL_DIV(tot / 4, 256)), dim3(256), 0, s, c.p, nsplit);
        else VC_LAUNCH((gemm_splitk_reduce_kernel<TO>), dim3((unsigned)VC_CEIL_DIV(tot, 256)), dim3(256), 0, s, c.p, nsplit);
    }
    return VC_OK;
}


// Several weight gradients dW_i[N_i, K_i] = dY_i[tok, N_i]^T X_i[tok, K_i] over the SAME tok rows in one launch of the persistent kernel (gemm_dma.h GdBatch) + one slab sum
// per problem.  calls[i]: tra = trb = 1, 16-bit operands, fp32 output; p.M = N_i (multiple of 8), p.N = K_i (multiple of 256), p.K = tok (multiple of 64, equal for all).
int vc_gemm_dma_wgrad_batched(GemmCall* calls, int n, float* scratch, size_t scratch_bytes, vc_stream_t s) {
    if (n < 1 || n > GD_MAXB) { vc_set_error("vc_gemm_dma_wgrad_batched: %d problems (1..%d)", n, GD_MAXB); return VC_ERR_ARG; }
    GdBatch bt = GdBatch();
    long tiles = 0, slab_floats = 0; double flops = 0, bytes = 0;
    for (int i = 0; i < GD_MAXB; ++i) bt.t0[i] = 0x7fffffff;
    for (int i = 0; i < n; ++i) {
        GemmCall& c = calls[i];
        { int rc = vc_gemm_prepare(c); if (rc) return rc; }
        const GemmParams& p = c.p;
        if (!(c.ct == VC_BF16 && c.sa == VC_BF16 && c.sb == VC_BF16 && c.to == VC_F32 && c.tra && c.trb && p.vecA && p.vecB && p.vecC && p.N % 256 == 0 && p.M % 8 == 0 &&
              p.K % GD_BK == 0 && p.K == calls[0].p.K && !p.bias && !p.act && !p.aux && !p.residual && !p.dact_src && !p.drop.key && !p.rowadd && p.alpha == 1.0f)) {
            vc_set_error("vc_gemm_dma_wgrad_batched: problem %d is not a plain 16-bit weight gradient of the common token count", i); return VC_ERR_UNSUPPORTED;
        }
        bt.A[i] = p.A; bt.B[i] = p.B; bt.M[i] = p.M; bt.N[i] = p.N; bt.lda[i] = p.lda; bt.ldb[i] = p.ldb; bt.t0[i] = (int)tiles; bt.tn[i] = p.N / 256;
        tiles += (long)VC_CEIL_DIV(p.M, GD_BM) * (p.N / 256); slab_floats += (long)p.M * p.N;
        flops += 2.0 * p.M * p.N * p.K; bytes += ((double)p.M + p.N) * p.K * 2 + (double)p.M * p.N * 4;
    }
    bt.n = n; bt.tiles_all = (int)tiles;
    // k-slices: one round of the 256 one-workgroup-per-CU slots, at least 8 k-tiles per item
    const int ktiles = calls[0].p.K / GD_BK;
    int ns = (int)(256 / tiles); if (ns < 1) ns = 1; if (ns > ktiles / 8) ns = ktiles / 8 > 0 ? ktiles / 8 : 1; if (ns > 64) ns = 64;
    while (ns > 1 && (size_t)ns * slab_floats * 4 > scratch_bytes) --ns;
    const int nt = VC_CEIL_DIV(ktiles, ns); ns = VC_CEIL_DIV(ktiles, nt);
    if (!scratch || (size_t)ns * slab_floats * 4 > scratch_bytes) { vc_set_error("vc_gemm_dma_wgrad_batched: slab scratch too small"); return VC_ERR_WORKSPACE; }
    { float* q = scratch; for (int i = 0; i < n; ++i) { bt.part[i] = q; q += (size_t)ns * calls[i].p.M * calls[i].p.N; } }
    GemmCall c = calls[0];
    c.p.k_per_split = nt * GD_BK; c.p.partial = scratch;
    using K = GdTile<256, 64>;
#ifndef VC_EMU
    static unsigned attr_set = 0;
    if (!(attr_set & vc_device_bit())) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_dma_kernel<float, true, true, 256, false, 8, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
        if (e != hipSuccess) { vc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return VC_ERR_LAUNCH; }
        attr_set |= vc_device_bit();
    }
#endif
    {
    ProfScope ps(VC_CAT_GEMM_WGRAD, flops, bytes, s, VC_TAG_GEMM_DMA);
    const int total = (int)tiles * ns, cus = 256 - 8 * (int)((c.flags >> VC_GF_RESERVE_SHIFT) & 15u);
    GdMini mn = {0, 0, total, 0};
    VC_LAUNCH((gemm_dma_kernel<float, true, true, 256, false, 8, 64>), dim3(total < cus ? total : cus), dim3(512), (K::LDS_BYTES), s, c.p, 1, (int)tiles, ns, total, 1, c.claim, mn, bt);
    GdBatchReduce rd = GdBatchReduce(); rd.n = n;                   // one slab-sum grid for all problems (ns = 1: the copy out of the slab)
    long q = 0;
    for (int i = 0; i < GD_MAXB + 1; ++i) rd.q0[i] = 0;
    for (int i = 0; i < n; ++i) { rd.q0[i] = q; rd.part[i] = bt.part[i]; rd.C[i] = (float*)calls[i].p.C; rd.N[i] = calls[i].p.N; rd.MN[i] = (long)calls[i].p.M * calls[i].p.N; rd.ldc[i] = calls[i].p.ldc; q += rd.MN[i] / 4; }
    for (int i = n; i < GD_MAXB + 1; ++i) rd.q0[i] = q;
    VC_LAUNCH(gemm_splitk_reduce4_batched_kernel, dim3((unsigned)VC_CEIL_DIV(q, 256)), dim3(256), 0, s, rd, ns);
    }
    return VC_OK;
}

// column-per-lane epilogue?  Interleaved A/B on the C2 shapes (profiles/r02_gemm_epilogue_ab.txt): with the 256-wide tile a lane owns 4
// adjacent columns (8-byte bf16 / 16-byte fp32 stores, full lines) and the plain epilogues gain 5-23 % (QKV forward 459 -> 370 us, dh
// dgrad 87 -> 67 us); with the 128-wide tile it owns 2 (twice the store / side-load instructions of the row form's 16-byte quads) and the
// fused residual epilogues LOSE 45-60 % (out-proj forward 196 -> 286 us) — those keep r01's row-per-lane form.
static bool use_col(const GemmCall& c, int BN) {
    if (VC_AB(epilogue, -1) >= 0) return VC_AB(epilogue, -1) != 0;
    return BN == 256;
}
int vc_gemm_dma_launch(GemmCall c, int nsplit, int BN, vc_stream_t s) {
    const int lay = c.tra * 2 + c.trb;
    // the 256-wide tile's fused epilogue (activation / pre-activation output / dropout) exists in the column-per-lane form only
    const bool fused = c.p.act || c.p.aux || c.p.drop.key || c.p.alpha != 1.0f;
    if (BN == 256 && fused) {
        if (lay != 0 || c.p.dact_src || c.p.residual) { vc_set_error("vc_gemm_dma_launch: 256-wide tile has no such fused epilogue"); return VC_ERR_UNSUPPORTED; }
        return c.to == VC_F32 ? gemm_launch_dma<float, false, false, 256, true>(c, nsplit, s) : gemm_launch_dma<vc_bf16, false, false, 256, true>(c, nsplit, s);
    }
#ifdef VCAD_AB
    if (BN == 256 && g_ab.waves == 4) {       // four-wave form of the hot instantiations
        if (lay == 3) return gemm_launch_dma<float, true, true, 256, false, 4>(c, nsplit, s);
        if (lay == 0 && use_col(c, BN)) return c.to == VC_F32 ? gemm_launch_dma<float, false, false, 256, true, 4>(c, nsplit, s) : gemm_launch_dma<vc_bf16, false, false, 256, true, 4>(c, nsplit, s);
    }
#endif
    if (BN == 256) {                       // plain epilogues only (checked by the dispatcher); no tr-read B instantiation
#ifdef VCAD_AB
        // r05 experiment (A/B build only): weight gradients on 32-deep stages in a four-stage ring — 96 KiB in flight per CU instead of 64 (gemm_dma.h: GdTile).  Measured
        // SLOWER in the model: 172 -> 211 us per launch, +0.77 ms per C2 step (profiles/r05_wgrad_bk32_ab.txt): twice the barriers and waits per byte cost more than the deeper queue buys.
        if (lay == 3 && g_ab.wgrad_bk32) return gemm_launch_dma<float, true, true, 256, false, 8, 32>(c, nsplit, s);
#endif
        if (lay == 3) return gemm_launch_dma<float, true, true, 256, false>(c, nsplit, s);
        if (lay == 0 && c.to == VC_F32) return use_col(c, BN) ? gemm_launch_dma<float, false, false, 256, true>(c, nsplit, s) : gemm_launch_dma<float, false, false, 256, false>(c, nsplit, s);
        if (lay == 0) return use_col(c, BN) ? gemm_launch_dma<vc_bf16, false, false, 256, true>(c, nsplit, s) : gemm_launch_dma<vc_bf16, false, false, 256, false>(c, nsplit, s);
        vc_set_error("vc_gemm_dma_launch: no 256-wide kernel for layout %d", lay); return VC_ERR_UNSUPPORTED;
    }
    if (lay == 3) return gemm_launch_dma<float, true, true, GD_BN, false>(c, nsplit, s);
    if (lay == 0 && c.to == VC_F32) return use_col(c, BN) ? gemm_launch_dma<float, false, false, GD_BN, true>(c, nsplit, s) : gemm_launch_dma<float, false, false, GD_BN, false>(c, nsplit, s);
    if (lay == 0) return use_col(c, BN) ? gemm_launch_dma<vc_bf16, false, false, GD_BN, true>(c, nsplit, s) : gemm_launch_dma<vc_bf16, false, false, GD_BN, false>(c, nsplit, s);
    return c.to == VC_F32 ? gemm_launch_dma<float, false, true, GD_BN, false>(c, nsplit, s) : gemm_launch_dma<vc_bf16, false, true, GD_BN, false>(c, nsplit, s);
}
