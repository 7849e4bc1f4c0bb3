// ops_gemm.hip — GEMM dispatch: picks the template instantiation, the 16-byte-load legality flags and the
// split-K factor (small tile counts with a long reduction, e.g. every wgrad: K = number of tokens).
#include "ops.h"
#include "gemm_dma.h"      // tile constants only: the persistent kernel is instantiated in ops_gemm_dma.hip
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";
void vc_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); }
const char* vc_get_error() { return g_err; }

// ---------------------------------------------------------------------------------------------- profiler
#ifndef VC_EMU
#include <vector>
namespace { struct PRec { hipEvent_t a, b; hipStream_t s; int cat, tag; double flops, bytes; }; bool g_on = false; std::vector<PRec*> g_recs;
            double g_tag_ms[VC_NTAG], g_tag_flops[VC_NTAG], g_tag_bytes[VC_NTAG]; int g_tag_n[VC_NTAG]; }
ProfScope::ProfScope(int cat, double flops, double bytes, vc_stream_t s, int tag) : rec(nullptr) {
    if (!g_on) return;
    PRec* r = new PRec(); r->cat = cat; r->tag = tag; r->flops = flops; r->bytes = bytes; r->s = s;
    (void)hipEventCreate(&r->a); (void)hipEventCreate(&r->b); (void)hipEventRecord(r->a, s);
    rec = r;
}
ProfScope::~ProfScope() { if (rec) { PRec* r = (PRec*)rec; (void)hipEventRecord(r->b, r->s); g_recs.push_back(r); } }
extern "C" void vcad_profile_begin(void) { g_on = true; }
bool vc_profile_on() { return g_on; }
// out arrays of VC_NCAT: milliseconds, flops, bytes, launches.  Synchronises the recorded events.
extern "C" int vcad_profile_end(double* ms, double* flops, double* bytes, int* launches) {
    g_on = false;
    for (int i = 0; i < VC_NCAT; ++i) { ms[i] = 0; flops[i] = 0; bytes[i] = 0; launches[i] = 0; }
    for (int i = 0; i < VC_NTAG; ++i) { g_tag_ms[i] = 0; g_tag_flops[i] = 0; g_tag_bytes[i] = 0; g_tag_n[i] = 0; }
    for (PRec* r : g_recs) {
        (void)hipEventSynchronize(r->b);
        float t = 0.f; (void)hipEventElapsedTime(&t, r->a, r->b);
        ms[r->cat] += t; flops[r->cat] += r->flops; bytes[r->cat] += r->bytes; launches[r->cat]++;
        g_tag_ms[r->tag] += t; g_tag_flops[r->tag] += r->flops; g_tag_bytes[r->tag] += r->bytes; g_tag_n[r->tag]++;
        (void)hipEventDestroy(r->a); (void)hipEventDestroy(r->b); delete r;
    }
    g_recs.clear();
    return 0;
}
// totals of the last vcad_profile_end for one kernel family (VC_TAG_*): out = {ms, flops, bytes, launches}
extern "C" int vcad_profile_kernel(int tag, double* out) {
    if (tag < 0 || tag >= VC_NTAG) return VC_ERR_ARG;
    out[0] = g_tag_ms[tag]; out[1] = g_tag_flops[tag]; out[2] = g_tag_bytes[tag]; out[3] = g_tag_n[tag];
    return 0;
}
#else
extern "C" int vcad_profile_kernel(int, double* out) { out[0] = out[1] = out[2] = out[3] = 0; return 0; }
ProfScope::ProfScope(int, double, double, vc_stream_t, int) : rec(nullptr) {}
ProfScope::~ProfScope() {}
extern "C" void vcad_profile_begin(void) {}
bool vc_profile_on() { return false; }
extern "C" int vcad_profile_end(double*, double*, double*, int*) { return 0; }
#endif

// the kernel instantiations live in their own translation units (one per family; gemm_launch.h)
int vc_gemm_launch_f32(GemmCall c, int nsplit, int lay, vc_stream_t s);
int vc_gemm_launch_x3(GemmCall c, int nsplit, int lay, vc_stream_t s);
int vc_gemm_launch_bf16_wgrad(GemmCall c, int nsplit, vc_stream_t s);              // tra = trb = 1, fp32 output, either source fp32 or bf16
int vc_gemm_launch_bf16(GemmCall c, int nsplit, int lay, vc_stream_t s);            // lay 0 / 1, B bf16

static size_t dsize(int t) { return t == VC_BF16 ? 2 : 4; }

// validation + the per-problem flags every kernel expects (activation remap, 16-byte legality of each operand)
int vc_gemm_prepare(GemmCall& c) {
    GemmParams& p = c.p;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) { vc_set_error("vc_gemm: empty problem %d %d %d", p.M, p.N, p.K); return VC_ERR_ARG; }
    if ((c.sa == VC_PK || c.sb == VC_PK || c.to == VC_PK) && c.ct != VC_X3) { vc_set_error("vc_gemm: pre-split operands / outputs belong to the bf16x3 GEMM only"); return VC_ERR_UNSUPPORTED; }
    { auto f32ish = [](int t) { return t == VC_F32 || t == VC_PK; };
      if ((c.ct == VC_F32 || c.ct == VC_X3) && !(f32ish(c.sa) && f32ish(c.sb) && f32ish(c.to))) { vc_set_error("vc_gemm: f32 / bf16x3 compute needs f32 (or pre-split) operands"); return VC_ERR_UNSUPPORTED; } }
    p.debug_skip = VC_AB(skip, 0);
    // bf16 mode: cheap erf (gemm.h) in both GEMM kernels, so results do not depend on the kernel choice.  bf16x3 (r04): the same — its absolute error
    // (<= 1.5e-7) sits a decade under the mode's own ~5e-6 per GEMM, and libm's erff made the two GELU epilogues of a ViT layer the slowest
    // launches per FLOP of the mode (302 / 357 us for 54 GFLOP, profiles/r04_x3_kernel_shapes.txt).  The exact-fp32 parity mode keeps erff.
    if (c.ct == VC_BF16 || c.ct == VC_X3) {
        if (p.act == VC_ACT_GELU) p.act = VC_ACT_GELU_FAST;
        if (p.dact_kind == VC_ACT_GELU) p.dact_kind = VC_ACT_GELU_FAST;
    }
    p.stagger = VC_AB(stagger, 0);
    p.vecA = (((uintptr_t)p.A) % 16 == 0) && ((p.lda * dsize(c.sa)) % 16 == 0);
    p.vecB = (((uintptr_t)p.B) % 16 == 0) && ((p.ldb * dsize(c.sb)) % 16 == 0);
    {   // row-wise vector epilogue: every tensor it touches must allow aligned 4-column accesses
        auto ok = [](const void* q, long ld, size_t es) { return !q || (((uintptr_t)q % 16 == 0) && ((ld * es) % 16 == 0)); };
        const size_t eo = dsize(c.to);
        p.vecC = (p.N % 4 == 0) && ok(p.C, p.ldc, eo) && ok(p.residual, p.ldr, 4) && ok(p.aux, p.ldaux, eo) &&
                 ok(p.dact_src, p.lddact, eo) && ok(p.rowadd, p.ld_rowadd, 4) && ok(p.bias, 4, 4);
        if (eo == 2) p.vecC = p.vecC && ((uintptr_t)p.C % 8 == 0);
    }
    if (VC_AB(skip, 0) & 16) p.vecC = 0;          // ablation: register-direct epilogue
    return VC_OK;
}

int vc_gemm(GemmCall c, float* scratch, size_t scratch_bytes, vc_stream_t s) {
    { int rc = vc_gemm_prepare(c); if (rc) return rc; }
    GemmParams& p = c.p;
    const unsigned fl = c.flags | VC_AB(gemm_flags, 0u);
    const int g_dma_mode = (fl & VC_GF_DMA_NEVER) ? 0 : ((fl & VC_GF_DMA_ALWAYS) ? 1 : -1);         // -1 = automatic, 0 = never, 1 = whenever legal
    const int g_dma_wide = (fl & VC_GF_WIDE_NEVER) ? 0 : ((fl & VC_GF_WIDE_ALWAYS) ? 1 : -1);
    const int g_mid_mode = (fl & VC_GF_MID_NEVER) ? 0 : ((fl & VC_GF_MID_ALWAYS) ? 1 : -1);
    const int g_force_tile = (fl & VC_GF_TILE64) ? 64 : ((fl & VC_GF_TILE128) ? 128 : 0);
    const int g_policy = VC_AB(policy, 0), g_debug_skip = VC_AB(skip, 0);
    c.flags = fl;
    int tag_dummy; int* const tag = c.kernel_out ? c.kernel_out : &tag_dummy;
    const int lay = c.tra * 2 + c.trb;
    // ---- persistent DMA-fed kernel for the big all-bf16 GEMMs (every large ViT Linear: forward, dgrad and wgrad): 256 x 256 tile where
    // the epilogue is plain (bias / k-slice slabs: QKV forward, dgrads through W^T, split-K wgrads), else 256 x 128
    const bool plain_epi = !p.act && !p.dact_src && !p.aux && !p.residual && !p.drop.key && p.alpha == 1.0f;
    // ... or a fused epilogue WITHOUT per-element side inputs (bias, pre-activation output, activation, dropout: the MLP's first Linear) on
    // the forward layout, whose column-per-lane epilogue has the registers for it
    // — measured in-model (profiles/r02_gemm_policy_ab.txt): no faster than the register-staged kernel for the GELU Linear (29.73 vs 29.56 ms per
    // step), so it is taken only when the wide tile is forced (tests) or asked for (policy bit 8)
    const bool wide_fused = lay == 0 && !p.dact_src && !p.residual && !p.rowadd && (g_dma_wide == 1 || (g_policy & 8));
    for (int pass = 0; pass < 2; ++pass) {
        const int BN = pass == 0 ? 256 : GD_BN;
        if (pass == 0 && (!(plain_epi || wide_fused) || g_dma_wide == 0)) continue;
        if (!(c.ct == VC_BF16 && c.sa == VC_BF16 && c.sb == VC_BF16 && lay != 2 && g_dma_mode != 0 && !(g_debug_skip & 31) && p.vecA && p.vecB && p.vecC &&
              p.N % BN == 0 && p.K % GD_BK == 0 && (!c.tra || p.M % 8 == 0) && (lay != 3 || c.to == VC_F32) && p.M >= 8 && !p.rowadd &&
              (double)p.lda * (c.tra ? p.K : p.M) * 2 < 2.0e9 && (double)p.ldb * (c.trb ? p.K : p.N) * 2 < 2.0e9)) continue;
        if (pass == 0 && lay == 1) continue;                        // (no 256-wide instantiation of the tr-read B layout: the hot dgrads go through W^T)
        const long tiles = (long)VC_CEIL_DIV(p.M, GD_BM) * (p.N / BN);
        const int ktiles = p.K / GD_BK;
        // one k-tile of one item, microseconds.  Token-reduction wgrads (both operands streamed from HBM along k, nothing to re-use in L2) run at the
        // pace of the ring's round trips, not of the matrix cores: ~2 us per k-tile in the model (QKV wgrad: 160 k-tiles in 355 us; the 512 x 512 MLP
        // wgrads: 64 k-tiles in 127 us, profiles/r04_kernel_shapes_c2.txt).  With the r01 constants (0.42 / 0.25) the slab term outweighed the k-loop and
        // the MLP wgrads ran as 25 slices on 200 of the 256 CUs; priced at what a k-tile costs they take 32 slices = one full round (r04).
        const double kt_us = lay == 3 ? (BN == 256 ? 2.2 : 2.0) : (BN == 256 ? 0.42 : 0.25);
        // k-slices: model time as (rounds over the 256 CUs) x (k-tiles per item) + the fp32 slab round trip of a split
        int best = 1; double bestc = 1e30;
        const size_t per = (size_t)p.M * p.N * sizeof(float);
        for (int ns = 1; ns <= 64 && ns <= ktiles / 4 + 1; ++ns) {
            if (ns > 1 && (!scratch || (size_t)ns * per > scratch_bytes)) break;
            const int nt = VC_CEIL_DIV(ktiles, ns);
            if (VC_CEIL_DIV(ktiles, nt) != ns) continue;
            const double rounds = (double)VC_CEIL_DIV(tiles * ns, 256);
            const double cost = rounds * (nt * kt_us + 1.5) + (ns > 1 ? ns * (double)per * 2 / 3.0e6 + 3.0 : 0.0);   // microseconds
            if (cost < bestc) { bestc = cost; best = ns; }
        }
        const int nt = VC_CEIL_DIV(ktiles, best);
        // Where the persistent kernel wins was measured in-model at the C2 shapes (A/B of one train step, profiles/r01_gemm_ab.md):
        // it loses where its one-workgroup-per-CU design has nothing to overlap a VALU-heavy epilogue with (GELU / GELU' — the
        // register-staged kernel's second co-resident workgroup hides that), on the long-K dgrad through ds_read_b64_tr_b16
        // (K >= 2048), and (r01 / r02 only, see below) on wgrads with fewer than 16 output tiles (the k-slice slabs dominated).
        // r03: with the straight-line slab / plain epilogues the persistent kernel also wins the wgrads with fewer than 16 output tiles (the
        // MLP's 512 x 512 ones) and is level on the activation epilogues: 26.20 -> 25.90 ms per step (profiles/r03_gemm_policy_ab.txt).  A/B
        // build: policy bit 1 / 2 restore r02's two exclusions.
        const bool wins = tiles * best >= 200 && (!(g_policy & 1) || (BN == 256 && wide_fused && !p.dact_src) || (!p.act && !p.dact_src)) && !(lay == 1 && p.K >= 2048) && (!(g_policy & 2) || !(lay == 3 && tiles < 16));
        // the wide tile halves the item count: with few, short items (N = 512, K = 512: 814 items of 8 k-tiles = 3.2 rounds) the last,
        // partly filled round costs more than the tile saves (measured: profiles/r02_gemm_wide_ab.txt) — long items amortise it
        const double fill = (double)(tiles * best) / (256.0 * VC_CEIL_DIV(tiles * best, 256));
        const bool wide_ok = BN == GD_BN || g_dma_wide == 1 || lay == 0 || fill >= 0.85 || nt >= 16;   // (forward layout: the column-per-lane epilogue of the wide tile wins even then)
        if ((g_dma_mode == 1 || wins) && wide_ok) {
            p.k_per_split = nt * GD_BK;
            p.partial = best > 1 ? scratch : nullptr;
            *tag = VC_TAG_GEMM_DMA;
            return vc_gemm_dma_launch(c, best, BN, s);
        }
    }
    // ---- mid-size problems with bf16 operands (the decoder's and the CAD ViT's Linears: a few hundred tiles, operands cold in this XCD's
    // L2): one tile per workgroup behind a six-stage DMA ring (gemm_mid.h)
    if (c.ct == VC_BF16 && c.sa == VC_BF16 && c.sb == VC_BF16 && !c.tra && g_mid_mode != 0 && !(g_debug_skip & 31) && p.vecA && p.vecB && p.vecC &&
        p.K % 64 == 0 && p.N % vc_gemm_mid_tile_n(c.trb) == 0 && (double)p.lda * p.M * 2 < 2.0e9 && (double)p.ldb * (c.trb ? p.K : p.N) * 2 < 2.0e9) {
        const long mt = (long)VC_CEIL_DIV(p.M, vc_gemm_mid_tile_m(c.trb)) * (p.N / vc_gemm_mid_tile_n(c.trb));
        // automatic for the k-contiguous-B (forward) layout only: in-model A/B at C2 (profiles/r02_gemm_mid_ab.txt) forward -0.35 ms per step,
        // but the row-contiguous-B variant (decoder dgrads through W, 64 x 128 tile) +0.26 ms against the register-staged kernel
        // (r06: up to two and a half rounds of the 256 one-workgroup-per-CU slots — the decoder's 3 072-wide in-projection is 768 tiles = three rounds here, 38 us, against 33 us on
        // the register-staged 128 x 128 tile with two workgroups per CU; profiles/r06_mid_rounds_ab.txt)
        if (g_mid_mode == 1 || (!c.trb && mt >= 96 && mt <= 640 && p.K >= 256)) { *tag = VC_TAG_GEMM_MID; return vc_gemm_mid_launch(c, s); }
    }
    *tag = VC_TAG_GEMM_REG;
    const int BK = (c.ct == VC_BF16) ? GemmCfg<vc_bf16>::BK : GemmCfg<float>::BK;
    // tile size: 128x128 by default; 64x64 when that grid would leave most of the 256 CUs idle (the decoder's
    // 2048-token GEMMs): 4x the blocks and no split-K pass.  Long token reductions still split K.
    long tiles128 = (long)VC_CEIL_DIV(p.M, 128) * VC_CEIL_DIV(p.N, 128);
    // (not for the long token reductions of wgrad: halving the tile doubles operand traffic per FLOP there; those split K instead)
    // (wgrad layout: up to 512 tiles — the heads' 6000x1024 gradient over 2 080 tokens runs 256 -> 153 us on the small tile)
    // (r04, measured and not kept: taking the big tile whenever the small tile's grid spills into a second, mostly empty round of the 512 co-resident
    // slots — 752 small tiles for the decoder's 1024-wide Linears at 2 976 rows — made the T = 186 step slower: dgrad family 9.46 -> 10.10 ms,
    // profiles/r04_tile_rounds_ab.txt; the 64 x 64 tile's second round is cheaper than 192 big tiles on 256 CUs)
    const bool small = g_force_tile ? (g_force_tile == 64) : (tiles128 < (lay == 3 ? 512 : 256) && p.K <= 4096);
    const int BT = small ? 64 : 128;
    long tiles = (long)VC_CEIL_DIV(p.M, BT) * VC_CEIL_DIV(p.N, BT);
    int nsplit = 1;
    if (scratch && tiles < (VC_AB(splitk_r06, 0) ? 512 : 256) && p.K >= 8 * BK) {
        // k-slices: two workgroups are co-resident per CU (512 slots).  r01-r03 took ceil(512 / tiles) slices — for 96 tiles (the ViT's QKV weight
        // gradient in the fp32 / bf16x3 modes) that is 6 slices = 576 workgroups: a full round plus a 64-workgroup tail, i.e. two rounds of K / 6 each
        // (1 373 us per call in the bf16x3 mode, profiles/r04_x3_kernel_shapes.txt).  Now: the slice count that minimises rounds x slice length.
        const int maxs0 = p.K / (4 * BK);
        const size_t per = (size_t)p.M * p.N * sizeof(float);
        int maxs = maxs0 < 64 ? maxs0 : 64;
        if ((size_t)maxs * per > scratch_bytes) maxs = (int)(scratch_bytes / per);
        // cost in units of "one workgroup walking the whole K": rounds(ns) / ns for the k-loop + ns slab round trips (write + read of the fp32 tile
        // grid at ~5 TB/s against ~0.3 us per k-tile of a workgroup)
        // r06 experiment (A/B build: vcad_debug_splitk_r06): in the model a k-tile of this kernel lasts ~1.4 us with 16-bit operands and ~3 us with an fp32 source (decoder
        // dgrads: 16 k-tiles in 22 us; the stem's fp32-source weight gradients: 32 k-tiles in 105 us), not 0.3 us — so price it that way and let problems of up to 512
        // tiles split?  Measured SLOWER: +0.66 % on the C2 step, +0.12 % at T = 186 (profiles/r06_splitk_rule_ab.txt): the extra slab round trips and reduce launches sit
        // on the critical path, the long k-loops they shorten mostly do not (the stem's weight gradients overlap the side stream's work).  r04's rule stays.
        const double kt_us = VC_AB(splitk_r06, 0) ? ((c.sa == VC_F32 || c.sb == VC_F32) ? 3.0 : 1.4) : 0.3;
        const double t_full = (double)(p.K / BK) * kt_us, slab = (double)p.M * p.N * 8.0 / 5.0e6 / (t_full > 1e-9 ? t_full : 1e-9);
        double best = 1e30;
        for (int ns = 1; ns <= maxs; ++ns) {
            const double cost = (double)VC_CEIL_DIV(tiles * ns, 512) / ns + slab * ns;
            if (cost < best - 1e-12) { best = cost; nsplit = ns; }
        }
        if (nsplit < 1) nsplit = 1;
    }
    int kps = VC_CEIL_DIV(p.K, nsplit); kps = VC_CEIL_DIV(kps, BK) * BK;
    nsplit = VC_CEIL_DIV(p.K, kps);
    p.k_per_split = kps;
    p.partial = nsplit > 1 ? scratch : nullptr;
    c.p.k_per_split = small ? -kps : kps;          // sign carries the tile-size choice to gemm_launch (restored there)

    // column groups of the register-staged kernel (gemm.h n_group): only where B overflows an XCD's L2 and re-reading A per group is the
    // smaller evil (the fp32 / bf16x3 QKV forward: A 0.21 GB x 3 groups against B 6.3 MB x 800 tile rows)
    c.p.n_group = 0;
    if (lay <= 1 && nsplit == 1) {
        const double esb = (double)dsize(c.sb), esa = (double)dsize(c.sa);
        const double b_bytes = (double)p.N * p.K * esb, a_bytes = (double)p.M * p.K * esa;
        const int bt = small ? 64 : 128, nx = VC_CEIL_DIV(p.N, bt), ny = VC_CEIL_DIV(p.M, bt);
        if (b_bytes > 3.0e6 && ny >= 16) {
            int gn = (int)(2.0e6 / ((double)bt * p.K * esb)); if (gn < 1) gn = 1;
            if (gn < nx && a_bytes * VC_CEIL_DIV(nx, gn) < 0.5 * b_bytes * ny) c.p.n_group = gn;
        }
    }
    if ((fl >> VC_GF_NGROUP_SHIFT) & 15u) c.p.n_group = (int)((fl >> VC_GF_NGROUP_SHIFT) & 15u);        // (tests)
    if (c.ct == VC_F32) return vc_gemm_launch_f32(c, nsplit, lay, s);
    if (c.ct == VC_X3) return vc_gemm_launch_x3(c, nsplit, lay, s);
    if (lay == 3) {            // wgrad: fp32 output always; either operand may be an fp32 tensor (converted while staging)
        if (c.to != VC_F32) { vc_set_error("vc_gemm: wgrad (tra=trb=1) writes fp32"); return VC_ERR_UNSUPPORTED; }
        return vc_gemm_launch_bf16_wgrad(c, nsplit, s);
    }
    if (c.sb == VC_BF16 && lay != 2) return vc_gemm_launch_bf16(c, nsplit, lay, s);
    vc_set_error("vc_gemm: unsupported combination ct=%d sa=%d sb=%d to=%d tra=%d trb=%d", c.ct, c.sa, c.sb, c.to, c.tra, c.trb);
    return VC_ERR_UNSUPPORTED;
}
