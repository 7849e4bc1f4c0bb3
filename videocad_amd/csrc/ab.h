// ab.h — the A/B build (-DVCAD_AB, `make ab` -> tools/_bin/libvcad_ab.so, used by tools/ only).
//
// The SHIPPED library keeps no process-global switches: which kernel runs is decided per problem, and the few knobs the tests need (force a
// tile size / the persistent kernels on small problems) are plain per-call flags (GemmCall::flags, vcad_op_gemm, vcad_set_gemm_flags).
// Every other selector below picks between kernels that compute the same result and exists for the measurements in profiles/ — the slower
// variants (four-wave form of the 256-wide tile, r01 attention kernels, fused-GELU epilogues in bf16 mode) and the
// pipeline-stage ablation branches are compiled ONLY into the A/B build.  VC_AB(field, default) is a compile-time constant in the product.
#pragma once
#ifdef VCAD_AB
struct VcAb {
    int policy;        // dispatcher rules: 1 = activation epilogues stay off the persistent kernel (r02), 2 = small-tile-count wgrads too (r02), 8 = fused wide epilogue
    int stagger;       // first-wave start offset of the register-staged kernel (units of s_sleep(127))
    int skip;          // pipeline-stage ablation mask (tools/gemm_ablate*.py)
    int epilogue;      // persistent kernel, k-contiguous B: -1 automatic, 0 row-per-lane (r01), 1 column-per-lane
    int variant;       // (unused since r04: the ping-pong kernel — measured 1.6-2x slower, profiles/r02_gemm_pingpong_ab.txt — was deleted)
    int waves;         // 256-wide tile: 8 waves (64 x 128 each) or 4 (128 x 128 each; measured slower)
    int attn_variant;  // 0 current attention kernels, 1 r01 kernels
    int split_gelu;    // bf16 ViT MLP: 1 = activation as its own pass behind a plain GEMM, 0 = fused epilogue (r01)
    int no_side;       // 1 = no library side stream
    int wgrad_bk32;    // 256-wide weight-gradient instantiation: 1 = 32-deep stages, four-stage ring (r05), 0 = two 64-deep stages
    int res_in_ln;     // 16-bit ViT layers: 1 = residual add inside the LayerNorm pass behind to_out / net.4 (r05), 0 = in the GEMM epilogue (r04)
    int cls_path;      // 16-bit engines, last ViT layer: 1 = class-token attention on (q W_k, normalised tokens) (r06, attn_cls.h), 0 = K / V projections of all tokens (r05)
    int frame_first;   // whole backward with the side stream forked: 1 = frame tower's upper stage enqueued before the CAD tower's stage (r06), 0 = after (r05)
    int pe_fold;       // 16-bit engines: 1 = patch-embedding LayerNorm affine folded into its Linear (r06), 0 = applied to the patches, dgrad + LayerNorm backward for its gradients (r05)
    int dec_h16;       // 16-bit engines: 1 = decoder LayerNorms also emit 16-bit copies for the Linears / deferred weight gradients behind them (r06), 0 = those read the fp32 stream (r05)
    int splitk_r06;    // register-staged kernel: 1 = r06 experiment (k-tile priced at 1.4 / 3 us, up to 512 tiles: slower), 0 = r04's rule (default)
    int batch_wg;      // full ViT layers: 1 = net.4 / net.0 / to_out weight gradients in one launch of the persistent kernel (r06), 0 = three launches (r05)
    int attn_pf;       // ViT attention backward: L2 warm-up distance in frames (r06 experiment, slower; 0 = off, default)
    unsigned gemm_flags;   // OR-ed into every GemmCall::flags
};
extern VcAb g_ab;
#define VC_AB(field, dflt) (g_ab.field)
#else
#define VC_AB(field, dflt) (dflt)
#endif
