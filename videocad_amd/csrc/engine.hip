// engine.hip — host-side orchestration of the train step: forward, fused loss, backward (staged by DDP
// bucket), clip + Adam.  Pure launch logic: every arithmetic op is one of the HIP kernels in this directory.
//
// Restates (reference file:line)
//   AutoRegressiveTransformer.forward .......... model/autoregressive_transformer.py:121-220
//   ViT encoder (vit-pytorch, ctor call) ....... model/trajectory_model.py:54-67, 90-100
//   nn.TransformerDecoder (post-norm, ReLU) .... model/autoregressive_transformer.py:54-62, 191-197
//   compute_loss / backward / clip / Adam ...... trainer.py:935-1063, 492-494
//
// HBM layout
//   * ONE flat fp32 parameter buffer (+ grads, Adam m, v of the same shape; + bf16 shadow in bf16 mode), tensors in
//     reverse-execution order so that each DDP bucket is a contiguous range that becomes final at the end of one
//     backward stage (bucket 0 = heads+decoder+stem is ~75 % of the bytes and is ready before the ViT backward starts).
//   * activations live in a caller-owned workspace, bump-allocated per (B, T); the fp32 residual stream and every
//     GEMM input needed by wgrad are kept (no recompute: 288 GB HBM), softmax probabilities are NOT kept (LSE only).
#include "ops.h"
#include "../../include/vcad.h"
#include <map>
#include <string>
#include <vector>
#include <string.h>
#include <stdio.h>

namespace {

struct PInfo { std::string name; long off, numel; long shape[4]; int ndim; };
struct Mat { const void* p; int dt; long ld; };
struct Epi {
    const float* bias = nullptr; int act = 0;
    const float* residual = nullptr; long ldr = 0;
    const float* rowadd = nullptr; int rdiv = 1, rmod = 0; long ldrow = 0;
    void* aux = nullptr; long ldaux = 0;
    const void* dact = nullptr; long lddact = 0; int dkind = 0;
    vc_drop drop = {0u, 0u, 1.0f};
};

struct TransposeJob { long src_off; long dst_off; int rows, cols; };   // S[src_off + r*cols + c] -> wT[dst_off + c*rows + r]
struct VitW {       // float offsets into the flat buffer
    long pos, cls, ln1w, ln1b, pew, peb, ln2w, ln2b, normw, normb;
    struct L { long anw, anb, qkv, ow, ob, fnw, fnb, w1, b1, w4, b4; long qkvT = -1, owT = -1, w1T = -1, w4T = -1; /* offsets into wT */ };
    std::vector<L> l;
};
struct DecW { long sa_w, sa_b, sa_ow, sa_ob, ca_w, ca_b, ca_ow, ca_ob, w1, b1, w2, b2, n1w, n1b, n2w, n2b, n3w, n3b; };

// Scratch and ViT-backward temporaries exist twice: lane 0 = the caller's stream (frame ViT, decoder), lane 1 = the side stream the
// CAD ViT (32 images: ~230 launch-bound kernels, 6 % of a step when serialised) runs on concurrently with the frame ViT.
struct Lane { float* scr_splitk; size_t scr_splitk_bytes; float* scr_colsum; size_t scr_colsum_bytes; float* scr_lnpart; size_t scr_lnpart_bytes;
              int* claim;          // 16 ints: ticket counters of the persistent GEMM's dynamic item claiming (gemm_dma.h), zero between launches
              float *t_dx, *t_dpe; void *t_dz, *t_dh, *t_dao, *t_dqkv, *t_dpn, *t_dum; float* t_delta; uint8_t *q8a, *q8as;
              void* t_y[2]; };     // r05, forward: 16-bit outputs of a ViT layer's to_out / net.4 Linear, added to the residual stream by the LayerNorm pass behind them

struct VitLayerActs { float* stat_a; void* h_a; void* qkv; float* lse; void* ao; float* xm; float* stat_f; void* h_f; void* z; void* g; float* xo;
                      // r05: per-layer homes of the backward's small reductions' partial rows — the two LayerNorm backwards' [blocks][dgamma | dbeta | bias gradient]
                      // and the activation-derivative pass's [blocks][b1 gradient] — so their column sums can be deferred (vcad_engine::VitColsums)
                      float *part_fn, *part_an, *part_b1; };
struct VitActs { long N; void* pn; float* pstat; float* pe; float* stat2; float* x0; std::vector<VitLayerActs> L; float* statn; void* e;
                 // r06, 16-bit engines: the patch embedding's LayerNorm(1024) affine is folded into its Linear (norm.h pe_fold_kernel): pn holds the normalised patches
                 // WITHOUT gamma / beta, pe_wf = 16-bit(W diag(gamma)), pe_bf = b + W beta (refreshed after every weight change), t_dwf = the folded Linear's weight
                 // gradient, from which one small kernel takes dW, dgamma and dbeta — no dgrad to the patches, no LayerNorm backward pass over them
                 bool pe_fold = false; void* pe_wf = nullptr; float* pe_bf = nullptr; float* t_dwf = nullptr; };
struct DecLayerActs { void* qkv_s; float* lse_s; void* ao_s; float* s1; float* st1; float* x1; void* q_c; void* kv_c; float* lse_c; void* ao_c;
                      float* s2; float* st2; float* x2; void* f1; float* s3; float* st3; float* x3;
                      // per-layer homes of the backward's dY tensors, so the layer's 7 weight gradients can be deferred (see Deferred)
                      void *g_du_ff, *g_df1, *g_du_ca, *g_dq, *g_dkv, *g_du_sa, *g_dqkv;
                      // ... and of the three LayerNorm backwards' dgamma / dbeta partial rows ([vc_ln_bwd_blocks(M)][2][H]), reduced by the same grouped column sum
                      float* ln_part[3];
                      // r06: 16-bit copies of the three post-norm outputs (second output of the LayerNorm pass): the Linears that consume them round to 16 bits
                      // while staging anyway — same values — and the deferred weight gradients read 2 instead of 4 bytes per element
                      void *x1h, *x2h, *x3h; };

}  // namespace

struct vcad_engine {
    vcad_config c;
    int dt;                       // activation / storage dtype (VC_F32 | VC_BF16)
    int ct;                       // GEMM compute type: = dt, or VC_X3 (bf16x3 on fp32 tensors: VCAD_BF16X3)
    size_t esz;
    std::vector<PInfo> plist; std::map<std::string, int> pindex; long ptotal = 0;
    std::vector<std::pair<long, long>> buckets;
    VitW wv[2]; std::vector<DecW> wd;      // [0] = state_embedding_model, [1] = cad_embedding_model
    long o_es_w, o_es_b, o_ei_w, o_ei_b, o_ip_w, o_ip_b, o_ea_w, o_ea_b, o_ts, o_h5_w, o_h5_b, o_h6_w, o_h6_b;
    float *P = nullptr, *G = nullptr, *Mm = nullptr, *Vv = nullptr; vc_bf16* S = nullptr; uint32_t* Spk = nullptr;
    char* ws = nullptr; size_t ws_bytes = 0; char* planned_ws = nullptr;
    // incremental inference (vcad_infer_begin / vcad_infer_step): per decoder layer the projected keys / values of every step so
    // far — self-attention over the tgt stream, cross-attention over the memory stream — laid out [B][Tmax][2H]
    int infer_T = 0, infer_t = 0, infer_u8 = 0; std::vector<void*> ic_kv_s, ic_kv_c;
    // ---- per-(B,T) plan
    int B = 0, T = 0; bool fwd_valid = false;
    const void* in_frames = nullptr; long in_fbstride = 0; const float* in_actions = nullptr; const void* in_cad = nullptr;
    // multiview branch (reference model/autoregressive_transformer.py:72-74,167-170): V views per clip run through the CAD tower with the CAD image
    // (one batch of B (1 + V) images, staged contiguously), embed_multiview maps their concatenated cls vectors to one more image_projection input
    const void* in_mv = nullptr; void* cadmv = nullptr; void* mvE = nullptr; float* t_dmvE = nullptr; long o_mv_w = -1, o_mv_b = -1;
    int in_u8 = 0;                // frames / cad are uint8 grayscale pixels, normalised inside the patchify kernels (vcad_forward_u8)
    VitActs va[2]; std::vector<DecLayerActs> da;
    float *ui, *cadterm, *mem, *act; void* cadE;
    void *tgt0h = nullptr, *memh = nullptr; bool dec_h16 = false;     // r06: 16-bit copies of the decoder's input and of the memory (DecLayerActs::x1h ..)
    float *xfinal;                // = da.back().x3
    // backward temporaries
    float *t_dmem, *t_dcur, *t_dui, *t_dcadterm, *t_dcadE, *t_dec, *t_des, *t_dpre;
    void *t_df1, *t_dq, *t_dkv, *t_dao_d, *t_dqkv_d;
    Lane lane[2];                 // scratch + ViT-backward temporaries per stream (see Lane)
    vcad_bucket_ready_fn bucket_cb = nullptr; void* bucket_cb_user = nullptr;      // vcad_set_bucket_callback: per-bucket hook of the whole backward
    vc_stream_t side = nullptr; vc_event_t ev_fork = nullptr, ev_fork2 = nullptr, ev_join = nullptr; bool side_ok = false, no_side = false, bwd_fork = false, bwd_side = false, side_pending = false;
    float *loss_rows; int* loss_arg; float *loss_small; int* loss_metrics; float *dl_cmds, *dl_pars; float* norm_part; float* norm_out;
    // fp16 build: the backward runs on gradients multiplied by grad_scale (a power of two: dlogits are scaled into dls_* on entry, every gradient bucket is
    // divided again — exactly — by the launch that finalises it), so that activation gradients of 1e-6..1e-8 stay inside fp16's normal range.  1 = off.
    float grad_scale = 1.0f; float *dls_cmds = nullptr, *dls_pars = nullptr;
    bool grad_scale_auto = false;      // fp16 engines until vcad_set_grad_scale names a value: the scale follows the planned batch (plan())
    // vcad_set_defer_unscale (r06, the native train step): the buckets stay scaled until the optimiser — the gradient-norm pass and Adam multiply by 1 / scale, and Adam
    // writes g / scale back (one more store stream in a pass that reads g anyway) instead of five read-modify-write passes over the gradient buffer — and vcad_loss
    // writes the scaled dlogits itself (two more passes).
    // Exact either way (a power of two).  grads_scaled_by: scale of the backward whose gradients sit in G still scaled (0 = G holds true gradients).
    bool defer_unscale = false; float grads_scaled_by = 0.0f; bool dls_valid = false;
    const float* bwd_dcmds = nullptr; const float* bwd_dpars = nullptr;
    // dropout (train mode): probability and the seed of the CURRENT forward (the backward regenerates the same masks)
    float drop_p = 0.f; uint64_t drop_seed = 0;
    // kernel-selection flags of this engine's GEMMs (VC_GF_*; tests put small batches on the kernels the C2 shapes take) and launches per kernel family
    unsigned gemm_flags = 0; long kernel_launches[VC_NTAG] = {0, 0, 0, 0, 0};
    // bf16 mode: transposed copies W^T[in][out] of the frame ViT's Linear weights, so every big dgrad is a k-contiguous
    // (ds_read_b128) GEMM instead of a ds_read_b64_tr_b16 one (measured: dqkv dgrad 569 -> 462 us); refreshed lazily after
    // every weight change (optimizer step / shadow sync / re-plan)
    vc_bf16* wT = nullptr; bool wT_fresh = false; std::vector<TransposeJob> wT_jobs;
    // r06: the frame tower's backward is two stages (DDP buckets); the upper one's last LayerNorm backward also emits the masked gradient the lower one's first
    // Linear consumes (and its bias gradient), as every layer boundary inside a stage does — the stage boundary used to cost a dropout pass + a column sum
    bool du_carry = false; Mat du_carry_mat{nullptr, 0, 0};
    bool pe_fresh[2] = {false, false};       // folded patch-embedding weights (VitActs::pe_wf / pe_bf) match the parameters
    // VCAD_FP8 forward mode (vcad_set_fp8): the four Linears of every full ViT layer run on the block-scaled fp8 matrix cores (gemm_mx8.h):
    // weights quantised from the fp32 master once per optimiser step (q8w / q8ws mirror the flat parameter layout: byte i <-> parameter i,
    // scale byte i/32), activations quantised into the lane's q8a / q8as right before each GEMM.  Backward is the bf16 path, unchanged.
    bool fp8 = false, q8_fresh = false; uint8_t* q8w = nullptr; uint8_t* q8ws = nullptr;
    // Train mode: the decoder's 56 weight gradients (each a ~25-40 us launch on 2 080 rows) and their 56 bias column sums are
    // DEFERRED to the end of the decoder backward and run as two grouped GEMM grids + one grouped column-sum — their dY inputs
    // live in per-layer buffers instead of shared temporaries.  Descriptor tables are built on first use (they hold pointers
    // into the bound gradient buffer and the planned workspace).
    struct Deferred {
        std::vector<GemmCall> calls[2]; GemmParams* d_probs[2] = {nullptr, nullptr}; int* d_tiles[2] = {nullptr, nullptr};
        int total_tiles[2] = {0, 0}; double flops[2] = {0, 0};
        std::vector<ColsumJob> cs; ColsumJob* d_cs = nullptr; float* cs_partial = nullptr; int cs_strips = 0, cs_chunks = 0;
        bool ready = false;
    } def;
    // The eight cross-attention K / V projections depend only on the decoder's memory: ONE grouped launch in front of the layer loop (8 x 16 x 16 = 2 048
    // tiles of 128 x 128 over the 256 CUs) instead of eight dependent 2 048-row launches of 256 tiles each inside it (r04)
    // r05: the ViT backward reduced every layer's LayerNorm-backward / activation-derivative partial rows with an in-line launch of 24 workgroups (~8 us each on
    // the critical path, 4 per full layer): they are now left in per-layer buffers and reduced by ONE grouped column sum at the end of the stage that
    // owns the parameters — the decoder's Deferred idea.  One job table per (tower, part of vit_backward); built on first use (device pointers inside).
    struct VitColsums { std::vector<ColsumJob> jobs; ColsumJob* d_jobs = nullptr; float* partial = nullptr; int strips = 0, chunks = 0; bool ready = false; } vcs[2][3];
    struct KvForward { std::vector<GemmCall> calls; GemmParams* d_probs = nullptr; int* d_tiles = nullptr; int total_tiles = 0; double flops = 0; bool ready = false; } kvf;
    // r06: the LAST ViT layer's attention consumes one query row per frame (pool = 'cls').  16-bit engines run it re-associated (attn_cls.h): g = q W_k and
    // c = p h per head instead of the K / V projections of all tokens, so two thirds of that layer's QKV Linear and of its dgrad / wgrad are never formed.
    // The per-head projections around the kernel (g = q W_k, out = c W_v^T, dc = dout W_v, dq = dg W_k^T) are batched launches of the DMA-ring GEMM (gemm_mid.h,
    // one problem per head); the K / V weight-gradient slices are one grouped launch whose descriptor table holds pointers into the gradient buffer and
    // the workspace and is rebuilt when either changes.  One instance per tower.
    struct ClsPath {
        bool on = false;
        void *q = nullptr, *g = nullptr, *c = nullptr, *dc = nullptr, *dg = nullptr, *dq = nullptr; float* r0 = nullptr;
        struct Tab { std::vector<GemmCall> calls; GemmParams* d_probs = nullptr; int* d_tiles = nullptr; int total_tiles = 0; double flops = 0; } wg;      // grouped weight gradients
        // K / V slices of to_qkv.weight's gradient: 2 H problems of 64 x D over N frames — 128 tiles with a 2 048-long reduction each at the benchmark shape
        // (150 us); the frames are cut into wg_split chunks, every chunk a problem of its own writing a slab [2 inner D] (the 2 H slices are contiguous in the
        // flat buffer), and one column-sum pass adds the slabs in order (deterministic)
        int wg_split = 1; float* wg_slab = nullptr;
        bool bwd_ready = false; int bwd_lane = -1;
    } cls[2];
};

namespace {

const int NB_BUCKETS = 5;      // heads + decoder | stem | CAD ViT | frame ViT upper | frame ViT lower + embed
const int CAD_STAGE = 2;       // the stage that may run on the side stream (vcad_backward_stage_side)

long add_param(vcad_engine* e, const std::string& name, std::initializer_list<long> shape) {
    PInfo p; p.name = name; p.ndim = (int)shape.size(); p.numel = 1;
    int i = 0; for (long s : shape) { p.shape[i++] = s; p.numel *= s; }
    for (; i < 4; ++i) p.shape[i] = 0;
    p.off = e->ptotal;
    e->ptotal += (p.numel + 127) / 128 * 128;         // 512-byte alignment (fp32), 256-byte for the bf16 shadow; a multiple of 128 elements keeps
                                                      // the fp8 copies' E8M0 scale rows (one byte per 32 elements) 4-byte aligned (gemm_mx8.h)
    e->pindex[name] = (int)e->plist.size();
    e->plist.push_back(p);
    return p.off;
}

void add_vit_layer(vcad_engine* e, int v, const std::string& pre, int L) {
    const vcad_config& c = e->c;
    const long D = c.vit_dim, inner = (long)c.vit_heads * c.vit_dim_head;
    VitW::L& l = e->wv[v].l[L];
    std::string a = pre + "transformer.layers." + std::to_string(L) + ".0.";
    std::string f = pre + "transformer.layers." + std::to_string(L) + ".1.net.";
    l.w4 = add_param(e, f + "4.weight", {D, c.vit_mlp}); l.b4 = add_param(e, f + "4.bias", {D});
    l.w1 = add_param(e, f + "1.weight", {c.vit_mlp, D}); l.b1 = add_param(e, f + "1.bias", {(long)c.vit_mlp});
    l.fnw = add_param(e, f + "0.weight", {D}); l.fnb = add_param(e, f + "0.bias", {D});
    l.ow = add_param(e, a + "to_out.0.weight", {D, inner}); l.ob = add_param(e, a + "to_out.0.bias", {D});
    l.qkv = add_param(e, a + "to_qkv.weight", {3 * inner, D});
    l.anw = add_param(e, a + "norm.weight", {D}); l.anb = add_param(e, a + "norm.bias", {D});
}
void add_vit_embed(vcad_engine* e, int v, const std::string& pre) {
    const vcad_config& c = e->c;
    const long D = c.vit_dim, pd = (long)c.patch_size * c.patch_size;
    const long ntok = (long)(c.image_size / c.patch_size) * (c.image_size / c.patch_size) + 1;
    VitW& w = e->wv[v];
    w.ln2w = add_param(e, pre + "to_patch_embedding.3.weight", {D}); w.ln2b = add_param(e, pre + "to_patch_embedding.3.bias", {D});
    w.pew = add_param(e, pre + "to_patch_embedding.2.weight", {D, pd}); w.peb = add_param(e, pre + "to_patch_embedding.2.bias", {D});
    w.ln1w = add_param(e, pre + "to_patch_embedding.1.weight", {pd}); w.ln1b = add_param(e, pre + "to_patch_embedding.1.bias", {pd});
    w.pos = add_param(e, pre + "pos_embedding", {1, ntok, D});
    w.cls = add_param(e, pre + "cls_token", {1, 1, D});
}

void build_params(vcad_engine* e) {
    const vcad_config& c = e->c;
    const long H = c.hidden_size, ff = c.dim_feedforward;
    // ---- bucket 0: heads, decoder L-1..0
    long b0 = e->ptotal;
    e->o_h6_w = add_param(e, "predict_action_class_0_999.weight", {(long)c.num_params * c.num_params_values, H});
    e->o_h6_b = add_param(e, "predict_action_class_0_999.bias", {(long)c.num_params * c.num_params_values});
    e->o_h5_w = add_param(e, "predict_action_class_0_4.weight", {(long)c.num_classes, H});
    e->o_h5_b = add_param(e, "predict_action_class_0_4.bias", {(long)c.num_classes});
    e->wd.resize(c.num_decoder_layers);
    for (int L = c.num_decoder_layers - 1; L >= 0; --L) {
        std::string p = "transformer_decoder.layers." + std::to_string(L) + ".";
        DecW& w = e->wd[L];
        w.n3w = add_param(e, p + "norm3.weight", {H}); w.n3b = add_param(e, p + "norm3.bias", {H});
        w.w2 = add_param(e, p + "linear2.weight", {H, ff}); w.b2 = add_param(e, p + "linear2.bias", {H});
        w.w1 = add_param(e, p + "linear1.weight", {ff, H}); w.b1 = add_param(e, p + "linear1.bias", {ff});
        w.n2w = add_param(e, p + "norm2.weight", {H}); w.n2b = add_param(e, p + "norm2.bias", {H});
        w.ca_ow = add_param(e, p + "multihead_attn.out_proj.weight", {H, H}); w.ca_ob = add_param(e, p + "multihead_attn.out_proj.bias", {H});
        w.ca_w = add_param(e, p + "multihead_attn.in_proj_weight", {3 * H, H}); w.ca_b = add_param(e, p + "multihead_attn.in_proj_bias", {3 * H});
        w.n1w = add_param(e, p + "norm1.weight", {H}); w.n1b = add_param(e, p + "norm1.bias", {H});
        w.sa_ow = add_param(e, p + "self_attn.out_proj.weight", {H, H}); w.sa_ob = add_param(e, p + "self_attn.out_proj.bias", {H});
        w.sa_w = add_param(e, p + "self_attn.in_proj_weight", {3 * H, H}); w.sa_b = add_param(e, p + "self_attn.in_proj_bias", {3 * H});
    }
    // ---- bucket 1: stem (small; split off so the heads + decoder all-reduce — 360 MB, 71 % of the bytes — starts before the stem's backward runs)
    e->buckets.push_back({b0, e->ptotal}); b0 = e->ptotal;
    e->o_ea_w = add_param(e, "embed_action.weight", {H, (long)c.act_dim}); e->o_ea_b = add_param(e, "embed_action.bias", {H});
    e->o_ts = c.enable_timestep_embedding ? add_param(e, "timestep_embedding.weight", {(long)c.max_ep_len, H}) : -1;
    // image_projection's fan-in is the reference's num_inputs = CAD + (past states) + (multiview), model/autoregressive_transformer.py:69-76
    e->o_ip_w = add_param(e, "image_projection.weight", {H, (1 + (c.enable_past_states ? 1 : 0) + (c.num_views > 0 ? 1 : 0)) * H}); e->o_ip_b = add_param(e, "image_projection.bias", {H});
    if (c.num_views > 0) { e->o_mv_w = add_param(e, "embed_multiview.weight", {H, (long)c.vit_dim * c.num_views}); e->o_mv_b = add_param(e, "embed_multiview.bias", {H}); }
    e->o_ei_w = add_param(e, "embed_image.weight", {H, (long)c.vit_dim}); e->o_ei_b = add_param(e, "embed_image.bias", {H});
    e->o_es_w = add_param(e, "embed_state.weight", {H, (long)c.vit_dim}); e->o_es_b = add_param(e, "embed_state.bias", {H});
    e->buckets.push_back({b0, e->ptotal});
    // ---- bucket 2: CAD ViT;  buckets 3,4: state ViT (upper / lower half)
    for (int v = 1; v >= 0; --v) {
        std::string pre = v == 0 ? "state_embedding_model." : "cad_embedding_model.";
        e->wv[v].l.resize(c.vit_depth);
        long bb = e->ptotal;
        e->wv[v].normw = add_param(e, pre + "transformer.norm.weight", {(long)c.vit_dim});
        e->wv[v].normb = add_param(e, pre + "transformer.norm.bias", {(long)c.vit_dim});
        const int split = c.vit_depth / 2;
        for (int L = c.vit_depth - 1; L >= 0; --L) {
            add_vit_layer(e, v, pre, L);
            if (v == 0 && L == split) { e->buckets.push_back({bb, e->ptotal}); bb = e->ptotal; }   // stage 2 = final norm + layers >= split
        }
        add_vit_embed(e, v, pre);
        e->buckets.push_back({bb, e->ptotal});
    }
}

#define g_frame_first VC_AB(frame_first, 1)   // A/B: 0 = r05's enqueue order of the whole backward (CAD tower's stage before the frame tower's)
#define g_dec_h16 VC_AB(dec_h16, 1)           // A/B: 0 = r05's decoder (in-proj / q-proj / linear1 / heads and the deferred weight gradients read the fp32 residual stream)
#define g_pe_fold VC_AB(pe_fold, 1)           // A/B: 0 = r05's patch embedding (LayerNorm affine applied to the patches; dgrad + LayerNorm backward for its parameter gradients)
#define g_cls_path VC_AB(cls_path, 1)         // A/B: 0 = r05's last ViT layer (K / V projections of all tokens + single-query attention kernels)
// ---------------------------------------------------------------------------------------------------------------
// workspace plan
// ---------------------------------------------------------------------------------------------------------------
struct Bump {
    char* base; size_t off = 0;
    template <typename U> U* take(size_t bytes) { off = (off + 255) & ~(size_t)255; U* p = (U*)(base + off); off += bytes; return p; }
};

size_t plan(vcad_engine* e, int B, int T, char* base) {
    const vcad_config& c = e->c;
    Bump b{base};
    const size_t es = e->esz;
    const long M = (long)B * T, H = c.hidden_size, D = c.vit_dim, inner = (long)c.vit_heads * c.vit_dim_head;
    const long g = c.image_size / c.patch_size, P = g * g, pd = (long)c.patch_size * c.patch_size;
    for (int v = 0; v < 2; ++v) {
        VitActs& a = e->va[v];
        a.N = v == 0 ? M : (long)B * (1 + c.num_views);
        const long R = a.N * (P + 1), Rp = a.N * P;
        a.pn = b.take<void>(Rp * pd * es); a.pstat = b.take<float>(Rp * 2 * 4); a.pe = b.take<float>(Rp * D * 4);
        a.stat2 = b.take<float>(Rp * 2 * 4); a.x0 = b.take<float>(R * D * 4);
        a.pe_fold = e->dt == VC_BF16 && e->ct == VC_BF16 && g_pe_fold; e->pe_fresh[v] = false;
        if (a.pe_fold) { a.pe_wf = b.take<void>((size_t)D * pd * es); a.pe_bf = b.take<float>((size_t)D * 4); a.t_dwf = b.take<float>((size_t)D * pd * 4); }
        a.L.resize(c.vit_depth);
        for (auto& l : a.L) {
            l.stat_a = b.take<float>(R * 2 * 4); l.h_a = b.take<void>(R * D * es); l.qkv = b.take<void>(R * 3 * inner * es);
            l.lse = b.take<float>(a.N * c.vit_heads * (P + 1) * 4); l.ao = b.take<void>(R * inner * es); l.xm = b.take<float>(R * D * 4);
            l.stat_f = b.take<float>(R * 2 * 4); l.h_f = b.take<void>(R * D * es); l.z = b.take<void>(R * c.vit_mlp * es);
            l.g = b.take<void>(R * c.vit_mlp * es); l.xo = b.take<float>(R * D * 4);
            l.part_fn = b.take<float>((size_t)vc_ln_bwd_blocks(R) * 3 * D * 4); l.part_an = b.take<float>((size_t)vc_ln_bwd_blocks(R) * 3 * D * 4);
            // (the activation-derivative pass with the b1 reduction exists on 16-bit engines only — ADVICE r05: fp32 engines planned 2 048 x vit_mlp floats per layer for nothing)
            l.part_b1 = e->dt == VC_BF16 ? b.take<float>((size_t)vc_dact_bwd_blocks(R, c.vit_mlp) * c.vit_mlp * 4) : nullptr;
        }
        for (int part = 0; part < 3; ++part) {
            vcad_engine::VitColsums& vc = e->vcs[v][part];
            vc.d_jobs = b.take<ColsumJob>((size_t)c.vit_depth * 5 * sizeof(ColsumJob));
            // pass-0 partials of the grouped kernel: per job ceil(rows / 128) x cols floats (LayerNorm jobs: <= 4 x 2 D and 4 x D; b1: <= 16 x vit_mlp)
            vc.partial = b.take<float>((size_t)c.vit_depth * (2 * 4 * 3L * D + 17L * c.vit_mlp) * 4);
            vc.ready = false;
        }
        a.statn = b.take<float>(a.N * 2 * 4); a.e = b.take<void>(a.N * D * es);
        {   // class-token attention of the last layer (vcad_engine::ClsPath)
            vcad_engine::ClsPath& cp = e->cls[v];
            cp.on = e->dt == VC_BF16 && e->ct == VC_BF16 && !e->fp8 && inner == (long)c.vit_heads * 64 && vc_cls_attn_ok(D, c.vit_heads, (int)(P + 1), c.vit_dim_head) && g_cls_path;
            cp.bwd_ready = false; cp.bwd_lane = -1;
            if (cp.on) {
                const size_t hd = (size_t)a.N * c.vit_heads * D * es;
                cp.q = b.take<void>(a.N * inner * es); cp.g = b.take<void>(hd); cp.c = b.take<void>(hd);
                cp.dc = b.take<void>(hd); cp.dg = b.take<void>(hd); cp.dq = b.take<void>(a.N * inner * es); cp.r0 = b.take<float>(a.N * D * 4);
                cp.wg_split = (int)(a.N / 256 < 1 ? 1 : (a.N / 256 > 16 ? 16 : a.N / 256));
                cp.wg_slab = cp.wg_split > 1 ? b.take<float>((size_t)cp.wg_split * 2 * inner * D * 4) : nullptr;
                const size_t np = (size_t)2 * c.vit_heads * cp.wg_split;
                cp.wg.d_probs = b.take<GemmParams>(np * sizeof(GemmParams)); cp.wg.d_tiles = b.take<int>((np + 1) * 4);
            }
        }
    }
    e->ui = b.take<float>(M * H * 4); e->cadE = b.take<void>((long)B * H * es); e->cadterm = b.take<float>((long)B * H * 4);
    if (c.num_views > 0) {
        e->cadmv = b.take<void>((long)B * (1 + c.num_views) * c.image_size * c.image_size * 4); e->mvE = b.take<void>((long)B * H * es);
        e->t_dmvE = b.take<float>((long)B * H * 4);
    }
    e->mem = b.take<float>(M * H * 4); e->act = b.take<float>(M * H * 4);
    e->da.resize(c.num_decoder_layers);
    for (auto& d : e->da) {
        d.qkv_s = b.take<void>(M * 3 * H * es); d.lse_s = b.take<float>((long)B * c.nhead * T * 4); d.ao_s = b.take<void>(M * H * es);
        d.s1 = b.take<float>(M * H * 4); d.st1 = b.take<float>(M * 2 * 4); d.x1 = b.take<float>(M * H * 4);
        d.q_c = b.take<void>(M * H * es); d.kv_c = b.take<void>(M * 2 * H * es); d.lse_c = b.take<float>((long)B * c.nhead * T * 4);
        d.ao_c = b.take<void>(M * H * es); d.s2 = b.take<float>(M * H * 4); d.st2 = b.take<float>(M * 2 * 4); d.x2 = b.take<float>(M * H * 4);
        d.f1 = b.take<void>(M * c.dim_feedforward * es); d.s3 = b.take<float>(M * H * 4); d.st3 = b.take<float>(M * 2 * 4); d.x3 = b.take<float>(M * H * 4);
        d.g_du_ff = b.take<void>(M * H * es); d.g_df1 = b.take<void>(M * c.dim_feedforward * es); d.g_du_ca = b.take<void>(M * H * es);
        d.g_dq = b.take<void>(M * H * es); d.g_dkv = b.take<void>(M * 2 * H * es); d.g_du_sa = b.take<void>(M * H * es); d.g_dqkv = b.take<void>(M * 3 * H * es);
        for (int i = 0; i < 3; ++i) d.ln_part[i] = b.take<float>((size_t)vc_ln_bwd_blocks(M) * 2 * H * 4);
        d.x1h = b.take<void>(M * H * es); d.x2h = b.take<void>(M * H * es); d.x3h = b.take<void>(M * H * es);
    }
    e->dec_h16 = e->dt == VC_BF16 && e->ct == VC_BF16 && g_dec_h16;
    e->tgt0h = b.take<void>(M * H * es); e->memh = b.take<void>(M * H * es);
    {   // deferred-wgrad descriptor tables (device) + column-sum partials
        const int nl = c.num_decoder_layers;
        for (int g = 0; g < 2; ++g) { e->def.d_probs[g] = b.take<GemmParams>((size_t)nl * 7 * sizeof(GemmParams)); e->def.d_tiles[g] = b.take<int>((size_t)(nl * 7 + 1) * 4); }
        e->def.d_cs = b.take<ColsumJob>((size_t)nl * 10 * sizeof(ColsumJob));
        e->def.cs_partial = b.take<float>((size_t)nl * (VC_CEIL_DIV(M, 128) + 1) * (7L * H + 2L * H + c.dim_feedforward + 3 * 2L * H) * 4);
        e->def.ready = false;
        e->kvf.d_probs = b.take<GemmParams>((size_t)nl * sizeof(GemmParams)); e->kvf.d_tiles = b.take<int>((size_t)(nl + 1) * 4); e->kvf.ready = false;
    }
    // backward temporaries + scratch, per lane: lane 0 (caller's stream) sized for the frame ViT / decoder, lane 1 (side stream) for the CAD ViT
    for (int ln = 0; ln < 2; ++ln) {
        Lane& l = e->lane[ln];
        const long Nv = ln == 0 ? M : (long)B * (1 + c.num_views);   // images this lane's ViT sees
        const long R = Nv * (P + 1), Rp = Nv * P;
        l.t_dx = b.take<float>(R * D * 4); l.t_dpe = b.take<float>(Rp * D * 4); l.t_dz = b.take<void>(R * c.vit_mlp * es);
        l.t_dh = b.take<void>(R * D * es); l.t_dao = b.take<void>(R * inner * es); l.t_dqkv = b.take<void>(R * 3 * inner * es);
        l.t_dpn = b.take<void>(Rp * pd * es);
        l.t_y[0] = b.take<void>(R * D * es); l.t_y[1] = b.take<void>(R * D * es);
        l.t_dum = b.take<void>((R * D > M * H ? R * D : M * H) * 4);
        const long dmax = (long)B * c.nhead * T, vmax = Nv * c.vit_heads * (P + 1);
        l.t_delta = b.take<float>((dmax > vmax ? dmax : vmax) * 4);
        l.scr_splitk_bytes = 64ul << 20; l.scr_splitk = b.take<float>(l.scr_splitk_bytes);
        l.scr_colsum_bytes = (ln == 0 ? 64ul : 16ul) << 20; l.scr_colsum = b.take<float>(l.scr_colsum_bytes);
        l.scr_lnpart_bytes = 1024ul * 2 * 1024 * 4; l.scr_lnpart = b.take<float>(l.scr_lnpart_bytes);
        l.claim = b.take<int>(64);
        l.q8a = nullptr; l.q8as = nullptr;
        if (e->fp8) { const long kmax = inner > D ? inner : D; l.q8a = b.take<uint8_t>(R * kmax); l.q8as = b.take<uint8_t>(R * kmax / 32); }
    }
    e->q8w = nullptr; e->q8ws = nullptr; e->q8_fresh = false;
    if (e->fp8) { e->q8w = b.take<uint8_t>(e->ptotal); e->q8ws = b.take<uint8_t>(e->ptotal / 32 + 64); }
    e->t_dmem = b.take<float>(M * H * 4); e->t_dcur = b.take<float>(M * H * 4); e->t_dui = b.take<float>(M * H * 4); e->t_dpre = b.take<float>(M * H * 4);
    e->t_dcadterm = b.take<float>((long)B * H * 4); e->t_dcadE = b.take<float>((long)B * H * 4);
    e->t_dec = b.take<float>((long)B * (1 + c.num_views) * D * 4); e->t_des = b.take<float>(M * D * 4);
    e->t_df1 = b.take<void>(M * c.dim_feedforward * es); e->t_dq = b.take<void>(M * H * es); e->t_dkv = b.take<void>(M * 2 * H * es);
    e->t_dao_d = b.take<void>(M * H * es); e->t_dqkv_d = b.take<void>(M * 3 * H * es);
    e->loss_rows = b.take<float>(M * 7 * 3 * 4); e->loss_arg = b.take<int>(M * 7 * 4);
    e->loss_small = b.take<float>(64 * 4); e->loss_metrics = b.take<int>(VC_NMETRIC * 4);
    const long nlog = (long)c.num_params * c.num_params_values;
    e->dl_cmds = b.take<float>(M * c.num_classes * 4); e->dl_pars = b.take<float>(M * nlog * 4);
    e->dls_cmds = e->dls_pars = nullptr;
    if (e->grad_scale_auto) {
        // the loss is a mean over M = B * T rows: 2 * 2^ceil(log2 M) undoes that factor, so the scaled dlogits — and everything behind them — have the
        // magnitudes of a one-row batch whatever the batch size (4096 at the benchmark's 2 048 rows); floor 1024 for small batches
        float a = 2.0f; while (a < 2.0f * (float)M && a < 1048576.0f) a *= 2.0f;
        e->grad_scale = a < 1024.0f ? 1024.0f : a;
    }
    if (e->grad_scale != 1.0f || e->grad_scale_auto) { e->dls_cmds = b.take<float>(M * c.num_classes * 4); e->dls_pars = b.take<float>(M * nlog * 4); }
    e->norm_part = b.take<float>(1024 * 4); e->norm_out = b.take<float>(8 * 4);
    e->wT = nullptr; e->wT_jobs.clear(); e->wT_fresh = false; e->pe_fresh[0] = e->pe_fresh[1] = false;
    if (e->dt == VC_BF16 && c.enable_past_states) {
        long off = 0;
        auto job = [&](long src, int rows, int cols) { e->wT_jobs.push_back(TransposeJob{src, off, rows, cols}); const long o = off; off += (long)rows * cols; return o; };
        for (auto& l : e->wv[0].l) {
            l.qkvT = job(l.qkv, 3 * inner, D); l.owT = job(l.ow, D, inner); l.w1T = job(l.w1, c.vit_mlp, D); l.w4T = job(l.w4, D, c.vit_mlp);
        }
        e->wT = b.take<vc_bf16>(off * 2);
    }
    return b.off + 256;
}

// ---------------------------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------------------------
// the ViT MLP's GELU / GELU' as their own passes behind plain GEMMs (bf16 mode): 1 = yes, 0 = fused into the GEMM epilogues (r01; A/B build only)
#define g_split_gelu VC_AB(split_gelu, 1)
#define g_no_side VC_AB(no_side, 0)
#define g_batch_wg VC_AB(batch_wg, 1)         // A/B: 0 = a full ViT layer's net.4 / net.0 / to_out weight gradients as three launches (r05)
#define g_res_in_ln VC_AB(res_in_ln, 1)       // A/B: 0 = r04's residual adds in the to_out / net.4 GEMM epilogues
#define CK(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
#define CK_(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

struct Ctx {
    vcad_engine* e; vc_stream_t s; int ln = 0;       // ln: which Lane's scratch / temporaries this context may touch
    const Lane& L() const { return e->lane[ln]; }
    int dt() const { return e->dt; }
    Mat W(long off, long ld) const {
        if (e->Spk) return Mat{(const void*)(e->Spk + off), VC_PK, ld};          // bf16x3: pre-split weights (gemm.h vc_pk)
        return e->dt == VC_BF16 ? Mat{(const void*)(e->S + off), VC_BF16, ld} : Mat{(const void*)(e->P + off), VC_F32, ld};
    }
    // transposed shadow (bf16 mode, frame ViT): offT < 0 -> not available
    bool hasT(long offT) const { return e->wT && offT >= 0; }
    Mat WT(long offT, long ld) const { return Mat{(const void*)(e->wT + offT), VC_BF16, ld}; }
    int refresh_wT() const {
        if (!e->wT || e->wT_fresh) return 0;
        {   // one grid for all of them (r06: 24 launches of 5-45 us sat between the stem's backward and the ViT's)
            std::vector<long> so, d_o; std::vector<int> rw, cl;
            for (const auto& j : e->wT_jobs) { so.push_back(j.src_off); d_o.push_back(j.dst_off); rw.push_back(j.rows); cl.push_back(j.cols); }
            int rc = vc_transpose_bf16_batched(e->S, e->wT, so.data(), d_o.data(), rw.data(), cl.data(), (int)so.size(), s); if (rc) return rc;
        }
        e->wT_fresh = true;
        return 0;
    }
    // dX[M,K] = dY[M,N] W[N,K] through W^T[K][N]: both operands k-contiguous
    int lin_dgrad_T(Mat dY, Mat WTm, Mat dX, int M, int N, int K, const Epi& ep) const { return gemm(dY, 0, WTm, 0, dX, M, K, N, ep, VC_CAT_GEMM_DGRAD + 1); }
    const float* Pf(long off) const { return e->P + off; }
    float* Gf(long off) const { return e->G + off; }
    Mat A32(const float* p, long ld) const { return Mat{p, VC_F32, ld}; }
    Mat AT(const void* p, long ld) const { return Mat{p, e->dt, ld}; }
    // bf16x3 mode (r04): tensors of the ViTs that only feed GEMMs / the x3 attention kernels are stored PRE-SPLIT (hi | lo words, VC_PK: same 4 bytes
    // per element as fp32) by their producers, so the consuming GEMMs unpack their tiles instead of splitting them (needs the pre-split weight shadow)
    bool pk_acts() const { return e->ct == VC_X3 && e->Spk != nullptr; }
    int vt(bool pk) const { return pk ? VC_PK : e->dt; }
    Mat VT(const void* p, long ld, bool pk) const { return Mat{p, vt(pk), ld}; }

    // dropout site ids: (module << 16) | (layer << 8) | kind; module 1 = frame ViT, 2 = CAD ViT, 3 = decoder
    enum { K_EMB = 1, K_ATTN = 2, K_OUT = 3, K_MLP_ACT = 4, K_MLP_OUT = 5, K_SA = 6, K_SA_OUT = 7, K_CA = 8, K_CA_OUT = 9, K_FF_ACT = 10, K_FF_OUT = 11 };
    vc_drop site(int module, int layer, int kind) const {
        vc_drop d = {0u, 0u, 1.0f};
        if (e->drop_p > 0.f) d = vc_drop_make(vc_drop_key(e->drop_seed, ((uint32_t)module << 16) | ((uint32_t)layer << 8) | (uint32_t)kind), e->drop_p);
        return d;
    }
    // masked copy of a residual-stream gradient: du = dx * mask (type T, compact [rows, cols]); returns the matrix to feed wgrad / dgrad
    int masked(const float* dx, long ldx, long rows, int cols, vc_drop d, Mat* out, void* dst = nullptr, bool pk = false) const {
        if (!d.key && !pk) { *out = A32(dx, ldx); return 0; }
        if (!dst) dst = L().t_dum;
        *out = VT(dst, cols, pk);                     // (pk: the pre-split copy is made even without a mask — it is what the consuming GEMMs take)
        return vc_dropout_mul(vt(pk), dx, ldx, dst, cols, rows, cols, d, s);
    }
    int gemm(Mat A, int tra, Mat B, int trb, Mat C, int M, int N, int K, const Epi& ep, int role = 0) const {
        GemmCall c; memset(&c, 0, sizeof(c));
        c.role = role; c.ct = e->ct; c.sa = A.dt; c.sb = B.dt; c.to = C.dt; c.tra = tra; c.trb = trb;
        GemmParams& p = c.p;
        p.A = A.p; p.B = B.p; p.C = (void*)C.p; p.M = M; p.N = N; p.K = K; p.lda = A.ld; p.ldb = B.ld; p.ldc = C.ld;
        p.alpha = 1.0f; p.bias = ep.bias; p.act = ep.act; p.residual = ep.residual; p.ldr = ep.ldr;
        p.rowadd = ep.rowadd; p.rowadd_div = ep.rdiv; p.rowadd_mod = ep.rmod; p.ld_rowadd = ep.ldrow;
        p.aux = ep.aux; p.ldaux = ep.ldaux; p.dact_src = ep.dact; p.lddact = ep.lddact; p.dact_kind = ep.dkind; p.drop = ep.drop;
        int tag = VC_TAG_NONE; c.flags = e->gemm_flags; c.kernel_out = &tag;
        // ticket-drawn items for the persistent kernel (gemm_dma.h): on when the engine shares the GPU with communication kernels (the data-parallel
        // trainer sets the flag; profiles/r03_gemm_hog_ab.txt: 32 occupied CUs cost 13-16 % instead of 39-56 %), off otherwise (1-3 % ticket latency)
        c.claim = (e->gemm_flags & VC_GF_DYNAMIC) ? L().claim : nullptr;
        const int rc = vc_gemm(c, L().scr_splitk, L().scr_splitk_bytes, s);
        if (!rc) ++e->kernel_launches[tag];
        return rc;
    }
    // Y[M,N] = X[M,K] W[N,K]^T (+ epilogue)
    int lin_fwd(Mat X, Mat Wm, Mat Y, int M, int N, int K, const Epi& ep) const { return gemm(X, 0, Wm, 0, Y, M, N, K, ep); }
    // the same Linear on the fp8 matrix cores (VCAD_FP8): X (compute dtype, compact [M, K]) is quantised into the lane's buffer, W comes from
    // the engine's quantised copy of the parameter tensor at w_off ([N, K], K % 128 == 0)
    int lin_fwd_q(Mat X, long w_off, Mat Y, int M, int N, int K, const Epi& ep) const {
        if (X.ld != K) { vc_set_error("lin_fwd_q: compact activations expected"); return VC_ERR_ARG; }
        CK_(vc_mx8_quant(X.dt, X.p, X.ld, L().q8a, L().q8as, M, K, s));
        Mx8Params q; memset(&q, 0, sizeof(q));
        GemmParams& p = q.g;
        p.A = L().q8a; p.B = e->q8w + w_off; p.C = (void*)Y.p; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = Y.ld; p.alpha = 1.0f;
        p.bias = ep.bias; p.act = ep.act; p.residual = ep.residual; p.ldr = ep.ldr; p.rowadd_div = 1;
        p.aux = ep.aux; p.ldaux = ep.ldaux; p.drop = ep.drop;
        q.sa = L().q8as; q.ldsa = K / 32; q.sb = e->q8ws + w_off / 32; q.ldsb = K / 32;
        return vc_gemm_mx8(q, Y.dt, s);
    }
    // quantised copies of the ViT Linear weights: rebuilt from the fp32 master whenever the parameters changed
    int refresh_q8() const {
        if (!e->fp8 || e->q8_fresh) return 0;
        const vcad_config& c = e->c; const int D = c.vit_dim, inner = c.vit_heads * c.vit_dim_head;
        for (int v = 0; v < 2; ++v)
            for (const auto& l : e->wv[v].l) {
                const long offs[4] = {l.qkv, l.ow, l.w1, l.w4}; const int rows[4] = {3 * inner, D, c.vit_mlp, D}, cols[4] = {D, inner, D, c.vit_mlp};
                for (int i = 0; i < 4; ++i) CK_(vc_mx8_quant(VC_F32, e->P + offs[i], cols[i], e->q8w + offs[i], e->q8ws + offs[i] / 32, rows[i], cols[i], s));
            }
        e->q8_fresh = true;
        return 0;
    }
    // dX[M,K] = dY[M,N] W[N,K]
    int lin_dgrad(Mat dY, Mat Wm, Mat dX, int M, int N, int K, const Epi& ep) const { return gemm(dY, 0, Wm, 1, dX, M, K, N, ep); }
    // dW[N,K] = dY[tok,N]^T X[tok,K]   (fp32, written);  db[N] = colsum(dY)
    int lin_wgrad(Mat dY, Mat X, float* dW, long lddw, float* db, int tok, int N, int K) const {
        CK(gemm(dY, 1, X, 1, Mat{dW, VC_F32, lddw}, N, K, tok, Epi()));
        if (db) CK(colsum(dY, tok, N, db, 0));
        return 0;
    }
    int colsum(Mat X, long rows, int cols, float* out, int accumulate, int batch = 1, long bsx = 0, long bso = 0) const {
        size_t need = (size_t)batch * vc_colsum_chunks(rows) * cols * 4;      // both tree levels
        if (need > L().scr_colsum_bytes) { vc_set_error("colsum scratch too small (%zu)", need); return VC_ERR_WORKSPACE; }
        return vc_colsum(X.dt, X.p, X.ld, rows, cols, out, accumulate, batch, bsx, bso, L().scr_colsum, s);
    }
    // add / sum32 (r05): x + add (the 16-bit branch output of the Linear in front) is written to sum32 as the new residual stream and normalised
    int ln_fwd(int tx, const void* x, long ldx, long wo, long bo, float* y32, long ldy32, void* yt, long ldyt, float* stats, long rows, int C, bool pk = false,
               const void* add = nullptr, float* sum32 = nullptr, vc_drop add_drop = vc_drop{0u, 0u, 1.0f}) const {
        LnFwdParams p; memset(&p, 0, sizeof(p));
        p.x = x; p.ldx = ldx; p.gamma = Pf(wo); p.beta = Pf(bo); p.y32 = y32; p.ldy32 = ldy32; p.yt = yt; p.ldyt = ldyt;
        p.stats = stats; p.rows = rows; p.eps = 1e-5f; p.add = add; p.ldadd = C; p.sum32 = sum32; p.ldsum = C; p.add_drop = add_drop;
        return vc_ln_fwd(tx, vt(pk), C, 0, p, s);
    }
    // dx32 (+T copy) = add_in + LNbwd(dy);  dgamma/dbeta written to the grad buffer
    // `du` (optional): also emit the masked copy du = T(dx * mask(site d)) that masked() would produce from dx32 in a second pass
    int ln_bwd(int td, const void* dy, long lddy, const float* x, long ldx, const float* stats, long wo, long bo,
               const float* add_in, long ldadd, float* dx32, long lddx, long rows, int C,
               vc_drop d = vc_drop{0u, 0u, 1.0f}, Mat* du = nullptr, void* du_dst = nullptr, float* du_colsum = nullptr, float* defer_partial = nullptr,
               bool du_pk = false, vc_drop d32 = vc_drop{0u, 0u, 1.0f}, int add_period = 0) const {
        LnBwdParams p; memset(&p, 0, sizeof(p)); p.drop32 = d32; p.add_period = add_period;
        p.dy = dy; p.lddy = lddy; p.x = x; p.ldx = ldx; p.stats = stats; p.gamma = Pf(wo);
        p.add_in = add_in; p.ldadd = ldadd; p.dx32 = dx32; p.lddx32 = lddx; p.rows = rows;
        if (du) {
            if (d.key || du_pk) { void* dst = du_dst ? du_dst : L().t_dum; p.dxt = dst; p.lddxt = C; p.drop = d; *du = VT(dst, C, du_pk); }   // (du_pk: pre-split copy even without a mask)
            else *du = A32(dx32, lddx);
        }
        // du_colsum: the bias gradient of the Linear that consumes du, reduced by this kernel instead of a column-sum pass over du
        // defer_partial: the dgamma / dbeta partial rows are left in that buffer (the caller reduces them later, in a grouped column sum)
        // (du_colsum with defer_partial: the kernel writes a third partial row per block — only its presence matters here, the deferred job table holds the destination)
        if (defer_partial) return vc_ln_bwd(td, VC_F32, vt(du_pk), C, 0, p, defer_partial, nullptr, nullptr, L().scr_colsum, s, du ? du_colsum : nullptr);
        return vc_ln_bwd(td, VC_F32, vt(du_pk), C, 0, p, L().scr_lnpart, Gf(wo), Gf(bo), L().scr_colsum, s, du ? du_colsum : nullptr);
    }
};

// The side stream of lane 1 (created on first use).  Without one (CPU emulator) the CAD ViT simply runs in line.
bool ensure_side(vcad_engine* e) {
    if (!vc_has_side_streams() || e->no_side || g_no_side || vc_profile_on()) return false;
    if (!e->side_ok) {
        if (vc_stream_create(&e->side) || vc_event_create(&e->ev_fork) || vc_event_create(&e->ev_fork2) || vc_event_create(&e->ev_join)) return false;
        e->side_ok = true;
    }
    return true;
}

// Class-token attention of tower v's last layer (vcad_engine::ClsPath).  Head h owns rows inner + 64 h .. (K slice) and 2 inner + 64 h .. (V slice) of
// to_qkv.weight [3 inner, D]; q / dout / out / dq are [N, inner] (64 columns per head), g / c / dc / dg are [N][H][D].
// One per-head projection for all heads: problem h at A + h bsa, W + h 64 D, C + h bsc (gemm_mid.h batched launch)
int cls_proj(const Ctx& cx, Mat A, long bsa, Mat W, int trb, Mat C, long bsc, long N, int Nn, int K, int role) {
    vcad_engine* e = cx.e;
    GemmCall gc; memset(&gc, 0, sizeof(gc));
    gc.role = role; gc.ct = e->ct; gc.sa = A.dt; gc.sb = W.dt; gc.to = C.dt; gc.tra = 0; gc.trb = trb; gc.flags = e->gemm_flags;
    GemmParams& p = gc.p;
    p.A = A.p; p.B = W.p; p.C = (void*)C.p; p.M = (int)N; p.N = Nn; p.K = K; p.lda = A.ld; p.ldb = W.ld; p.ldc = C.ld; p.alpha = 1.0f; p.rowadd_div = 1;
    CK(vc_gemm_mid_batched(gc, e->c.vit_heads, bsa, 64L * e->c.vit_dim, bsc, cx.s));
    ++e->kernel_launches[VC_TAG_GEMM_MID];
    return 0;
}
// descriptor table of the K / V weight-gradient slices (grouped launch): dW_k,h = q_h^T dg_h and dW_v,h = dout_h^T c_h, the frames cut into wg_split chunks
// (chunk s -> slab s; a single chunk goes straight into the gradient buffer)
int build_cls_wgrads(const Ctx& cx, int v) {
    vcad_engine* e = cx.e; const vcad_config& c = e->c; const VitW& w = e->wv[v]; VitActs& a = e->va[v];
    vcad_engine::ClsPath& cp = e->cls[v];
    const int D = c.vit_dim, H = c.vit_heads, inner = H * c.vit_dim_head;
    const long N = a.N, HD = (long)H * D;
    const VitW::L& wl = w.l[c.vit_depth - 1];
    const size_t es = e->esz;
    auto at = [&](const void* base, long off_elems) { return (const void*)((const char*)base + (size_t)off_elems * es); };
    vcad_engine::ClsPath::Tab& tb = cp.wg;
    tb.calls.clear(); tb.flops = 0;
    auto add = [&](Mat A, Mat B, float* C, int K) {
        GemmCall gc; memset(&gc, 0, sizeof(gc));
        gc.ct = e->ct; gc.sa = A.dt; gc.sb = B.dt; gc.to = VC_F32; gc.tra = 1; gc.trb = 1;
        GemmParams& p = gc.p;
        p.A = A.p; p.B = B.p; p.C = (void*)C; p.M = 64; p.N = D; p.K = K; p.lda = A.ld; p.ldb = B.ld; p.ldc = D; p.alpha = 1.0f; p.rowadd_div = 1;
        tb.calls.push_back(gc); tb.flops += 2.0 * 64 * D * K;
    };
    const void* dao = cx.L().t_dao;                    // (the out-projection's dgrad leaves the class rows compact: [N, inner])
    const int S = cp.wg_split; const long per = VC_CEIL_DIV(VC_CEIL_DIV(N, (long)S), 8) * 8;
    for (int h = 0; h < H; ++h)
        for (int sp = 0; sp < S; ++sp) {
            const long f0 = sp * per, nf = (f0 + per <= N ? per : N - f0);
            if (nf <= 0) { vc_set_error("internal: empty frame chunk in the class-token weight gradients"); return VC_ERR_ARG; }
            float* ok_ = S > 1 ? cp.wg_slab + (size_t)sp * 2 * inner * D + (size_t)64 * h * D : cx.Gf(wl.qkv + ((long)inner + 64 * h) * D);
            float* ov_ = S > 1 ? cp.wg_slab + (size_t)sp * 2 * inner * D + (size_t)inner * D + (size_t)64 * h * D : cx.Gf(wl.qkv + (2L * inner + 64 * h) * D);
            add(cx.AT(at(cp.q, f0 * inner + 64 * h), inner), cx.AT(at(cp.dg, f0 * HD + (long)h * D), HD), ok_, (int)nf);
            add(cx.AT(at(dao, f0 * inner + 64 * h), inner), cx.AT(at(cp.c, f0 * HD + (long)h * D), HD), ov_, (int)nf);
        }
    const int n = (int)tb.calls.size();
    std::vector<GemmParams> probs(n); std::vector<int> tiles(n + 1);
    CK(vc_gemm_grouped_prepare(tb.calls.data(), n, probs.data(), tiles.data(), 64));
    tb.total_tiles = tiles[n];
    CK(vc_upload(tb.d_probs, probs.data(), (size_t)n * sizeof(GemmParams), cx.s));
    CK(vc_upload(tb.d_tiles, tiles.data(), (size_t)(n + 1) * sizeof(int), cx.s));
    cp.bwd_ready = true; cp.bwd_lane = cx.ln;
    return 0;
}

int vit_forward(const Ctx& cx, int v, const void* img, long img_T, long img_bstride) {
    vcad_engine* e = cx.e; const vcad_config& c = e->c; const VitW& w = e->wv[v]; VitActs& a = e->va[v];
    const int D = c.vit_dim, inner = c.vit_heads * c.vit_dim_head, g = c.image_size / c.patch_size, P = g * g, pd = c.patch_size * c.patch_size;
    const long N = a.N, R = N * (P + 1), Rp = N * P;
    if (pd != 1024 || D != 512) { vc_set_error("engine: ViT dims (patch_dim=%d, dim=%d) unsupported by the LN kernels", pd, D); return VC_ERR_UNSUPPORTED; }
    {   // patchify + LN(1024)
        LnFwdParams p; memset(&p, 0, sizeof(p));
        p.x = img; p.gamma = cx.Pf(w.ln1w); p.beta = cx.Pf(w.ln1b); p.yt = a.pn; p.ldyt = pd; p.stats = a.pstat; p.rows = Rp; p.eps = 1e-5f;
        p.img = c.image_size; p.patch = c.patch_size; p.P = (int)img_T; p.ldx = img_bstride; p.u8 = v == 1 && e->in_u8 ? 1 : e->in_u8;     // (the CAD image is always one gray plane)
        if (a.pe_fold) {                                          // gamma / beta live in the folded Linear (VitActs): the patches are stored normalised only
            p.gamma = p.beta = nullptr;
            if (!e->pe_fresh[v]) { CK(vc_pe_fold(cx.Pf(w.pew), cx.Pf(w.peb), cx.Pf(w.ln1w), cx.Pf(w.ln1b), a.pe_wf, a.pe_bf, D, pd, cx.s)); e->pe_fresh[v] = true; }
        }
        CK(vc_ln_fwd(VC_F32, e->dt, pd, 1, p, cx.s));
    }
    if (a.pe_fold) { Epi ep; ep.bias = a.pe_bf; CK(cx.lin_fwd(cx.AT(a.pn, pd), cx.AT(a.pe_wf, pd), cx.A32(a.pe, D), (int)Rp, D, pd, ep)); }
    else { Epi ep; ep.bias = cx.Pf(w.peb); CK(cx.lin_fwd(cx.AT(a.pn, pd), cx.W(w.pew, pd), cx.A32(a.pe, D), (int)Rp, D, pd, ep)); }
    {   // LN(512) + cls + pos  -> x0
        LnFwdParams p; memset(&p, 0, sizeof(p));
        p.x = a.pe; p.ldx = D; p.gamma = cx.Pf(w.ln2w); p.beta = cx.Pf(w.ln2b); p.y32 = a.x0; p.ldy32 = D; p.stats = a.stat2;
        p.rows = R; p.eps = 1e-5f; p.pos = cx.Pf(w.pos); p.cls = cx.Pf(w.cls); p.P = P; p.drop = cx.site(v + 1, 0, Ctx::K_EMB);
        CK(vc_ln_fwd(VC_F32, VC_F32, D, 2, p, cx.s));
    }
    const float* x = a.x0;
    const long TD = (long)(P + 1) * D, TI = (long)(P + 1) * inner;      // per-frame strides: cls rows are rows n*(P+1)
    // r05, 16-bit engines: a full layer's to_out / net.4 Linears write their (bias + dropout) output y as a 16-bit tensor with the persistent kernel's
    // wide-tile epilogue, and the LayerNorm pass that follows adds it to the residual stream (x + y -> xm / xo in fp32, then the statistics):
    // the residual epilogue of the 128-wide tile ran at 0.17 matrix-core busy (profiles/r04_summary.md), the LayerNorm pass is HBM-bound either way.
    const bool res_in_ln = e->dt == VC_BF16 && e->ct == VC_BF16 && !e->fp8 && g_res_in_ln;
    // The branch's dropout moves with the add (same site, same element indices: the backward regenerates the same masks), so the Linear's epilogue is the
    // plain one — bias folded into the accumulator start, convert, store.
    const void* pend = nullptr; float* pend_sum = nullptr; vc_drop pend_drop = vc_drop{0u, 0u, 1.0f};     // branch output not yet added: the next LayerNorm pass writes x + mask * pend -> pend_sum
    for (int L = 0; L < c.vit_depth; ++L) {
        const VitW::L& wl = w.l[L]; VitLayerActs& l = a.L[L];
        // Only the cls token of the LAST layer is consumed (pool='cls'): its Q projection, attention output, out-proj and
        // MLP are computed for the cls row only; K/V still need every token.  Same numbers, ~60 % less work in that layer.
        const bool cls_only = (L == c.vit_depth - 1);
        const bool pk = cx.pk_acts() && !cls_only;          // bf16x3: this layer's GEMM-only tensors (h_a, qkv, ao, h_f, z, g) are stored pre-split
        const float scale = 1.0f / sqrtf((float)c.vit_dim_head);
        const char* q = (const char*)l.qkv;
        CK(cx.ln_fwd(VC_F32, x, D, wl.anw, wl.anb, nullptr, 0, l.h_a, D, l.stat_a, R, D, pk, pend, pend_sum, pend_drop));
        if (pend) { x = pend_sum; pend = nullptr; pend_sum = nullptr; }
        AttnParams ap; memset(&ap, 0, sizeof(ap));
        ap.q = q; ap.k = q + (size_t)inner * e->esz; ap.v = q + (size_t)2 * inner * e->esz; ap.o = l.ao;
        ap.ldq = ap.ldk = ap.ldv = 3 * inner; ap.ldo = inner; ap.lse = l.lse;
        ap.B = (int)N; ap.H = c.vit_heads; ap.Tq = ap.Tk = P + 1; ap.window = P + 1; ap.causal = 0; ap.scale = scale;
        ap.drop = cx.site(v + 1, L, Ctx::K_ATTN); ap.x3 = pk ? 2 : (e->ct == VC_X3 ? 1 : 0);
        const vc_drop d_out = cx.site(v + 1, L, Ctx::K_OUT), d_act = cx.site(v + 1, L, Ctx::K_MLP_ACT), d_mlp = cx.site(v + 1, L, Ctx::K_MLP_OUT);
        if (!cls_only) {
            const bool q8 = e->fp8 && e->dt == VC_BF16;       // VCAD_FP8: these four Linears on the block-scaled fp8 matrix cores
            if (q8) CK(cx.lin_fwd_q(cx.AT(l.h_a, D), wl.qkv, cx.AT(l.qkv, 3 * inner), (int)R, 3 * inner, D, Epi()));
            else CK(cx.lin_fwd(cx.VT(l.h_a, D, pk), cx.W(wl.qkv, D), cx.VT(l.qkv, 3 * inner, pk), (int)R, 3 * inner, D, Epi()));
            CK(vc_attn_fwd(e->dt, c.vit_dim_head, ap, cx.s));
            if (res_in_ln) {
                { Epi ep; ep.bias = cx.Pf(wl.ob); CK(cx.lin_fwd(cx.AT(l.ao, inner), cx.W(wl.ow, inner), cx.AT(cx.L().t_y[0], D), (int)R, D, inner, ep)); }
                CK(cx.ln_fwd(VC_F32, x, D, wl.fnw, wl.fnb, nullptr, 0, l.h_f, D, l.stat_f, R, D, false, cx.L().t_y[0], l.xm, d_out));
            } else {
            { Epi ep; ep.bias = cx.Pf(wl.ob); ep.residual = x; ep.ldr = D; ep.drop = d_out;
              if (q8) CK(cx.lin_fwd_q(cx.AT(l.ao, inner), wl.ow, cx.A32(l.xm, D), (int)R, D, inner, ep));
              else CK(cx.lin_fwd(cx.VT(l.ao, inner, pk), cx.W(wl.ow, inner), cx.A32(l.xm, D), (int)R, D, inner, ep)); }
            CK(cx.ln_fwd(VC_F32, l.xm, D, wl.fnw, wl.fnb, nullptr, 0, l.h_f, D, l.stat_f, R, D, pk));
            }
            if (!q8 && e->dt == VC_BF16 && g_split_gelu) {     // bf16: plain GEMM (persistent kernel) -> z, then the activation pass (norm.h: act_fwd_bf16_kernel)
                { Epi ep; ep.bias = cx.Pf(wl.b1); CK(cx.lin_fwd(cx.AT(l.h_f, D), cx.W(wl.w1, D), cx.AT(l.z, c.vit_mlp), (int)R, c.vit_mlp, D, ep)); }
                CK(vc_act_fwd_bf16(l.z, l.g, R, c.vit_mlp, VC_ACT_GELU, d_act, cx.s));
            } else {
              Epi ep; ep.bias = cx.Pf(wl.b1); ep.act = VC_ACT_GELU; ep.aux = l.z; ep.ldaux = c.vit_mlp; ep.drop = d_act;
              if (q8) CK(cx.lin_fwd_q(cx.AT(l.h_f, D), wl.w1, cx.AT(l.g, c.vit_mlp), (int)R, c.vit_mlp, D, ep));
              else CK(cx.lin_fwd(cx.VT(l.h_f, D, pk), cx.W(wl.w1, D), cx.VT(l.g, c.vit_mlp, pk), (int)R, c.vit_mlp, D, ep)); }     // (pk: the pre-activation side output z is pre-split too — it has the output's type)
            if (res_in_ln) {     // xo = xm + y: written by the NEXT layer's first LayerNorm pass (every layer, the class-token-only last one included, starts with one over all rows)
                Epi ep; ep.bias = cx.Pf(wl.b4); CK(cx.lin_fwd(cx.AT(l.g, c.vit_mlp), cx.W(wl.w4, c.vit_mlp), cx.AT(cx.L().t_y[1], D), (int)R, D, c.vit_mlp, ep));
                pend = cx.L().t_y[1]; pend_sum = l.xo; pend_drop = d_mlp;
            } else {
              Epi ep; ep.bias = cx.Pf(wl.b4); ep.residual = l.xm; ep.ldr = D; ep.drop = d_mlp;
              if (q8) CK(cx.lin_fwd_q(cx.AT(l.g, c.vit_mlp), wl.w4, cx.A32(l.xo, D), (int)R, D, c.vit_mlp, ep));
              else CK(cx.lin_fwd(cx.VT(l.g, c.vit_mlp, pk), cx.W(wl.w4, c.vit_mlp), cx.A32(l.xo, D), (int)R, D, c.vit_mlp, ep)); }
        } else {
            vcad_engine::ClsPath& cp = e->cls[v];
            if (cp.on) {        // r06: scores and context on (g = q W_k, normalised tokens) — no K / V projections (attn_cls.h)
                const long HD = (long)c.vit_heads * D;
                const Mat Wk = cx.W(wl.qkv + (long)inner * D, D), Wv = cx.W(wl.qkv + 2L * inner * D, D);
                CK(cx.lin_fwd(cx.AT(l.h_a, TD), cx.W(wl.qkv, D), cx.AT(cp.q, inner), (int)N, inner, D, Epi()));                                      // Q: cls rows
                CK(cls_proj(cx, cx.AT(cp.q, inner), 64, Wk, 1, cx.AT(cp.g, HD), D, N, D, 64, VC_CAT_GEMM_FWD + 1));                                  // g_h = q_h W_k,h
                ClsAttnParams ca; memset(&ca, 0, sizeof(ca));
                ca.ha = l.h_a; ca.ld_ha = D; ca.g = cp.g; ca.c = cp.c; ca.lse = l.lse; ca.N = (int)N; ca.H = c.vit_heads; ca.P1 = P + 1; ca.scale = scale; ca.drop = ap.drop;
                CK(vc_cls_attn_fwd(ca, cx.s));
                CK(cls_proj(cx, cx.AT(cp.c, HD), D, Wv, 0, cx.AT(l.ao, TI), 64, N, 64, D, VC_CAT_GEMM_FWD + 1));                                     // out_h = c_h W_v,h^T (class rows of ao)
            } else {
            CK(cx.lin_fwd(cx.AT(l.h_a, D), cx.W(wl.qkv + (long)inner * D, D), cx.AT(q + (size_t)inner * e->esz, 3 * inner), (int)R, 2 * inner, D, Epi()));   // K, V: all tokens
            CK(cx.lin_fwd(cx.AT(l.h_a, TD), cx.W(wl.qkv, D), cx.AT(l.qkv, 3 * TI), (int)N, inner, D, Epi()));                                       // Q: cls rows
            ap.Tq = 1; ap.ldq = 3 * TI; ap.ldo = TI;                                   // query row b -> cls row of frame b
            CK(vc_attn_fwd(e->dt, c.vit_dim_head, ap, cx.s));
            }
            { Epi ep; ep.bias = cx.Pf(wl.ob); ep.residual = x; ep.ldr = TD; ep.drop = d_out; CK(cx.lin_fwd(cx.AT(l.ao, TI), cx.W(wl.ow, inner), cx.A32(l.xm, TD), (int)N, D, inner, ep)); }
            CK(cx.ln_fwd(VC_F32, l.xm, TD, wl.fnw, wl.fnb, nullptr, 0, l.h_f, D, l.stat_f, N, D));                   // h_f, z, g: compact [N, .]
            { Epi ep; ep.bias = cx.Pf(wl.b1); ep.act = VC_ACT_GELU; ep.aux = l.z; ep.ldaux = c.vit_mlp; ep.drop = d_act;
              CK(cx.lin_fwd(cx.AT(l.h_f, D), cx.W(wl.w1, D), cx.AT(l.g, c.vit_mlp), (int)N, c.vit_mlp, D, ep)); }
            { Epi ep; ep.bias = cx.Pf(wl.b4); ep.residual = l.xm; ep.ldr = TD; ep.drop = d_mlp;
              CK(cx.lin_fwd(cx.AT(l.g, c.vit_mlp), cx.W(wl.w4, c.vit_mlp), cx.A32(l.xo, TD), (int)N, D, c.vit_mlp, ep)); }
        }
        x = pend ? (const float*)l.xm : (const float*)l.xo;          // (with a pending branch output the stream is still xm; xo appears with the next LayerNorm pass)
    }
    if (pend) { vc_set_error("internal: ViT forward ended with an unadded branch output"); return VC_ERR_ARG; }       // (cannot happen: the last layer is the class-token-only one)
    // final LN on the cls row only (pool = 'cls', mlp_head = Identity)
    CK(cx.ln_fwd(VC_F32, x, (long)(P + 1) * D, w.normw, w.normb, nullptr, 0, a.e, D, a.statn, N, D));
    return 0;
}

// Job table of one (tower, part) of vit_backward: every layer's deferred column sums (vcad_engine::VitColsums), in the layer loop's order and under its
// conditions — norm weight / bias pairs are adjacent in the flat buffer (one 2 D-wide job), the third partial row of a LayerNorm backward is the bias
// gradient of the Linear that consumes the masked gradient it emitted (to_out.bias / the layer below's net.4 bias).
int build_vit_colsums(const Ctx& cx, int v, int part) {
    vcad_engine* e = cx.e; const vcad_config& c = e->c; const VitW& w = e->wv[v]; VitActs& a = e->va[v];
    vcad_engine::VitColsums& vc = e->vcs[v][part];
    const int D = c.vit_dim, g = c.image_size / c.patch_size, P = g * g;
    const long N = a.N, R = N * (P + 1);
    const int split_l = c.vit_depth / 2;
    int Lhi = c.vit_depth - 1, Llo = 0;
    if (part == 1) Llo = split_l; if (part == 2) Lhi = split_l - 1;
    vc.jobs.clear();
    auto add = [&](const float* x, long ld, long rows, int cols, float* out) {
        ColsumJob j; memset(&j, 0, sizeof(j));
        j.x = x; j.ld = ld; j.rows = (int)rows; j.cols = cols; j.out = out; j.is_bf16 = 0;
        vc.jobs.push_back(j);
    };
    for (int L = Lhi; L >= Llo; --L) {
        const VitW::L& wl = w.l[L]; const VitLayerActs& l = a.L[L];
        const bool cls_only = (L == c.vit_depth - 1);
        const long Rm = cls_only ? N : R;
        if (wl.fnb != wl.fnw + D || wl.anb != wl.anw + D) { vc_set_error("internal: ViT LayerNorm weight / bias not adjacent"); return VC_ERR_ARG; }
        if (e->dt == VC_BF16 && g_split_gelu && !cls_only) add(l.part_b1, c.vit_mlp, vc_dact_bwd_blocks(Rm, c.vit_mlp), c.vit_mlp, cx.Gf(wl.b1));
        add(l.part_fn, 3L * D, vc_ln_bwd_blocks(Rm), 2 * D, cx.Gf(wl.fnw));
        add(l.part_fn + 2 * D, 3L * D, vc_ln_bwd_blocks(Rm), D, cx.Gf(wl.ob));
        if (L > Llo || (part == 1 && L > 0)) {              // (part 1's lowest layer hands its masked gradient to part 2: vcad_engine::du_carry)
            add(l.part_an, 3L * D, vc_ln_bwd_blocks(R), 2 * D, cx.Gf(wl.anw));
            add(l.part_an + 2 * D, 3L * D, vc_ln_bwd_blocks(R), D, cx.Gf(w.l[L - 1].b4));
        } else add(l.part_an, 2L * D, vc_ln_bwd_blocks(R), 2 * D, cx.Gf(wl.anw));
    }
    int strips = 0, chunks = 1; long poff = 0;
    for (auto& j : vc.jobs) {
        j.strip_start = strips; j.part_off = poff; strips += VC_CEIL_DIV(j.cols, 256);
        const int ch = VC_CEIL_DIV(j.rows, 128); poff += (long)ch * j.cols; chunks = ch > chunks ? ch : chunks;
    }
    if ((int)vc.jobs.size() > c.vit_depth * 5 || poff > (long)c.vit_depth * (2 * 4 * 3L * D + 17L * c.vit_mlp)) { vc_set_error("internal: ViT column-sum table overflows its plan"); return VC_ERR_WORKSPACE; }
    vc.strips = strips; vc.chunks = chunks;
    CK(vc_upload(vc.d_jobs, vc.jobs.data(), vc.jobs.size() * sizeof(ColsumJob), cx.s));
    vc.ready = true;
    return 0;
}

// de: fp32 [N, D] gradient of the cls embedding.  first/last layer range lets the caller split into DDP stages.
int vit_backward(const Ctx& cx, int v, const float* de, int part /*0 = whole, 1 = top half (cls-LN + upper layers), 2 = bottom half + embed*/,
                 const void* img, long img_T, long img_bstride) {
    vcad_engine* e = cx.e; const vcad_config& c = e->c; const VitW& w = e->wv[v]; VitActs& a = e->va[v];
    const int D = c.vit_dim, inner = c.vit_heads * c.vit_dim_head, g = c.image_size / c.patch_size, P = g * g, pd = c.patch_size * c.patch_size;
    const long N = a.N, R = N * (P + 1), Rp = N * P;
    float* dx = cx.L().t_dx;
    if (v == 0) CK(cx.refresh_wT());
    const int split = c.vit_depth / 2;
    int Lhi = c.vit_depth - 1, Llo = 0;
    if (part == 1) Llo = split; if (part == 2) Lhi = split - 1;
    const bool defer_cs = true;                                                // the layers' small column sums: one grouped launch at the end (VitColsums)
    // r06: emb_dropout's backward mask applied by layer 0's norm backward — when THIS call both walks layer 0 and runs the embedding's backward
    const bool emb_in_ln = e->dt == VC_BF16 && e->ct == VC_BF16 && part != 1 && Llo == 0 && Lhi >= 0;
    if (defer_cs && !e->vcs[v][part].ready) CK(build_vit_colsums(cx, v, part));
    // The class-token-only last layer leaves the stream gradient on class rows only.  r01-r05 cleared all of dx first and the layer's attention-norm backward read
    // the zeros back; 16-bit engines with more than one layer now tell that pass which rows exist (LnBwdParams::add_period): no 210 MB memset, no 210 MB of zeros read.
    const bool sparse_dx = e->dt == VC_BF16 && e->ct == VC_BF16 && c.vit_depth > 1 && part != 2;
    if (part != 2) {
        if (!sparse_dx) CK(vc_memset_async(dx, 0, (size_t)R * D * 4, cx.s));
        const float* xl = a.L[c.vit_depth - 1].xo;
        CK(cx.ln_bwd(VC_F32, de, D, xl, (long)(P + 1) * D, a.statn, w.normw, w.normb, nullptr, 0, dx, (long)(P + 1) * D, N, D));
    }
    const long TD = (long)(P + 1) * D, TI = (long)(P + 1) * inner;
    Mat du{nullptr, VC_F32, 0}; bool have_du = false;          // masked gradient entering the current Linear (and whether the previous LayerNorm backward already produced it)
    if (v == 0 && part == 2 && e->du_carry) { du = e->du_carry_mat; have_du = true; }      // ... by the upper stage's last one
    if (v == 0) e->du_carry = false;
    for (int L = Lhi; L >= Llo; --L) {
        const VitW::L& wl = w.l[L]; VitLayerActs& l = a.L[L];
        const float* xin = L == 0 ? a.x0 : a.L[L - 1].xo;
        const bool cls_only = (L == c.vit_depth - 1);          // see vit_forward: dx is non-zero on cls rows only here
        const long Rm = cls_only ? N : R;                       // rows the MLP / out-proj backward runs over
        const long ldx = cls_only ? TD : D, ldao = cls_only ? TI : inner;
        const bool pk = cx.pk_acts() && !cls_only;              // bf16x3: this layer's saved tensors and GEMM-only gradients (du, dz, dao, dqkv) are pre-split (vit_forward)
        // MLP (x' = xm + drop(W4 drop(gelu(z)) + b4)): the gradient entering W4 is dx * mask_out
        if (!have_du) CK(cx.masked(dx, ldx, Rm, D, cx.site(v + 1, L, Ctx::K_MLP_OUT), &du, nullptr, pk));     // else: emitted by the layer above's LayerNorm backward
        const bool have_db4 = have_du;
        have_du = false;
        // r06: a full layer's three small weight gradients (net.4, net.0, to_out: token reductions over the same R rows) run as ONE launch of the persistent kernel once
        // the third one's masked gradient exists (ops_gemm_dma.hip vc_gemm_dma_wgrad_batched) — train mode only: every dY is then a private 16-bit copy that lives that long
        // (the attention norm's backward emits its du into a forward-only temporary instead of over the MLP's)
        const bool batch3 = g_batch_wg && e->dt == VC_BF16 && e->ct == VC_BF16 && !e->fp8 && !cls_only && !pk && e->drop_p > 0.f && Rm % 64 == 0 && (R >= 8192 || (e->gemm_flags & VC_GF_DMA_ALWAYS)) && du.dt == e->dt;      // (small towers stay on the small-problem kernels; tests force the persistent kernel)
        const Mat du_mlp = du;
        if (!batch3) CK(cx.lin_wgrad(du, cx.VT(l.g, c.vit_mlp, pk), cx.Gf(wl.w4), c.vit_mlp, have_db4 ? nullptr : cx.Gf(wl.b4), (int)Rm, D, c.vit_mlp));   // (b4's gradient: from the LayerNorm backward that emitted du)
        else if (!have_db4) CK(cx.colsum(du, Rm, D, cx.Gf(wl.b4), 0));
        bool have_db1 = false;
        { Epi ep; ep.dact = l.z; ep.lddact = c.vit_mlp; ep.dkind = VC_ACT_GELU; ep.drop = cx.site(v + 1, L, Ctx::K_MLP_ACT);
          const bool split = e->dt == VC_BF16 && g_split_gelu && !cls_only;      // bf16: plain dgrad, then the activation-derivative pass
          const Epi epg = split ? Epi() : ep;
          if (cx.hasT(wl.w4T)) CK(cx.lin_dgrad_T(du, cx.WT(wl.w4T, D), cx.AT(cx.L().t_dz, c.vit_mlp), (int)Rm, D, c.vit_mlp, epg));
          else CK(cx.lin_dgrad(du, cx.W(wl.w4, c.vit_mlp), cx.VT(cx.L().t_dz, c.vit_mlp, pk), (int)Rm, D, c.vit_mlp, epg));
          // (the activation-derivative pass also reduces its output over rows: b1's gradient without a column-sum pass over dz)
          if (split && defer_cs) CK(vc_dact_bwd_bf16(cx.L().t_dz, l.z, Rm, c.vit_mlp, VC_ACT_GELU, ep.drop, cx.s, cx.Gf(wl.b1), l.part_b1, (size_t)vc_dact_bwd_blocks(Rm, c.vit_mlp) * c.vit_mlp * 4, nullptr, true));
          else if (split) CK(vc_dact_bwd_bf16(cx.L().t_dz, l.z, Rm, c.vit_mlp, VC_ACT_GELU, ep.drop, cx.s, cx.Gf(wl.b1), cx.L().scr_lnpart, cx.L().scr_lnpart_bytes, cx.L().scr_colsum));
          have_db1 = split; }
        if (!batch3) CK(cx.lin_wgrad(cx.VT(cx.L().t_dz, c.vit_mlp, pk), cx.VT(l.h_f, D, pk), cx.Gf(wl.w1), D, have_db1 ? nullptr : cx.Gf(wl.b1), (int)Rm, c.vit_mlp, D));
        else if (!have_db1) CK(cx.colsum(cx.AT(cx.L().t_dz, c.vit_mlp), Rm, c.vit_mlp, cx.Gf(wl.b1), 0));
        if (cx.hasT(wl.w1T)) CK(cx.lin_dgrad_T(cx.AT(cx.L().t_dz, c.vit_mlp), cx.WT(wl.w1T, c.vit_mlp), cx.AT(cx.L().t_dh, D), (int)Rm, c.vit_mlp, D, Epi()));
        else CK(cx.lin_dgrad(cx.VT(cx.L().t_dz, c.vit_mlp, pk), cx.W(wl.w1, D), cx.AT(cx.L().t_dh, D), (int)Rm, c.vit_mlp, D, Epi()));
        // attention block (xm = x + drop(Wo ao + bo)): the LayerNorm backward also emits du = dx * mask_out
        CK(cx.ln_bwd(e->dt, cx.L().t_dh, D, l.xm, ldx, l.stat_f, wl.fnw, wl.fnb, dx, ldx, dx, ldx, Rm, D, cx.site(v + 1, L, Ctx::K_OUT), &du, batch3 ? cx.L().t_y[0] : nullptr, cx.Gf(wl.ob), defer_cs ? l.part_fn : nullptr, pk));
        if (batch3 && du.dt == e->dt) {
            GemmCall gc[3];
            auto fill = [&](GemmCall& g, Mat dY, Mat X, long w_off, int N_, int K_) {
                memset(&g, 0, sizeof(g));
                g.ct = e->ct; g.sa = dY.dt; g.sb = X.dt; g.to = VC_F32; g.tra = 1; g.trb = 1; g.flags = e->gemm_flags; g.claim = (e->gemm_flags & VC_GF_DYNAMIC) ? cx.L().claim : nullptr;
                g.p.A = dY.p; g.p.B = X.p; g.p.C = (void*)cx.Gf(w_off); g.p.M = N_; g.p.N = K_; g.p.K = (int)Rm; g.p.lda = dY.ld; g.p.ldb = X.ld; g.p.ldc = K_; g.p.alpha = 1.0f; g.p.rowadd_div = 1;
            };
            fill(gc[0], du_mlp, cx.AT(l.g, c.vit_mlp), wl.w4, D, c.vit_mlp);
            fill(gc[1], cx.AT(cx.L().t_dz, c.vit_mlp), cx.AT(l.h_f, D), wl.w1, c.vit_mlp, D);
            fill(gc[2], du, cx.AT(l.ao, ldao), wl.ow, D, inner);
            CK(vc_gemm_dma_wgrad_batched(gc, 3, cx.L().scr_splitk, cx.L().scr_splitk_bytes, cx.s));
            ++e->kernel_launches[VC_TAG_GEMM_DMA];
        } else {
            if (batch3) {      // (cannot happen in train mode: the norm backward emitted no private copy) the two deferred gradients on their own
                CK(cx.lin_wgrad(du_mlp, cx.AT(l.g, c.vit_mlp), cx.Gf(wl.w4), c.vit_mlp, nullptr, (int)Rm, D, c.vit_mlp));
                CK(cx.lin_wgrad(cx.AT(cx.L().t_dz, c.vit_mlp), cx.AT(l.h_f, D), cx.Gf(wl.w1), D, nullptr, (int)Rm, c.vit_mlp, D));
            }
            CK(cx.lin_wgrad(du, cx.VT(l.ao, ldao, pk), cx.Gf(wl.ow), inner, nullptr, (int)Rm, D, inner));       // (ob's gradient: reduced by the LayerNorm backward above)
        }
        if (cx.hasT(wl.owT)) CK(cx.lin_dgrad_T(du, cx.WT(wl.owT, D), cx.AT(cx.L().t_dao, inner), (int)Rm, D, inner, Epi()));
        else CK(cx.lin_dgrad(du, cx.W(wl.ow, inner), cx.VT(cx.L().t_dao, inner, pk), (int)Rm, D, inner, Epi()));
        if (cls_only && e->cls[v].on) {      // r06 (attn_cls.h): dc = dout W_v -> kernel (dg, dh of every token) -> dq = dg W_k^T; weight gradients of the three slices
            vcad_engine::ClsPath& cp = e->cls[v];
            const long HD = (long)c.vit_heads * D;
            const Mat Wk = cx.W(wl.qkv + (long)inner * D, D), Wv = cx.W(wl.qkv + 2L * inner * D, D);
            if (!cp.bwd_ready || cp.bwd_lane != cx.ln) CK(build_cls_wgrads(cx, v));
            CK(cls_proj(cx, cx.AT(cx.L().t_dao, inner), 64, Wv, 1, cx.AT(cp.dc, HD), D, N, D, 64, VC_CAT_GEMM_DGRAD + 1));                           // dc_h = dout_h W_v,h
            ClsAttnParams ca; memset(&ca, 0, sizeof(ca));
            ca.ha = l.h_a; ca.ld_ha = D; ca.g = cp.g; ca.dc = cp.dc; ca.lse = l.lse; ca.dg = cp.dg; ca.dha = cx.L().t_dh; ca.ld_dha = D; ca.r0 = cp.r0;
            ca.N = (int)N; ca.H = c.vit_heads; ca.P1 = P + 1; ca.scale = 1.0f / sqrtf((float)c.vit_dim_head); ca.drop = cx.site(v + 1, L, Ctx::K_ATTN);
            CK(vc_cls_attn_bwd(ca, cx.s));
            CK(cls_proj(cx, cx.AT(cp.dg, HD), D, Wk, 0, cx.AT(cp.dq, inner), 64, N, 64, D, VC_CAT_GEMM_DGRAD + 1));                                   // dq_h = dg_h W_k,h^T
            CK(vc_gemm_grouped_launch(cp.wg.calls[0], cp.wg.d_probs, cp.wg.d_tiles, (int)cp.wg.calls.size(), cp.wg.total_tiles, cp.wg.flops, cx.s, 64));   // K and V slices of to_qkv.weight
            e->kernel_launches[VC_TAG_GEMM_GROUPED]++;
            if (cp.wg_split > 1) CK(cx.colsum(cx.A32(cp.wg_slab, 2L * inner * D), cp.wg_split, 2 * inner * D, cx.Gf(wl.qkv + (long)inner * D), 0));
            CK(cx.lin_wgrad(cx.AT(cp.dq, inner), cx.AT(l.h_a, TD), cx.Gf(wl.qkv), D, nullptr, (int)N, inner, D));                    // Q slice: class rows only
            { Epi ep; ep.residual = cp.r0; ep.ldr = D;                                                                              // class row of dh: + dq W_q (fp32 sum, rounded once)
              CK(cx.lin_dgrad(cx.AT(cp.dq, inner), cx.W(wl.qkv, D), cx.AT(cx.L().t_dh, TD), (int)N, inner, D, ep)); }
        } else {
        {
            AttnParams p; memset(&p, 0, sizeof(p));
            const char* q = (const char*)l.qkv; char* dq = (char*)cx.L().t_dqkv;
            p.q = q; p.k = q + (size_t)inner * e->esz; p.v = q + (size_t)2 * inner * e->esz;
            p.ldq = p.ldk = p.ldv = 3 * inner; p.lse = l.lse; p.delta = cx.L().t_delta;
            p.dout = cx.L().t_dao; p.lddo = inner; p.dq = dq; p.dk = dq + (size_t)inner * e->esz; p.dv = dq + (size_t)2 * inner * e->esz;
            p.lddq = p.lddk = p.lddv = 3 * inner;
            p.B = (int)N; p.H = c.vit_heads; p.Tq = p.Tk = P + 1; p.window = P + 1; p.causal = 0; p.scale = 1.0f / sqrtf((float)c.vit_dim_head);
            p.drop = cx.site(v + 1, L, Ctx::K_ATTN); p.x3 = pk ? 2 : (e->ct == VC_X3 ? 1 : 0);
            if (cls_only) {       // one query (the cls row) per frame; dQ of every other token is zero — the Q third of every row (dK / dV are written for all tokens)
                CK(vc_zero_cols(cx.L().t_dqkv, (long)3 * inner * e->esz, R, (long)inner * e->esz, cx.s));
                p.Tq = 1; p.ldq = 3 * TI; p.lddq = 3 * TI;
            }
            CK(vc_attn_bwd(e->dt, c.vit_dim_head, p, cx.s));
        }
        CK(cx.lin_wgrad(cx.VT(cx.L().t_dqkv, 3 * inner, pk), cx.VT(l.h_a, D, pk), cx.Gf(wl.qkv), D, nullptr, (int)R, 3 * inner, D));
        if (cx.hasT(wl.qkvT)) CK(cx.lin_dgrad_T(cx.AT(cx.L().t_dqkv, 3 * inner), cx.WT(wl.qkvT, 3 * inner), cx.AT(cx.L().t_dh, D), (int)R, 3 * inner, D, Epi()));
        else CK(cx.lin_dgrad(cx.VT(cx.L().t_dqkv, 3 * inner, pk), cx.W(wl.qkv, D), cx.AT(cx.L().t_dh, D), (int)R, 3 * inner, D, Epi()));
        }
        // (the layer below is never the cls-only one: the du it receives is pre-split whenever the mode stores pre-split tensors)
        if (L > Llo || (part == 1 && L > 0)) {
            CK(cx.ln_bwd(e->dt, cx.L().t_dh, D, xin, D, l.stat_a, wl.anw, wl.anb, dx, D, dx, D, R, D, cx.site(v + 1, L - 1, Ctx::K_MLP_OUT), &du, nullptr, cx.Gf(w.l[L - 1].b4), defer_cs ? l.part_an : nullptr, cx.pk_acts(),
                         vc_drop{0u, 0u, 1.0f}, sparse_dx && cls_only ? P + 1 : 0)); have_du = true;
            if (L == Llo && v == 0) { e->du_carry = true; e->du_carry_mat = du; }        // stage boundary: the lower stage (vit_backward part 2, next call) starts from this du
        }
        // bottom layer: the embedding dropout (everything below sees dx * mask) rides on this pass's fp32 output (16-bit engines; norm.h LnBwdParams::drop32)
        else CK(cx.ln_bwd(e->dt, cx.L().t_dh, D, xin, D, l.stat_a, wl.anw, wl.anb, dx, D, dx, D, R, D, vc_drop{0u, 0u, 1.0f}, nullptr, nullptr, nullptr, defer_cs ? l.part_an : nullptr, false,
                          emb_in_ln ? cx.site(v + 1, 0, Ctx::K_EMB) : vc_drop{0u, 0u, 1.0f}));
    }
    if (defer_cs) { const vcad_engine::VitColsums& vc = e->vcs[v][part]; CK(vc_colsum_grouped(vc.d_jobs, (int)vc.jobs.size(), vc.strips, vc.chunks, vc.partial, cx.s)); }
    if (part != 1) {
        { const vc_drop d = cx.site(v + 1, 0, Ctx::K_EMB);      // emb_dropout: everything below sees dx * mask
          if (d.key && !emb_in_ln) CK(vc_dropout_mul(VC_F32, dx, D, dx, D, R, D, d, cx.s)); }
        // pos / cls gradients: column sums over frames of dx viewed as [N, (P+1)*D]
        CK(cx.colsum(cx.A32(dx, (long)(P + 1) * D), N, (P + 1) * D, cx.Gf(w.pos), 0));
        CK(vc_memcpy_d2d_async(cx.Gf(w.cls), cx.Gf(w.pos), (size_t)D * 4, cx.s));
        {   // LN(512) backward through the embed mapping
            LnBwdParams p; memset(&p, 0, sizeof(p));
            p.dy = dx; p.lddy = D; p.x = a.pe; p.ldx = D; p.stats = a.stat2; p.gamma = cx.Pf(w.ln2w);
            p.rows = Rp; p.P = P;
            // bf16 mode: the gradient entering the patch-embed Linear is emitted in bf16 (the GEMMs rounded it to bf16 while staging anyway — same
            // values), so its wgrad / dgrad are all-bf16 problems for the persistent kernel instead of fp32-source ones for the register-staged one
            if (e->dt == VC_BF16) { p.dxt = cx.L().t_dpe; p.lddxt = D; } else { p.dx32 = cx.L().t_dpe; p.lddx32 = D; }
            CK(vc_ln_bwd(VC_F32, VC_F32, e->dt == VC_BF16 ? VC_BF16 : VC_F32, D, 2, p, cx.L().scr_lnpart, cx.Gf(w.ln2w), cx.Gf(w.ln2b), cx.L().scr_colsum, cx.s));
        }
        const Mat dpe = e->dt == VC_BF16 ? cx.AT(cx.L().t_dpe, D) : cx.A32(cx.L().t_dpe, D);
        if (a.pe_fold) {     // folded LayerNorm affine: dWf = dpe^T (normalised patches) carries dW, dgamma and dbeta (norm.h pe_fold_bwd_kernel); the frames take no gradient
            CK(cx.lin_wgrad(dpe, cx.AT(a.pn, pd), a.t_dwf, pd, cx.Gf(w.peb), (int)Rp, D, pd));
            if (w.ln1b != w.ln1w + pd) { vc_set_error("internal: patch-embedding LayerNorm weight / bias not adjacent"); return VC_ERR_ARG; }
            CK(vc_pe_fold_bwd(a.t_dwf, cx.Gf(w.peb), cx.Pf(w.pew), cx.Pf(w.ln1w), cx.Pf(w.ln1b), cx.Gf(w.pew), cx.Gf(w.ln1w), D, pd, cx.L().scr_lnpart, cx.L().scr_colsum, cx.s));
        } else {
        CK(cx.lin_wgrad(dpe, cx.AT(a.pn, pd), cx.Gf(w.pew), pd, cx.Gf(w.peb), (int)Rp, D, pd));
        CK(cx.lin_dgrad(dpe, cx.W(w.pew, pd), cx.AT(cx.L().t_dpn, pd), (int)Rp, D, pd, Epi()));
        {   // LN(1024) parameter gradients (input frames need no gradient)
            LnBwdParams p; memset(&p, 0, sizeof(p));
            p.dy = cx.L().t_dpn; p.lddy = pd; p.x = img; p.ldx = img_bstride; p.stats = a.pstat; p.gamma = cx.Pf(w.ln1w); p.rows = Rp;
            p.img = c.image_size; p.patch = c.patch_size; p.P = (int)img_T; p.u8 = v == 1 && e->in_u8 ? 1 : e->in_u8;
            CK(vc_ln_bwd(e->dt, VC_F32, e->dt, pd, 1, p, cx.L().scr_lnpart, cx.Gf(w.ln1w), cx.Gf(w.ln1b), cx.L().scr_colsum, cx.s));
        }
        }
    }
    return 0;
}

int dec_attn(const Ctx& cx, bool bwd, const void* q, long ldq, const void* k, const void* v, long ldkv, void* o, float* lse,
             int window, const void* dout, void* dq, void* dk, void* dv, long lddq, long lddkv, vc_drop drop) {
    vcad_engine* e = cx.e; const vcad_config& c = e->c;
    AttnParams p; memset(&p, 0, sizeof(p));
    p.q = q; p.k = k; p.v = v; p.o = o; p.ldq = ldq; p.ldk = p.ldv = ldkv; p.ldo = c.hidden_size; p.lse = lse;
    p.B = e->B; p.H = c.nhead; p.Tq = p.Tk = e->T; p.window = window; p.causal = 1; p.drop = drop;
    const int hd = c.hidden_size / c.nhead;
    p.scale = 1.0f / sqrtf((float)hd);
    if (!bwd) return vc_attn_fwd(e->dt, hd, p, cx.s);
    p.dout = dout; p.lddo = c.hidden_size; p.dq = dq; p.dk = dk; p.dv = dv; p.lddq = lddq; p.lddk = p.lddv = lddkv; p.delta = cx.L().t_delta;
    return vc_attn_bwd(e->dt, hd, p, cx.s);
}

// descriptor table of the batched cross-attention K / V projections (vcad_engine::KvForward): per layer kv_c = memory W_kv^T + b_kv
int build_kv_forward(const Ctx& cx) {
    vcad_engine* e = cx.e; const vcad_config& c = e->c;
    const int H = c.hidden_size; const long M = (long)e->B * e->T;
    vcad_engine::KvForward& kf = e->kvf;
    kf.calls.clear(); kf.flops = 0;
    for (int L = 0; L < c.num_decoder_layers; ++L) {
        const DecW& w = e->wd[L]; const DecLayerActs& d = e->da[L];
        const Mat A = cx.A32(e->mem, H), Wm = cx.W(w.ca_w + (long)H * H, H), C = cx.AT(d.kv_c, 2 * H);
        GemmCall gc; memset(&gc, 0, sizeof(gc));
        gc.ct = e->ct; gc.sa = A.dt; gc.sb = Wm.dt; gc.to = C.dt; gc.tra = 0; gc.trb = 0;
        GemmParams& p = gc.p;
        p.A = A.p; p.B = Wm.p; p.C = (void*)C.p; p.M = (int)M; p.N = 2 * H; p.K = H; p.lda = A.ld; p.ldb = Wm.ld; p.ldc = C.ld; p.alpha = 1.0f;
        p.bias = cx.Pf(w.ca_b + H); p.rowadd_div = 1;
        kf.calls.push_back(gc); kf.flops += 2.0 * M * 2 * H * H;
    }
    const int n = (int)kf.calls.size();
    std::vector<GemmParams> probs(n); std::vector<int> tiles(n + 1);
    CK(vc_gemm_grouped_prepare(kf.calls.data(), n, probs.data(), tiles.data()));
    kf.total_tiles = tiles[n];
    CK(vc_upload(kf.d_probs, probs.data(), (size_t)n * sizeof(GemmParams), cx.s));
    CK(vc_upload(kf.d_tiles, tiles.data(), (size_t)(n + 1) * sizeof(int), cx.s));
    kf.ready = true;
    return 0;
}

int engine_forward(vcad_engine* e, float* cmds_out, float* pars_out, vc_stream_t s) {
    Ctx cx{e, s}; const vcad_config& c = e->c;
    const int B = e->B, T = e->T, H = c.hidden_size, D = c.vit_dim; const long M = (long)B * T;
    const size_t es = e->esz;
    if (H != 1024) { vc_set_error("engine: hidden_size %d unsupported (LN kernels: 1024)", H); return VC_ERR_UNSUPPORTED; }
    // wiring (reference model/autoregressive_transformer.py:149-213): past_actions -> tgt = action embeddings, causal self-attention;
    // else past_states -> tgt = UI embeddings; else tgt = memory; in the last two cases self-attention is band-limited too.
    // memory = tanh(image_projection([ui, cad])) only when BOTH flags are set, otherwise tanh(cad embedding) repeated over time.
    const bool pa = c.enable_past_actions, ps = c.enable_past_states;
    const float* ts = c.enable_timestep_embedding ? cx.Pf(e->o_ts) : nullptr;
    CK(cx.refresh_q8());
    // the persistent GEMM's ticket counters are left at zero by every launch; re-zeroed here so that an aborted launch cannot poison the next step
    for (int ln = 0; ln < 2; ++ln) CK(vc_memset_async(e->lane[ln].claim, 0, 64, s));
    // CAD ViT (B images) on the side stream, issued first so that its ~100 small kernels slot in beside the frame ViT's big ones
    const bool fork = ps && ensure_side(e);
    Ctx cxs{e, fork ? e->side : s, 1};
    if (fork) { CK(vc_event_record(e->ev_fork, s)); CK(vc_stream_wait_event(e->side, e->ev_fork)); }
    const int V = c.num_views; const long SS = (long)c.image_size * c.image_size;
    const long ldip = (1 + (ps ? 1 : 0) + (V > 0 ? 1 : 0)) * (long)H;                   // leading dimension of image_projection.weight
    if (V > 0) {
        if (!e->in_mv) { vc_set_error("engine: num_views = %d but no multiview images were given (vcad_set_multiview)", V); return VC_ERR_ARG; }
        if (ps && !pa) { vc_set_error("engine: num_views > 0 with past states but no past actions (the reference's image_projection shapes do not match there either)"); return VC_ERR_UNSUPPORTED; }
        const size_t px = e->in_u8 ? 1 : 4;                                              // (CAD-tower pixels: one gray plane, uint8 or fp32)
        CK(vc_memcpy_d2d_async(e->cadmv, e->in_cad, (size_t)B * SS * px, cxs.s));
        CK(vc_memcpy_d2d_async((char*)e->cadmv + (size_t)B * SS * px, e->in_mv, (size_t)B * V * SS * px, cxs.s));
    }
    CK(vit_forward(cxs, 1, V > 0 ? e->cadmv : e->in_cad, 1, SS));
    if (fork) CK(vc_event_record(e->ev_join, e->side));
    if (ps) {
        CK(vit_forward(cx, 0, e->in_frames, T, e->in_fbstride));
        Epi ep; ep.bias = cx.Pf(e->o_es_b); ep.rowadd = ts; ep.rdiv = T; ep.rmod = 1; ep.ldrow = H; ep.act = VC_ACT_TANH;
        CK(cx.lin_fwd(cx.AT(e->va[0].e, D), cx.W(e->o_es_w, D), cx.A32(e->ui, H), (int)M, H, D, ep));
    }
    if (fork) CK(vc_stream_wait_event(s, e->ev_join));
    { Epi ep; ep.bias = cx.Pf(e->o_ei_b); CK(cx.lin_fwd(cx.AT(e->va[1].e, D), cx.W(e->o_ei_w, D), cx.AT(e->cadE, H), B, H, D, ep)); }
    // multiview: rows B.. of the CAD tower's cls vectors are [B][V * D] (image B + b V + v); embed_multiview -> mvE [B, H]
    if (V > 0) { Epi ep; ep.bias = cx.Pf(e->o_mv_b);
                 CK(cx.lin_fwd(cx.AT((const char*)e->va[1].e + (size_t)B * D * es, (long)V * D), cx.W(e->o_mv_w, (long)V * D), cx.AT(e->mvE, H), B, H, V * D, ep)); }
    if (pa && ps) {
        { Epi ep; ep.bias = cx.Pf(e->o_ip_b);
          CK(cx.lin_fwd(cx.AT(e->cadE, H), cx.W(e->o_ip_w + H, ldip), cx.A32(e->cadterm, H), B, H, H, ep)); }
        if (V > 0) { Epi ep; ep.residual = e->cadterm; ep.ldr = H;                       // cadterm += mvE W_ip[:, 2H:3H]^T
                     CK(cx.lin_fwd(cx.AT(e->mvE, H), cx.W(e->o_ip_w + 2 * H, ldip), cx.A32(e->cadterm, H), B, H, H, ep)); }
        { Epi ep; ep.rowadd = e->cadterm; ep.rdiv = T; ep.rmod = 0; ep.ldrow = H; ep.act = VC_ACT_TANH;
          CK(cx.lin_fwd(cx.A32(e->ui, H), cx.W(e->o_ip_w, ldip), cx.A32(e->mem, H), (int)M, H, H, ep)); }
    } else if (V > 0) {                                                                 // memory = tanh(image_projection([cad, multiview])) repeated over time
        { Epi ep; ep.bias = cx.Pf(e->o_ip_b); CK(cx.lin_fwd(cx.AT(e->cadE, H), cx.W(e->o_ip_w, ldip), cx.A32(e->cadterm, H), B, H, H, ep)); }
        { Epi ep; ep.residual = e->cadterm; ep.ldr = H; CK(cx.lin_fwd(cx.AT(e->mvE, H), cx.W(e->o_ip_w + H, ldip), cx.A32(e->cadterm, H), B, H, H, ep)); }
        CK(vc_bcast_tanh(VC_F32, e->cadterm, e->mem, M, H, T, s));
    } else {
        CK(vc_bcast_tanh(e->dt, e->cadE, e->mem, M, H, T, s));
    }
    const bool h16 = e->dec_h16;
    if (pa) CK(vc_embed_action(h16 ? e->dt : VC_F32, e->in_actions, cx.Pf(e->o_ea_w), cx.Pf(e->o_ea_b), ts, e->act, h16 ? e->tgt0h : nullptr, M, H, c.act_dim, T, s));
    const float* tgt = pa ? e->act : (ps ? e->ui : e->mem);
    if (h16 && !pa) CK(vc_cast(e->dt, tgt, e->tgt0h, M * H, s));
    if (h16) CK(vc_cast(e->dt, e->mem, e->memh, M * H, s));
    const int sa_window = pa ? T : c.window_size;
    const float* x = tgt;
    const void* xh = e->tgt0h;                                 // 16-bit copy of x (h16)
    // every layer's cross-attention K / V projection of the memory in one grouped launch (bf16 mode; the other modes keep the per-layer launches)
    const bool kv_batched = e->dt == VC_BF16 && e->ct == VC_BF16 && vc_gemm_grouped_has_forward();
    if (kv_batched) {
        if (!e->kvf.ready) CK(build_kv_forward(cx));
        CK(vc_gemm_grouped_launch(e->kvf.calls[0], e->kvf.d_probs, e->kvf.d_tiles, (int)e->kvf.calls.size(), e->kvf.total_tiles, e->kvf.flops, s));
        e->kernel_launches[VC_TAG_GEMM_GROUPED]++;
    }
    for (int L = 0; L < c.num_decoder_layers; ++L) {
        const DecW& w = e->wd[L]; DecLayerActs& d = e->da[L];
        { Epi ep; ep.bias = cx.Pf(w.sa_b); CK(cx.lin_fwd(h16 ? cx.AT(xh, H) : cx.A32(x, H), cx.W(w.sa_w, H), cx.AT(d.qkv_s, 3 * H), (int)M, 3 * H, H, ep)); }
        { const char* q = (const char*)d.qkv_s;
          CK(dec_attn(cx, false, q, 3 * H, q + (size_t)H * es, q + (size_t)2 * H * es, 3 * H, d.ao_s, d.lse_s, sa_window, nullptr, nullptr, nullptr, nullptr, 0, 0, cx.site(3, L, Ctx::K_SA))); }
        { Epi ep; ep.bias = cx.Pf(w.sa_ob); ep.residual = x; ep.ldr = H; ep.drop = cx.site(3, L, Ctx::K_SA_OUT); CK(cx.lin_fwd(cx.AT(d.ao_s, H), cx.W(w.sa_ow, H), cx.A32(d.s1, H), (int)M, H, H, ep)); }
        CK(cx.ln_fwd(VC_F32, d.s1, H, w.n1w, w.n1b, d.x1, H, h16 ? d.x1h : nullptr, H, d.st1, M, H));
        { Epi ep; ep.bias = cx.Pf(w.ca_b); CK(cx.lin_fwd(h16 ? cx.AT(d.x1h, H) : cx.A32(d.x1, H), cx.W(w.ca_w, H), cx.AT(d.q_c, H), (int)M, H, H, ep)); }
        if (!kv_batched) { Epi ep; ep.bias = cx.Pf(w.ca_b + H); CK(cx.lin_fwd(cx.A32(e->mem, H), cx.W(w.ca_w + (long)H * H, H), cx.AT(d.kv_c, 2 * H), (int)M, 2 * H, H, ep)); }
        { const char* kv = (const char*)d.kv_c;
          CK(dec_attn(cx, false, d.q_c, H, kv, kv + (size_t)H * es, 2 * H, d.ao_c, d.lse_c, c.window_size, nullptr, nullptr, nullptr, nullptr, 0, 0, cx.site(3, L, Ctx::K_CA))); }
        { Epi ep; ep.bias = cx.Pf(w.ca_ob); ep.residual = d.x1; ep.ldr = H; ep.drop = cx.site(3, L, Ctx::K_CA_OUT); CK(cx.lin_fwd(cx.AT(d.ao_c, H), cx.W(w.ca_ow, H), cx.A32(d.s2, H), (int)M, H, H, ep)); }
        CK(cx.ln_fwd(VC_F32, d.s2, H, w.n2w, w.n2b, d.x2, H, h16 ? d.x2h : nullptr, H, d.st2, M, H));
        { Epi ep; ep.bias = cx.Pf(w.b1); ep.act = VC_ACT_RELU; ep.drop = cx.site(3, L, Ctx::K_FF_ACT); CK(cx.lin_fwd(h16 ? cx.AT(d.x2h, H) : cx.A32(d.x2, H), cx.W(w.w1, H), cx.AT(d.f1, c.dim_feedforward), (int)M, c.dim_feedforward, H, ep)); }
        { Epi ep; ep.bias = cx.Pf(w.b2); ep.residual = d.x2; ep.ldr = H; ep.drop = cx.site(3, L, Ctx::K_FF_OUT);
          CK(cx.lin_fwd(cx.AT(d.f1, c.dim_feedforward), cx.W(w.w2, c.dim_feedforward), cx.A32(d.s3, H), (int)M, H, c.dim_feedforward, ep)); }
        CK(cx.ln_fwd(VC_F32, d.s3, H, w.n3w, w.n3b, d.x3, H, h16 ? d.x3h : nullptr, H, d.st3, M, H));
        x = d.x3; xh = d.x3h;
    }
    e->xfinal = (float*)x;
    const int n5 = c.num_classes, n6 = c.num_params * c.num_params_values;
    const Mat xf = h16 ? cx.AT(xh, H) : cx.A32(x, H);
    { Epi ep; ep.bias = cx.Pf(e->o_h5_b); CK(cx.lin_fwd(xf, cx.W(e->o_h5_w, H), cx.A32(cmds_out, n5), (int)M, n5, H, ep)); }
    { Epi ep; ep.bias = cx.Pf(e->o_h6_b); CK(cx.lin_fwd(xf, cx.W(e->o_h6_w, H), cx.A32(pars_out, n6), (int)M, n6, H, ep)); }
    return 0;
}

// stage 0: heads + decoder (bucket 0).  Leaves the gradient of the decoder input in t_dcur and that of the memory in t_dmem.
// Descriptor tables of the deferred decoder weight gradients (vcad_engine::Deferred): group 0 = activations kept in the
// compute dtype (f1, ao_c, ao_s), group 1 = fp32 residual-stream inputs (x2, x1, memory, layer input).
int build_deferred(const Ctx& cx, const float* tgt0) {
    vcad_engine* e = cx.e; const vcad_config& c = e->c;
    const int H = c.hidden_size, ff = c.dim_feedforward; const long M = (long)e->B * e->T;
    vcad_engine::Deferred& df = e->def;
    df.calls[0].clear(); df.calls[1].clear(); df.cs.clear();
    auto add = [&](int g, Mat dY, Mat X, long w_off, long lddw, long b_off, int N, int K) {
        GemmCall gc; memset(&gc, 0, sizeof(gc));
        gc.ct = e->ct; gc.sa = dY.dt; gc.sb = X.dt; gc.to = VC_F32; gc.tra = 1; gc.trb = 1;
        GemmParams& p = gc.p;
        p.A = dY.p; p.B = X.p; p.C = (void*)cx.Gf(w_off); p.M = N; p.N = K; p.K = (int)M; p.lda = dY.ld; p.ldb = X.ld; p.ldc = lddw; p.alpha = 1.0f;
        df.calls[g].push_back(gc);
        ColsumJob j; memset(&j, 0, sizeof(j));
        j.x = dY.p; j.ld = dY.ld; j.rows = (int)M; j.cols = N; j.out = cx.Gf(b_off); j.is_bf16 = dY.dt == VC_BF16;
        df.cs.push_back(j);
    };
    for (int L = 0; L < c.num_decoder_layers; ++L) {
        const DecW& w = e->wd[L]; const DecLayerActs& d = e->da[L];
        const float* xin = L == 0 ? tgt0 : e->da[L - 1].x3;
        const void* xinh = L == 0 ? e->tgt0h : e->da[L - 1].x3h;
        const bool h16 = e->dec_h16; const int gx = h16 ? 0 : 1;             // with the 16-bit copies every problem has the all-16-bit signature: one group
        add(0, cx.AT(d.g_du_ff, H), cx.AT(d.f1, ff), w.w2, ff, w.b2, H, ff);
        add(gx, cx.AT(d.g_df1, ff), h16 ? cx.AT(d.x2h, H) : cx.A32(d.x2, H), w.w1, H, w.b1, ff, H);
        add(0, cx.AT(d.g_du_ca, H), cx.AT(d.ao_c, H), w.ca_ow, H, w.ca_ob, H, H);
        add(gx, cx.AT(d.g_dq, H), h16 ? cx.AT(d.x1h, H) : cx.A32(d.x1, H), w.ca_w, H, w.ca_b, H, H);
        add(gx, cx.AT(d.g_dkv, 2 * H), h16 ? cx.AT(e->memh, H) : cx.A32(e->mem, H), w.ca_w + (long)H * H, H, w.ca_b + H, 2 * H, H);
        add(0, cx.AT(d.g_du_sa, H), cx.AT(d.ao_s, H), w.sa_ow, H, w.sa_ob, H, H);
        add(gx, cx.AT(d.g_dqkv, 3 * H), h16 ? cx.AT(xinh, H) : cx.A32(xin, H), w.sa_w, H, w.sa_b, 3 * H, H);
        // the three LayerNorm backwards' partial rows [blocks][dgamma | dbeta]: norm{1,2,3}.weight and .bias are adjacent in the flat buffer
        const long nw[3] = {w.n1w, w.n2w, w.n3w}, nb[3] = {w.n1b, w.n2b, w.n3b};
        for (int i = 0; i < 3; ++i) {
            if (nb[i] != nw[i] + H) { vc_set_error("internal: LayerNorm weight / bias not adjacent"); return VC_ERR_ARG; }
            ColsumJob j; memset(&j, 0, sizeof(j));
            j.x = d.ln_part[i]; j.ld = 2L * H; j.rows = (int)vc_ln_bwd_blocks(M); j.cols = 2 * H; j.out = cx.Gf(nw[i]); j.is_bf16 = 0;
            df.cs.push_back(j);
        }
    }
    for (int g = 0; g < 2; ++g) {
        const int n = (int)df.calls[g].size();
        if (n == 0) { df.total_tiles[g] = 0; df.flops[g] = 0; continue; }       // (16-bit copies of the residual stream: every problem is in group 0)
        std::vector<GemmParams> probs(n); std::vector<int> tiles(n + 1);
        CK(vc_gemm_grouped_prepare(df.calls[g].data(), n, probs.data(), tiles.data()));
        df.total_tiles[g] = tiles[n]; df.flops[g] = 0;
        for (const auto& gc : df.calls[g]) df.flops[g] += 2.0 * gc.p.M * gc.p.N * gc.p.K;
        CK(vc_upload(df.d_probs[g], probs.data(), (size_t)n * sizeof(GemmParams), cx.s));
        CK(vc_upload(df.d_tiles[g], tiles.data(), (size_t)(n + 1) * sizeof(int), cx.s));
    }
    int strips = 0; long poff = 0; const int chunks = (int)VC_CEIL_DIV(M, 128);      // (no job has more rows than M)
    for (auto& j : df.cs) { j.strip_start = strips; j.part_off = poff; strips += VC_CEIL_DIV(j.cols, 256); poff += (long)VC_CEIL_DIV(j.rows, 128) * j.cols; }
    df.cs_strips = strips; df.cs_chunks = chunks;
    CK(vc_upload(df.d_cs, df.cs.data(), df.cs.size() * sizeof(ColsumJob), cx.s));
    df.ready = true;
    return 0;
}

// fp16 build: bucket b is complete on `on` — divide the gradient scale out again (exact: a power of two)
int unscale_bucket(vcad_engine* e, int b, vc_stream_t on) {
    if (e->grad_scale == 1.0f) return 0;
    if (e->defer_unscale) { e->grads_scaled_by = e->grad_scale; return 0; }       // (the optimiser divides: vcad_optimizer_step_groups)
    const long lo = e->buckets[b].first, hi = e->buckets[b].second;
    return vc_scale(e->G + lo, e->G + lo, hi - lo, 1.0f / e->grad_scale, on);
}

int backward_stage0(vcad_engine* e, const float* dcmds, const float* dpars, vc_stream_t s) {
    Ctx cx{e, s}; const vcad_config& c = e->c;
    const int B = e->B, T = e->T, H = c.hidden_size, D = c.vit_dim, ff = c.dim_feedforward; const long M = (long)B * T;
    const size_t es = e->esz;
    const int n5 = c.num_classes, n6 = c.num_params * c.num_params_values;
    float* dx = e->t_dcur;
    const float* xf = e->xfinal;
    const bool pa = c.enable_past_actions, ps = c.enable_past_states, tsE = c.enable_timestep_embedding;
    const float* tgt0 = pa ? e->act : (ps ? e->ui : e->mem);
    const int sa_window = pa ? T : c.window_size;
    e->grads_scaled_by = 0.0f;                 // (this backward rewrites every bucket)
    if (e->grad_scale != 1.0f) {
        // (deferred mode: vcad_loss of this forward already wrote scale x dlogits there)
        if (!(e->dls_valid && dcmds == e->dl_cmds && dpars == e->dl_pars)) { CK(vc_scale(dcmds, e->dls_cmds, M * n5, e->grad_scale, s)); CK(vc_scale(dpars, e->dls_pars, M * n6, e->grad_scale, s)); }
        dcmds = e->dls_cmds; dpars = e->dls_pars;
    }
    CK(cx.lin_wgrad(cx.A32(dcmds, n5), cx.A32(xf, H), cx.Gf(e->o_h5_w), H, cx.Gf(e->o_h5_b), (int)M, n5, H));
    CK(cx.lin_wgrad(cx.A32(dpars, n6), cx.A32(xf, H), cx.Gf(e->o_h6_w), H, cx.Gf(e->o_h6_b), (int)M, n6, H));
    CK(cx.lin_dgrad(cx.A32(dpars, n6), cx.W(e->o_h6_w, H), cx.A32(dx, H), (int)M, n6, H, Epi()));
    { Epi ep; ep.residual = dx; ep.ldr = H; CK(cx.lin_dgrad(cx.A32(dcmds, n5), cx.W(e->o_h5_w, H), cx.A32(dx, H), (int)M, n5, H, ep)); }
    // Train mode (every dY below is a private masked / bf16 copy): the 7 weight gradients and bias column sums of each layer are
    // recorded, not launched — see vcad_engine::Deferred.  Eval-mode backward (p = 0: du aliases the live dx) launches them in place.
    const bool defer = e->drop_p > 0.f;
    if (defer && !e->def.ready) CK(build_deferred(cx, tgt0));
    for (int L = c.num_decoder_layers - 1; L >= 0; --L) {
        const DecW& w = e->wd[L]; DecLayerActs& d = e->da[L];
        const float* xin = L == 0 ? tgt0 : e->da[L - 1].x3;
        // ---- FFN   x3 = LN3(x2 + drop(W2 drop(relu(W1 x2 + b1)) + b2))
        Mat du;
        CK(cx.ln_bwd(VC_F32, dx, H, d.s3, H, d.st3, w.n3w, w.n3b, nullptr, 0, dx, H, M, H, cx.site(3, L, Ctx::K_FF_OUT), &du, d.g_du_ff, nullptr, defer ? d.ln_part[2] : nullptr));
        if (!defer) CK(cx.lin_wgrad(du, cx.AT(d.f1, ff), cx.Gf(w.w2), ff, cx.Gf(w.b2), (int)M, H, ff));
        { Epi ep; ep.dact = d.f1; ep.lddact = ff; ep.dkind = VC_ACT_RELU; ep.drop = cx.site(3, L, Ctx::K_FF_ACT);   // f1 > 0 <=> z > 0 and kept
          CK(cx.lin_dgrad(du, cx.W(w.w2, ff), cx.AT(d.g_df1, ff), (int)M, H, ff, ep)); }
        if (!defer) CK(cx.lin_wgrad(cx.AT(d.g_df1, ff), cx.A32(d.x2, H), cx.Gf(w.w1), H, cx.Gf(w.b1), (int)M, ff, H));
        { Epi ep; ep.residual = dx; ep.ldr = H; CK(cx.lin_dgrad(cx.AT(d.g_df1, ff), cx.W(w.w1, H), cx.A32(dx, H), (int)M, ff, H, ep)); }
        // ---- cross attention
        CK(cx.ln_bwd(VC_F32, dx, H, d.s2, H, d.st2, w.n2w, w.n2b, nullptr, 0, dx, H, M, H, cx.site(3, L, Ctx::K_CA_OUT), &du, d.g_du_ca, nullptr, defer ? d.ln_part[1] : nullptr));
        if (!defer) CK(cx.lin_wgrad(du, cx.AT(d.ao_c, H), cx.Gf(w.ca_ow), H, cx.Gf(w.ca_ob), (int)M, H, H));
        CK(cx.lin_dgrad(du, cx.W(w.ca_ow, H), cx.AT(e->t_dao_d, H), (int)M, H, H, Epi()));
        { const char* kv = (const char*)d.kv_c; char* dkv = (char*)d.g_dkv;
          CK(dec_attn(cx, true, d.q_c, H, kv, kv + (size_t)H * es, 2 * H, d.ao_c, d.lse_c, c.window_size, e->t_dao_d, d.g_dq, dkv, dkv + (size_t)H * es, H, 2 * H,
                      cx.site(3, L, Ctx::K_CA))); }
        if (!defer) {
            CK(cx.lin_wgrad(cx.AT(d.g_dq, H), cx.A32(d.x1, H), cx.Gf(w.ca_w), H, cx.Gf(w.ca_b), (int)M, H, H));
            CK(cx.lin_wgrad(cx.AT(d.g_dkv, 2 * H), cx.A32(e->mem, H), cx.Gf(w.ca_w + (long)H * H), H, cx.Gf(w.ca_b + H), (int)M, 2 * H, H));
        }
        { Epi ep; ep.residual = dx; ep.ldr = H; CK(cx.lin_dgrad(cx.AT(d.g_dq, H), cx.W(w.ca_w, H), cx.A32(dx, H), (int)M, H, H, ep)); }
        { Epi ep; if (L != c.num_decoder_layers - 1) { ep.residual = e->t_dmem; ep.ldr = H; }
          CK(cx.lin_dgrad(cx.AT(d.g_dkv, 2 * H), cx.W(w.ca_w + (long)H * H, H), cx.A32(e->t_dmem, H), (int)M, 2 * H, H, ep)); }
        // ---- self attention
        CK(cx.ln_bwd(VC_F32, dx, H, d.s1, H, d.st1, w.n1w, w.n1b, nullptr, 0, dx, H, M, H, cx.site(3, L, Ctx::K_SA_OUT), &du, d.g_du_sa, nullptr, defer ? d.ln_part[0] : nullptr));
        if (!defer) CK(cx.lin_wgrad(du, cx.AT(d.ao_s, H), cx.Gf(w.sa_ow), H, cx.Gf(w.sa_ob), (int)M, H, H));
        CK(cx.lin_dgrad(du, cx.W(w.sa_ow, H), cx.AT(e->t_dao_d, H), (int)M, H, H, Epi()));
        { const char* q = (const char*)d.qkv_s; char* dq = (char*)d.g_dqkv;
          CK(dec_attn(cx, true, q, 3 * H, q + (size_t)H * es, q + (size_t)2 * H * es, 3 * H, d.ao_s, d.lse_s, sa_window, e->t_dao_d,
                      dq, dq + (size_t)H * es, dq + (size_t)2 * H * es, 3 * H, 3 * H, cx.site(3, L, Ctx::K_SA))); }
        if (!defer) CK(cx.lin_wgrad(cx.AT(d.g_dqkv, 3 * H), cx.A32(xin, H), cx.Gf(w.sa_w), H, cx.Gf(w.sa_b), (int)M, 3 * H, H));
        { Epi ep; ep.residual = dx; ep.ldr = H; CK(cx.lin_dgrad(cx.AT(d.g_dqkv, 3 * H), cx.W(w.sa_w, H), cx.A32(dx, H), (int)M, 3 * H, H, ep)); }
    }
    if (defer) {
        // nothing downstream reads these gradients before the optimiser: in the whole-backward entry point they go to the side
        // stream (which vcad_backward joins at the end) and run beside the stem and the ViT backward
        vcad_engine::Deferred& df = e->def;
        vc_stream_t gs = s;
        if (e->bwd_side) { CK(vc_event_record(e->ev_fork2, s)); CK(vc_stream_wait_event(e->side, e->ev_fork2)); gs = e->side; }
        for (int g = 0; g < 2; ++g)
            if (!df.calls[g].empty()) CK(vc_gemm_grouped_launch(df.calls[g][0], df.d_probs[g], df.d_tiles[g], (int)df.calls[g].size(), df.total_tiles[g], df.flops[g], gs));
        CK(vc_colsum_grouped(df.d_cs, (int)df.cs.size(), df.cs_strips, df.cs_chunks, df.cs_partial, gs));
        CK(unscale_bucket(e, 0, gs));
        return 0;
    }
    return unscale_bucket(e, 0, s);
}

// stage 1: stem (bucket 1; reference model/autoregressive_transformer.py:144-178).  dx = gradient of the decoder's tgt input (left in t_dcur by stage 0).
// Leaves d(cls_state) in t_des and d(cls_cad) in t_dec.
int backward_stage_stem(vcad_engine* e, vc_stream_t s) {
    Ctx cx{e, s}; const vcad_config& c = e->c;
    const int B = e->B, T = e->T, H = c.hidden_size, D = c.vit_dim; const long M = (long)B * T;
    float* dx = e->t_dcur;
    const bool pa = c.enable_past_actions, ps = c.enable_past_states, tsE = c.enable_timestep_embedding;
    float* dpre = e->t_dpre;
    if (tsE) CK(vc_memset_async(cx.Gf(e->o_ts), 0, (size_t)c.max_ep_len * H * 4, s));   // rows >= T receive no gradient
    if (pa) {                                                                          // tgt = tanh(embed_action(a) + ts)
        CK(vc_dtanh(VC_F32, dx, e->act, dpre, nullptr, M * H, s));
        CK(cx.lin_wgrad(cx.A32(dpre, H), cx.A32(e->in_actions, c.act_dim), cx.Gf(e->o_ea_w), c.act_dim, cx.Gf(e->o_ea_b), (int)M, H, c.act_dim));
        if (tsE) CK(cx.colsum(cx.A32(dpre, (long)T * H), B, T * H, cx.Gf(e->o_ts), 1));
    } else if (!ps) {
        CK(vc_add_inplace(e->t_dmem, dx, M * H, s));                                   // tgt = memory
    }
    const float* dui = nullptr;                                                        // gradient w.r.t. ui (post-tanh)
    CK(vc_dtanh(VC_F32, e->t_dmem, e->mem, dpre, nullptr, M * H, s));                  // d pre-tanh of the memory
    const int V = c.num_views; const long ldip = (1 + (ps ? 1 : 0) + (V > 0 ? 1 : 0)) * (long)H;
    if (pa && ps) {
        CK(cx.gemm(cx.A32(dpre, H), 1, cx.A32(e->ui, H), 1, Mat{cx.Gf(e->o_ip_w), VC_F32, ldip}, H, H, (int)M, Epi()));
        CK(cx.colsum(cx.A32(dpre, H), T, H, e->t_dcadterm, 0, B, (long)T * H, H));    // sum over t -> [B, H]
        CK(cx.colsum(cx.A32(e->t_dcadterm, H), B, H, cx.Gf(e->o_ip_b), 0));
        CK(cx.gemm(cx.A32(e->t_dcadterm, H), 1, cx.AT(e->cadE, H), 1, Mat{cx.Gf(e->o_ip_w + H), VC_F32, ldip}, H, H, B, Epi()));
        CK(cx.lin_dgrad(cx.A32(e->t_dcadterm, H), cx.W(e->o_ip_w + H, ldip), cx.A32(e->t_dcadE, H), B, H, H, Epi()));
        if (V > 0) {
            CK(cx.gemm(cx.A32(e->t_dcadterm, H), 1, cx.AT(e->mvE, H), 1, Mat{cx.Gf(e->o_ip_w + 2 * H), VC_F32, ldip}, H, H, B, Epi()));
            CK(cx.lin_dgrad(cx.A32(e->t_dcadterm, H), cx.W(e->o_ip_w + 2 * H, ldip), cx.A32(e->t_dmvE, H), B, H, H, Epi()));
        }
        CK(cx.lin_dgrad(cx.A32(dpre, H), cx.W(e->o_ip_w, ldip), cx.A32(e->t_dui, H), (int)M, H, H, Epi()));
        dui = e->t_dui;
    } else if (V > 0) {                                                                // memory = tanh(image_projection([cad, multiview])[b]): sum over t, then through the projection
        CK(cx.colsum(cx.A32(dpre, H), T, H, e->t_dcadterm, 0, B, (long)T * H, H));
        CK(cx.colsum(cx.A32(e->t_dcadterm, H), B, H, cx.Gf(e->o_ip_b), 0));
        CK(cx.gemm(cx.A32(e->t_dcadterm, H), 1, cx.AT(e->cadE, H), 1, Mat{cx.Gf(e->o_ip_w), VC_F32, ldip}, H, H, B, Epi()));
        CK(cx.gemm(cx.A32(e->t_dcadterm, H), 1, cx.AT(e->mvE, H), 1, Mat{cx.Gf(e->o_ip_w + H), VC_F32, ldip}, H, H, B, Epi()));
        CK(cx.lin_dgrad(cx.A32(e->t_dcadterm, H), cx.W(e->o_ip_w, ldip), cx.A32(e->t_dcadE, H), B, H, H, Epi()));
        CK(cx.lin_dgrad(cx.A32(e->t_dcadterm, H), cx.W(e->o_ip_w + H, ldip), cx.A32(e->t_dmvE, H), B, H, H, Epi()));
        if (ps) dui = dx;
    } else {
        CK(cx.colsum(cx.A32(dpre, H), T, H, e->t_dcadE, 0, B, (long)T * H, H));       // memory = tanh(cadE[b]): sum over t
        if (ps) dui = dx;                                                              // tgt = ui
    }
    CK(cx.lin_wgrad(cx.A32(e->t_dcadE, H), cx.AT(e->va[1].e, D), cx.Gf(e->o_ei_w), D, cx.Gf(e->o_ei_b), B, H, D));
    CK(cx.lin_dgrad(cx.A32(e->t_dcadE, H), cx.W(e->o_ei_w, D), cx.A32(e->t_dec, D), B, H, D, Epi()));
    if (V > 0) {      // embed_multiview backward; its input gradient IS rows B.. of d(cls) of the CAD tower
        const char* mvx = (const char*)e->va[1].e + (size_t)B * D * e->esz;
        CK(cx.lin_wgrad(cx.A32(e->t_dmvE, H), cx.AT(mvx, (long)V * D), cx.Gf(e->o_mv_w), (long)V * D, cx.Gf(e->o_mv_b), B, H, V * D));
        CK(cx.lin_dgrad(cx.A32(e->t_dmvE, H), cx.W(e->o_mv_w, (long)V * D), cx.A32(e->t_dec + (long)B * D, (long)V * D), B, H, V * D, Epi()));
    }
    if (ps) {
        CK(vc_dtanh(VC_F32, dui, e->ui, dpre, nullptr, M * H, s));                     // d pre-tanh of the state embedding
        CK(cx.lin_wgrad(cx.A32(dpre, H), cx.AT(e->va[0].e, D), cx.Gf(e->o_es_w), D, cx.Gf(e->o_es_b), (int)M, H, D));
        CK(cx.lin_dgrad(cx.A32(dpre, H), cx.W(e->o_es_w, D), cx.A32(e->t_des, D), (int)M, H, D, Epi()));
        if (tsE) CK(cx.colsum(cx.A32(dpre, (long)T * H), B, T * H, cx.Gf(e->o_ts), 1));
    }
    return 0;
}

}  // namespace

// ===============================================================================================================
// C ABI
// ===============================================================================================================
extern "C" {

const char* vcad_last_error(void) { return vc_get_error(); }
#ifdef VC_EMU
const char* vcad_version(void) { return "videocad_amd 0.2 (host emulator build: tests only)"; }
#else
const char* vcad_version(void) { return "videocad_amd 0.2 (gfx950, " VC_S16_NAME " storage)"; }
#endif

const char* vcad_storage_format(void) { return VC_S16_NAME; }
// Gradient scale (fp16 engines; a power of two, 1 = off): the backward multiplies the incoming dlogits by it and divides every gradient bucket by it again when
// the bucket is complete, so the gradient buffer always holds true gradients.  Changes the workspace plan: call before the next forward.
int vcad_set_grad_scale(vcad_engine* e, float scale) {
    int ex = 0;
    if (scale != 0.0f && (!(scale >= 1.0f) || frexpf(scale, &ex) != 0.5f || scale > 16777216.0f)) { vc_set_error("vcad_set_grad_scale: %g is neither 0 (automatic) nor a power of two in [1, 2^24]", (double)scale); return VC_ERR_ARG; }
    e->grad_scale_auto = scale == 0.0f;
    if (scale != 0.0f) e->grad_scale = scale;
    e->planned_ws = nullptr; e->B = e->T = 0; e->fwd_valid = false; e->infer_T = 0;          // (the plan holds the scaled dlogits copies and, in automatic mode, fixes the value)
    return 0;
}
float vcad_grad_scale(const vcad_engine* e) { return e->grad_scale; }
int vcad_set_defer_unscale(vcad_engine* e, int on) {
    // (turned off while the buffer still holds scaled gradients — a train step that raised between its backward and its optimiser step: the next optimiser step still
    // divides, the next backward rewrites the buffer with true gradients)
    e->defer_unscale = on != 0; e->dls_valid = false;
    return 0;
}

int vcad_engine_create(const vcad_config* cfg, vcad_engine** out) {
    if (!cfg || !out) { vc_set_error("null argument"); return VC_ERR_ARG; }
#ifdef VC_H16
    if (cfg->dtype != VCAD_F16) { vc_set_error("dtype %d: this build of the library runs VCAD_F16 engines (fp16 storage); every other mode is in libvcad_hip.so", cfg->dtype); return VC_ERR_ARG; }
    const bool s16 = cfg->dtype == VCAD_F16;
#else
    if (cfg->dtype == VCAD_F16) { vc_set_error("VCAD_F16: this build of the library stores bf16; the fp16 build is libvcad_hip_f16.so"); return VC_ERR_ARG; }
    if (cfg->dtype != VCAD_F32 && cfg->dtype != VCAD_BF16 && cfg->dtype != VCAD_BF16X3) { vc_set_error("bad dtype %d", cfg->dtype); return VC_ERR_ARG; }
    const bool s16 = cfg->dtype == VCAD_BF16;
#endif
    if (cfg->hidden_size % cfg->nhead) { vc_set_error("hidden_size %% nhead != 0"); return VC_ERR_ARG; }
    const int hd = cfg->hidden_size / cfg->nhead;
    if ((hd != 256 && hd != 128 && hd != 64) || cfg->vit_dim_head != 64) { vc_set_error("head dims (%d, %d) unsupported (decoder 64/128/256, ViT 64)", hd, cfg->vit_dim_head); return VC_ERR_UNSUPPORTED; }
    if (cfg->num_views < 0 || cfg->num_views > 8) { vc_set_error("num_views %d out of range (0..8)", cfg->num_views); return VC_ERR_ARG; }
    if (cfg->num_views > 0 && cfg->enable_past_states && !cfg->enable_past_actions) { vc_set_error("num_views > 0 needs past actions when past states are on (the reference's image_projection fan-in does not match its inputs otherwise)"); return VC_ERR_UNSUPPORTED; }
    if (cfg->window_size < 1) { vc_set_error("window_size must be > 0 (reference model/autoregressive_transformer.py:52)"); return VC_ERR_ARG; }
    // every shape constraint of the kernels is checked HERE, so an unsupported reference config fails at construction (INTEGRATION.md
    // lists which of the reference's model_configs these exclude), never at the first forward
    if (cfg->hidden_size != 1024) { vc_set_error("hidden_size %d unsupported: the LayerNorm / stem kernels are specialised for 1024", cfg->hidden_size); return VC_ERR_UNSUPPORTED; }
    if (cfg->vit_dim != 512 || cfg->patch_size * cfg->patch_size != 1024 || cfg->image_size % cfg->patch_size) {
        vc_set_error("ViT dims (dim=%d, patch=%d, image=%d) unsupported: LayerNorm kernels need dim 512 and 32x32 patches", cfg->vit_dim, cfg->patch_size, cfg->image_size); return VC_ERR_UNSUPPORTED; }
    { const int g = cfg->image_size / cfg->patch_size; if (g * g + 1 > 64) { vc_set_error("ViT token count %d exceeds the 64-token attention tile", g * g + 1); return VC_ERR_UNSUPPORTED; } }
    if (cfg->num_classes != 5 || cfg->num_params != 6 || cfg->num_params_values != 1000) { vc_set_error("heads must be 5 + 6x1000 (reference model/autoregressive_transformer.py:218)"); return VC_ERR_UNSUPPORTED; }
    if (cfg->dim_feedforward % 8 || cfg->vit_mlp % 8) { vc_set_error("dim_feedforward / vit_mlp must be multiples of 8 (16-byte bf16 rows)"); return VC_ERR_UNSUPPORTED; }
    // 16-bit engines run the ViT MLP's activation derivative as a pass that also reduces the first Linear's bias gradient (partial rows of 256 / (vit_mlp / 8)
    // token rows per workgroup) and defer that reduction: widths the pass has no row blocks for are refused here, not with a fault in the first backward
    // (vit_mlp >= 2056 divided by zero in the workspace plan, 768 / 1536 / 3072 failed in the backward).  The reference hardcodes mlp_dim = 512.
    if (s16 && !vc_dact_bwd_fused_ok(cfg->vit_mlp)) {
        vc_set_error("vit_mlp %d unsupported by the 16-bit engines: vit_mlp / 8 must divide 256 (64, 128, 256, 512, 1024, 2048; the reference uses 512)", cfg->vit_mlp); return VC_ERR_UNSUPPORTED; }
    vcad_engine* e = new vcad_engine();
    e->c = *cfg; e->dt = s16 ? VC_BF16 : VC_F32; e->ct = cfg->dtype == VCAD_BF16X3 ? VC_X3 : e->dt; e->esz = s16 ? 2 : 4;
#ifdef VC_H16
    if (s16) { e->grad_scale = 4096.0f; e->grad_scale_auto = true; }          // fp16 engines: automatic gradient scale (vcad_set_grad_scale), 4096 until the first plan
#endif
    if (cfg->vit_depth < 1 || cfg->num_decoder_layers < 1) { vc_set_error("vit_depth / num_decoder_layers must be >= 1"); delete e; return VC_ERR_ARG; }
    build_params(e);
    if ((int)e->buckets.size() != NB_BUCKETS) { vc_set_error("internal: %d buckets", (int)e->buckets.size()); delete e; return VC_ERR_ARG; }
    *out = e;
    return 0;
}
void vcad_engine_destroy(vcad_engine* e) {
    if (e && e->side_ok) { vc_stream_destroy(e->side); vc_event_destroy(e->ev_fork); vc_event_destroy(e->ev_fork2); vc_event_destroy(e->ev_join); }
    delete e;
}

int64_t vcad_param_total(const vcad_engine* e) { return e->ptotal; }
int vcad_param_count(const vcad_engine* e) { return (int)e->plist.size(); }
int vcad_param_info(const vcad_engine* e, int i, char* name, size_t cap, int64_t* off, int64_t* numel, int64_t shape[4], int* ndim) {
    if (i < 0 || i >= (int)e->plist.size()) { vc_set_error("param index %d out of range", i); return VC_ERR_ARG; }
    const PInfo& p = e->plist[i];
    if (name && cap) { strncpy(name, p.name.c_str(), cap - 1); name[cap - 1] = 0; }
    if (off) *off = p.off; if (numel) *numel = p.numel; if (ndim) *ndim = p.ndim;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = p.shape[k];
    return 0;
}
int vcad_bucket_count(const vcad_engine* e) { return (int)e->buckets.size(); }
int vcad_bucket_range(const vcad_engine* e, int b, int64_t* begin, int64_t* end) {
    if (b < 0 || b >= (int)e->buckets.size()) { vc_set_error("bucket %d out of range", b); return VC_ERR_ARG; }
    *begin = e->buckets[b].first; *end = e->buckets[b].second; return 0;
}
int vcad_bind(vcad_engine* e, float* params, float* grads, float* m, float* v, void* shadow) {
    if (!params) { vc_set_error("vcad_bind: params is null"); return VC_ERR_ARG; }
    if (e->dt == VC_BF16 && !shadow) { vc_set_error("vcad_bind: bf16 engine needs a shadow buffer"); return VC_ERR_ARG; }
    e->P = params; e->G = grads; e->Mm = m; e->Vv = v; e->S = e->dt == VC_BF16 ? (vc_bf16*)shadow : nullptr;
    e->Spk = e->ct == VC_X3 ? (uint32_t*)shadow : nullptr;      // optional: without it the bf16x3 GEMMs split the fp32 weights while staging
    e->wT_fresh = false; e->pe_fresh[0] = e->pe_fresh[1] = false; e->q8_fresh = false; e->def.ready = false; e->kvf.ready = false;
    for (int v = 0; v < 2; ++v) for (int part = 0; part < 3; ++part) e->vcs[v][part].ready = false;
    for (int v = 0; v < 2; ++v) e->cls[v].bwd_ready = false;
    return 0;
}
// No CPU fallback: an engine whose buffers still live in host memory (a model built on "cpu" and not yet moved) is refused by every entry point that
// would launch kernels over them — with an error, not a memory fault.  (vc_rt.h: the emulator build's "device" is host memory.)
static int need_device(const vcad_engine* e, const char* what) {
    if (e->P && !vc_is_device_ptr(e->P)) {
        vc_set_error("%s: the bound parameter buffer is not device memory — this library has no CPU fallback: move the model to a ROCm device (model.to('cuda'))", what);
        return VC_ERR_ARG;
    }
    return 0;
}
int vcad_sync_shadow(vcad_engine* e, void* stream) {
    CK(need_device(e, "vcad_sync_shadow"));
    if (e->Spk) return vc_pack_x3(e->P, e->Spk, e->ptotal, (vc_stream_t)stream);
    if (e->dt != VC_BF16) return 0;
    if (!e->P || !e->S) { vc_set_error("vcad_sync_shadow: not bound"); return VC_ERR_ARG; }
    e->wT_fresh = false; e->pe_fresh[0] = e->pe_fresh[1] = false; e->q8_fresh = false;
    return vc_cast(VC_BF16, e->P, e->S, e->ptotal, (vc_stream_t)stream);
}
size_t vcad_workspace_bytes(const vcad_engine* e, int B, int T) {
    vcad_engine tmp = *e;
    return plan(&tmp, B, T, nullptr);
}
int vcad_set_workspace(vcad_engine* e, void* ws, size_t bytes) { e->ws = (char*)ws; e->ws_bytes = bytes; e->fwd_valid = false; e->B = e->T = 0; e->planned_ws = nullptr; e->infer_T = 0; return 0; }

// VCAD_FP8 forward mode: the ViT's Linear layers on the block-scaled fp8 matrix cores (bf16 engines only; backward unchanged).  Changes the
// workspace plan: call before the next forward.
int vcad_set_fp8(vcad_engine* e, int on) {
#ifdef VC_H16
    if (on) { vc_set_error("vcad_set_fp8: the fp8 forward mode exists in the bf16 build only"); return VC_ERR_UNSUPPORTED; }
#endif
    if (on && e->dt != VC_BF16) { vc_set_error("vcad_set_fp8: fp8 forward GEMMs need a bf16 engine"); return VC_ERR_UNSUPPORTED; }
    if (on && (e->c.vit_dim % 128 || (e->c.vit_heads * e->c.vit_dim_head) % 128 || e->c.vit_mlp % 128)) { vc_set_error("vcad_set_fp8: ViT widths must be multiples of 128"); return VC_ERR_UNSUPPORTED; }
    e->fp8 = on != 0; e->planned_ws = nullptr; e->B = e->T = 0; e->fwd_valid = false; e->infer_T = 0;
    return 0;
}
// kernel-selection flags of this engine's Linear layers (VCAD_GEMM_* in vcad.h; 0 = automatic) and launch counts per kernel family
int vcad_set_gemm_flags(vcad_engine* e, uint32_t flags) { e->gemm_flags = flags; return 0; }
int64_t vcad_kernel_launches(const vcad_engine* e, int family) { return (family >= 0 && family < VC_NTAG) ? e->kernel_launches[family] : -1; }
int vcad_set_side_stream(vcad_engine* e, int on) { e->no_side = !on; return 0; }
// multiview images of the NEXT forward: [B][num_views] gray planes in the CAD image's pixel format (fp32 normalised, or uint8 when the forward is
// vcad_forward_u8 / _rgb8), contiguous; device pointer, read during the forward only (the engine stages its own copy for the backward)
int vcad_set_multiview(vcad_engine* e, const void* images) {
    if (e->c.num_views <= 0 && images) { vc_set_error("vcad_set_multiview: engine was created with num_views = 0"); return VC_ERR_ARG; }
    if (images && ((uintptr_t)images & 3)) { vc_set_error("vcad_set_multiview: pointer must be 4-byte aligned"); return VC_ERR_ARG; }
    e->in_mv = images; return 0;
}
int vcad_set_dropout(vcad_engine* e, float p, uint64_t seed) {
    if (!(p >= 0.f && p < 1.f)) { vc_set_error("vcad_set_dropout: p must be in [0, 1)"); return VC_ERR_ARG; }
    e->drop_p = p; e->drop_seed = seed;
    return 0;
}
// debug / test hook: the keep-multipliers (0 or 1/(1-p)) a site applies to elements 0..n-1, written to a HOST buffer
int vcad_dropout_mask(const vcad_engine* e, int module, int layer, int kind, int64_t n, float* host_out) {
    Ctx cx{const_cast<vcad_engine*>(e), nullptr};
    const vc_drop d = cx.site(module, layer, kind);
    for (int64_t i = 0; i < n; ++i) host_out[i] = d.key ? vc_drop_mul(d, (uint32_t)i) : 1.0f;
    return 0;
}

// ... elements first .. first + n - 1 of the site's index space (r06: the masks of a few clips of a benchmark-sized batch without materialising 50 M floats per site)
int vcad_dropout_mask_range(const vcad_engine* e, int module, int layer, int kind, int64_t first, int64_t n, float* host_out) {
    if (first < 0 || n < 0 || first + n > 4294967296LL) { vc_set_error("vcad_dropout_mask_range: [%lld, +%lld) outside the 32-bit index space", (long long)first, (long long)n); return VC_ERR_ARG; }
    Ctx cx{const_cast<vcad_engine*>(e), nullptr};
    const vc_drop d = cx.site(module, layer, kind);
    for (int64_t i = 0; i < n; ++i) host_out[i] = d.key ? vc_drop_mul(d, (uint32_t)(first + i)) : 1.0f;
    return 0;
}

static int forward_any(vcad_engine* e, const void* frames, int64_t fbstride, const float* actions, const void* cad, int u8, int B, int T,
                       float* cmds_out, float* pars_out, void* stream);
int vcad_forward(vcad_engine* e, const float* frames, int64_t fbstride, const float* actions, const float* cad, int B, int T,
                 float* cmds_out, float* pars_out, void* stream) {
    return forward_any(e, frames, fbstride, actions, cad, 0, B, T, cmds_out, pars_out, stream);
}
int vcad_forward_u8(vcad_engine* e, const uint8_t* frames, int64_t fbstride, const float* actions, const uint8_t* cad, int B, int T,
                    float* cmds_out, float* pars_out, void* stream) {
    if (((uintptr_t)frames | (uintptr_t)cad | (uintptr_t)fbstride) & 3) { vc_set_error("vcad_forward_u8: frames / cad / batch stride must be 4-byte aligned"); return VC_ERR_ARG; }
    return forward_any(e, frames, fbstride, actions, cad, 1, B, T, cmds_out, pars_out, stream);
}
int vcad_forward_rgb8(vcad_engine* e, const uint8_t* frames_rgb, int64_t fbstride, const float* actions, const uint8_t* cad, int B, int T,
                      float* cmds_out, float* pars_out, void* stream) {
    if (((uintptr_t)frames_rgb | (uintptr_t)cad) & 3 || (fbstride & 3)) { vc_set_error("vcad_forward_rgb8: frames / cad must be 4-byte aligned, the batch stride a multiple of 4 pixels"); return VC_ERR_ARG; }
    return forward_any(e, frames_rgb, fbstride, actions, cad, 2, B, T, cmds_out, pars_out, stream);
}
static int forward_any(vcad_engine* e, const void* frames, int64_t fbstride, const float* actions, const void* cad, int u8, int B, int T,
                       float* cmds_out, float* pars_out, void* stream) {
    if (!e->P) { vc_set_error("vcad_forward: parameters not bound"); return VC_ERR_ARG; }
    CK(need_device(e, "vcad_forward"));
    if (B < 1 || T < 1 || T > e->c.max_ep_len) { vc_set_error("vcad_forward: bad B=%d T=%d", B, T); return VC_ERR_ARG; }
    // any horizon up to max_ep_len (reference: 1 000): bf16 mode on the block-streaming decoder attention (attn_mfma.h), the fp32 / bf16x3 modes on the
    // wave-per-row kernels (attn.h: up to sixteen 64-key pieces per query)
    if (T > 1024) { vc_set_error("vcad_forward: T=%d exceeds the attention kernels' 1024-key limit", T); return VC_ERR_UNSUPPORTED; }
    if ((double)B * T * 50.0 * 3072.0 >= 4294967296.0) { vc_set_error("vcad_forward: B*T=%d too large for 32-bit dropout indices", B * T); return VC_ERR_UNSUPPORTED; }
    if (!e->ws) { vc_set_error("vcad_forward: no workspace"); return VC_ERR_WORKSPACE; }
    if (B != e->B || T != e->T || e->planned_ws != e->ws) {     // same (B, T, workspace): pointers, deferred tables and W^T jobs stay valid
        { vcad_engine tmp = *e; const size_t need = plan(&tmp, B, T, nullptr);
          if (need > e->ws_bytes) { vc_set_error("vcad_forward: workspace %zu < %zu bytes", e->ws_bytes, need); return VC_ERR_WORKSPACE; } }
        plan(e, B, T, e->ws);
        e->planned_ws = e->ws; e->infer_T = 0;
    }
    e->B = B; e->T = T; e->in_frames = frames; e->in_fbstride = fbstride; e->in_actions = actions; e->in_cad = cad; e->in_u8 = u8;
    e->fwd_valid = false; e->dls_valid = false;
    int rc = engine_forward(e, cmds_out, pars_out, (vc_stream_t)stream);
    if (rc) return rc;
    if (vc_last_launch_error()) { vc_set_error("vcad_forward: kernel launch failed"); return VC_ERR_LAUNCH; }
    e->fwd_valid = true;
    return 0;
}

static void fill_loss_params(vcad_engine* e, LossParams& p, const float* cmds, const float* pars, const float* targets, int B, int T,
                             int use_mse, const float* label_w_host, const float* class_w) {
    memset(&p, 0, sizeof(p));
    const vcad_config& c = e->c; const long M = (long)B * T;
    p.cmds = cmds; p.ldc = c.num_classes; p.pars = pars; p.ldp = (long)c.num_params * c.num_params_values; p.targets = targets;
    p.M = M; p.T = T; p.use_mse = use_mse;
    const int tol[6] = {2, 2, 50, 200, 500, 2};            // reference trainer.py:827 (TOLERANCE = 3)
    const int above[6] = {0, 0, 1, 1, 1, 0};               // reference trainer.py:829 (metrics only; the loss window is always one-sided)
    const int p2l[6] = {0, 0, 1, 1, 2, 3};                 // reference trainer.py:825
    for (int i = 0; i < 6; ++i) { p.tol[i] = tol[i]; p.above[i] = above[i]; p.param_to_label[i] = p2l[i]; }
    for (int i = 0; i < 5; ++i) p.label_w[i] = label_w_host[i];      // class_weights.json "Label", read by the caller at run time (reference trainer.py:822-825)
    p.class_w = class_w;
    p.row_num = e->loss_rows; p.row_den = e->loss_rows + M * 7; p.row_lse = e->loss_rows + 2 * M * 7; p.row_arg = e->loss_arg;
    p.loss_out = e->loss_small; p.scales = e->loss_small + 16; p.metrics = e->loss_metrics;
    p.dcmds = e->dl_cmds; p.lddc = c.num_classes; p.dpars = e->dl_pars; p.lddp = p.ldp; p.gmul = 1.0f;
    e->dls_valid = false;
    if (e->defer_unscale && e->grad_scale != 1.0f && e->dls_cmds) {      // the dlogits pass writes scale x dlogits where the backward looks for them (dl_* stay stale: native train step only)
        p.dcmds = e->dls_cmds; p.dpars = e->dls_pars; p.gmul = e->grad_scale; e->dls_valid = true;
    }
}

int vcad_loss(vcad_engine* e, const float* cmds, const float* pars, const float* targets, int B, int T, int use_mse,
              const float* label_weights_host, const float* class_w, float* loss_out, int32_t* metrics_out, void* stream) {
    if (!e->ws || B != e->B || T != e->T) { vc_set_error("vcad_loss: call vcad_forward with the same (B,T) first"); return VC_ERR_ARG; }
    if (e->c.num_classes != 5 || e->c.num_params != 6 || e->c.num_params_values != 1000) { vc_set_error("vcad_loss: heads must be 5 + 6x1000 (reference autoregressive_transformer.py:218)"); return VC_ERR_UNSUPPORTED; }
    if (!use_mse && !class_w) { vc_set_error("vcad_loss: use_mse=0 needs class_weights"); return VC_ERR_ARG; }
    if (!label_weights_host) { vc_set_error("vcad_loss: label_weights (class_weights.json \"Label\", 5 host floats) is required"); return VC_ERR_ARG; }
    LossParams p; fill_loss_params(e, p, cmds, pars, targets, B, T, use_mse, label_weights_host, class_w);
    vc_stream_t s = (vc_stream_t)stream;
    CK(vc_loss_fwd(p, s));
    CK(vc_loss_bwd(p, s));
    if (loss_out) CK(vc_memcpy_d2d_async(loss_out, e->loss_small, 8 * 4, s));
    if (metrics_out) CK(vc_memcpy_d2d_async(metrics_out, e->loss_metrics, VC_NMETRIC * 4, s));
    return 0;
}

int vcad_dlogits_offsets(const vcad_engine* e, size_t* off_cmds, size_t* off_params) {
    if (!e->ws || !e->B) { vc_set_error("vcad_dlogits_offsets: no forward yet"); return VC_ERR_ARG; }
    *off_cmds = (size_t)((const char*)e->dl_cmds - e->ws); *off_params = (size_t)((const char*)e->dl_pars - e->ws);
    return 0;
}

int vcad_backward_stage(vcad_engine* e, int stage, const float* dcmds, const float* dpars, void* stream) {
    if (!e->fwd_valid) { vc_set_error("vcad_backward: no forward activations (call vcad_forward first)"); return VC_ERR_ARG; }
    if (!e->G) { vc_set_error("vcad_backward: gradient buffer not bound"); return VC_ERR_ARG; }
    if ((dcmds == nullptr) != (dpars == nullptr)) { vc_set_error("vcad_backward: dcmds/dparams must both be given or both be NULL"); return VC_ERR_ARG; }
    vc_stream_t s = (vc_stream_t)stream; Ctx cx{e, s};
    const long img2 = (long)e->c.image_size * e->c.image_size;
    int rc = 0;
    switch (stage) {
        case 0: rc = backward_stage0(e, dcmds ? dcmds : e->dl_cmds, dpars ? dpars : e->dl_pars, s); break;
        case 1: rc = backward_stage_stem(e, s); break;
        case CAD_STAGE: { Ctx c1{e, e->bwd_fork ? e->side : s, 1}; rc = vit_backward(c1, 1, e->t_dec, 0, e->c.num_views > 0 ? e->cadmv : e->in_cad, 1, img2); } break;
        case 3: rc = e->c.enable_past_states ? vit_backward(cx, 0, e->t_des, 1, e->in_frames, e->T, e->in_fbstride) : 0; break;
        case 4: rc = e->c.enable_past_states ? vit_backward(cx, 0, e->t_des, 2, e->in_frames, e->T, e->in_fbstride) : 0; break;
        default: vc_set_error("vcad_backward_stage: stage %d out of range", stage); return VC_ERR_ARG;
    }
    if (rc) return rc;
    if (stage > 0) CK(unscale_bucket(e, stage, stage == CAD_STAGE && e->bwd_fork ? e->side : s));       // (stage 0 does its own: its last launches may be on the side stream)
    if (vc_last_launch_error()) { vc_set_error("vcad_backward: kernel launch failed"); return VC_ERR_LAUNCH; }
    return 0;
}
int vcad_side_stage(const vcad_engine*) { return CAD_STAGE; }
// Data-parallel callers: the CAD ViT's backward (stage vcad_side_stage()) launched on the side stream; its bucket may be all-reduced only after
// vcad_join_side() has made the waiting stream wait for it.  When the side stream is not used (enable_past_states off, side stream disabled, profiler
// recording, stream creation failed) the stage runs on the caller's stream in line and vcad_join_side() is a no-op: a communication stream must then be ordered
// behind the caller's stream itself (videocad_amd/trainer.py: GradSync.reduce_range waits for both).
int vcad_backward_stage_side(vcad_engine* e, int stage, const float* dcmds, const float* dpars, void* stream) {
    if (stage != CAD_STAGE) { vc_set_error("vcad_backward_stage_side: only stage %d (CAD ViT) can run on the side stream", CAD_STAGE); return VC_ERR_ARG; }
    vc_stream_t s = (vc_stream_t)stream;
    const bool fork = e->c.enable_past_states && ensure_side(e);
    if (fork) { CK(vc_event_record(e->ev_fork, s)); CK(vc_stream_wait_event(e->side, e->ev_fork)); }
    e->bwd_fork = fork;
    int rc = vcad_backward_stage(e, CAD_STAGE, dcmds, dpars, stream);
    e->bwd_fork = false;
    if (rc) return rc;
    if (fork) { CK(vc_event_record(e->ev_join, e->side)); e->side_pending = true; }
    return 0;
}
int vcad_join_side(vcad_engine* e, void* stream) {
    if (e->side_pending) { CK(vc_stream_wait_event((vc_stream_t)stream, e->ev_join)); e->side_pending = false; }
    return 0;
}

// ---- gradient wire format of the data-parallel exchange (norm.h: wire_*_kernel; trainer.py: GradSync grad_wire = "half")
static int wire_range(const vcad_engine* e, int64_t lo, int64_t hi, const char* what) {
    if (!e->G) { vc_set_error("%s: buffers not bound", what); return VC_ERR_ARG; }
    if (lo < 0 || hi < lo || hi > e->ptotal || (lo & 3)) { vc_set_error("%s: range [%lld, %lld) outside the gradient buffer or not 4-float aligned", what, (long long)lo, (long long)hi); return VC_ERR_ARG; }
    return 0;
}
int vcad_wire_amax(vcad_engine* e, int64_t lo, int64_t hi, float* amax_out, void* stream) {
    CK(wire_range(e, lo, hi, "vcad_wire_amax"));
    if (!amax_out) { vc_set_error("vcad_wire_amax: amax_out is null"); return VC_ERR_ARG; }
    return vc_wire_amax(e->G + lo, hi - lo, amax_out, (vc_stream_t)stream);
}
int vcad_wire_pack(vcad_engine* e, int64_t lo, int64_t hi, void* wire, const float* amax, int world, void* stream) {
    CK(wire_range(e, lo, hi, "vcad_wire_pack"));
    return vc_wire_pack(e->G + lo, wire, hi - lo, amax, world < 1 ? 1 : world, (vc_stream_t)stream);
}
int vcad_wire_unpack(vcad_engine* e, int64_t lo, int64_t hi, const void* wire, const float* amax, int world, void* stream) {
    CK(wire_range(e, lo, hi, "vcad_wire_unpack"));
    return vc_wire_unpack(wire, e->G + lo, hi - lo, amax, world < 1 ? 1 : world, (vc_stream_t)stream);
}

// Whole backward: after stages 0-1 (heads + decoder, stem) the CAD ViT's backward (stage 2) is independent of the frame ViT's
// (stages 3-4), so it runs on the side stream beside them.  The staged entry point (data-parallel callers all-reduce a bucket as
// soon as its stage returns) keeps everything on the caller's stream.
int vcad_set_bucket_callback(vcad_engine* e, vcad_bucket_ready_fn fn, void* user) { e->bucket_cb = fn; e->bucket_cb_user = user; return 0; }

int vcad_backward(vcad_engine* e, const float* dcmds, const float* dpars, void* stream) {
    vc_stream_t s = (vc_stream_t)stream;
    const bool fork = e->c.enable_past_states && ensure_side(e);
    // per-bucket hook (vcad_set_bucket_callback): called right after the launches that finalise a bucket have been enqueued, with the stream they are on
    auto ready = [&](int b, vc_stream_t on) -> int {
        if (!e->bucket_cb) return 0;
        const long lo = e->buckets[b].first, hi = e->buckets[b].second;
        const int rc = e->bucket_cb(e->bucket_cb_user, b, e->G + lo, (int64_t)(hi - lo), (void*)on);
        if (rc) { vc_set_error("vcad_backward: bucket callback returned %d for bucket %d", rc, b); return VC_ERR_ARG; }
        return 0;
    };
    e->bwd_side = fork;
    int rc0 = vcad_backward_stage(e, 0, dcmds, dpars, stream);
    e->bwd_side = false;
    if (rc0) return rc0;
    CK(ready(0, fork && e->drop_p > 0.f ? e->side : s));       // (train mode: the deferred decoder weight gradients were enqueued on the side stream, behind stage 0)
    CK(vcad_backward_stage(e, 1, dcmds, dpars, stream));
    CK(ready(1, s));
    if (fork) { CK(vc_event_record(e->ev_fork, s)); CK(vc_stream_wait_event(e->side, e->ev_fork)); }
    // r06: the CAD tower's ~180 launches are 5-20 us kernels — the GPU retires them as fast as the host enqueues them — and they used to be enqueued BEFORE the
    // frame tower's first backward stage: the caller's stream sat idle (1.2-1.4 ms in the kernel trace) until the host got there.  With the side stream forked
    // (and no per-bucket hook that expects stage order) the frame tower's upper stage is enqueued first: its 70-340 us kernels keep the GPU busy while the host
    // feeds the side stream.  Same kernels, same streams, same events — only the host's enqueue order changes.
    const bool frame_first = fork && !e->bucket_cb && NB_BUCKETS > CAD_STAGE + 1 && g_frame_first;
    if (frame_first) CK(vcad_backward_stage(e, CAD_STAGE + 1, dcmds, dpars, stream));
    e->bwd_fork = fork;
    int rc = vcad_backward_stage(e, CAD_STAGE, dcmds, dpars, stream);
    e->bwd_fork = false;
    if (rc) return rc;
    CK(ready(CAD_STAGE, fork ? e->side : s));
    if (fork) CK(vc_event_record(e->ev_join, e->side));
    for (int st = CAD_STAGE + 1 + (frame_first ? 1 : 0); st < NB_BUCKETS; ++st) { CK(vcad_backward_stage(e, st, dcmds, dpars, stream)); CK(ready(st, s)); }
    if (fork) CK(vc_stream_wait_event(s, e->ev_join));
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// Incremental inference (reference model/autoregressive_transformer.py:222-275 `sequential_inference`).
// The reference re-runs the whole forward on the prefix [0..t] for every step t: t+1 ViT passes and a full decoder pass per step,
// O(T^2) frames encoded per clip.  The model is causal everywhere (causal / banded self-attention, banded cross-attention,
// post-norm layers, per-frame ViT), so position t's activations never change once computed: each step here encodes ONE new frame
// per clip, projects ONE row through every decoder layer and attends to the cached keys / values of the earlier steps.
// ---------------------------------------------------------------------------------------------------------------
static size_t infer_plan(vcad_engine* e, int B, int Tmax, char* base) {
    const vcad_config& c = e->c;
    size_t off = plan(e, B, 1, base);                     // one frame per clip in flight: the (B, T = 1) activation plan
    e->ic_kv_s.assign(c.num_decoder_layers, nullptr); e->ic_kv_c.assign(c.num_decoder_layers, nullptr);
    const size_t per = (size_t)B * Tmax * 2 * c.hidden_size * e->esz;
    for (int L = 0; L < c.num_decoder_layers; ++L) {
        off = (off + 255) & ~(size_t)255; e->ic_kv_s[L] = base + off; off += per;
        off = (off + 255) & ~(size_t)255; e->ic_kv_c[L] = base + off; off += per;
    }
    return off + 256;
}

size_t vcad_infer_workspace_bytes(const vcad_engine* e, int B, int Tmax) {
    vcad_engine tmp = *e;
    return infer_plan(&tmp, B, Tmax, nullptr);
}

static int infer_begin_any(vcad_engine* e, const void* cad, int u8, int B, int Tmax, void* stream) {
    if (!e->P) { vc_set_error("vcad_infer_begin: parameters not bound"); return VC_ERR_ARG; }
    CK(need_device(e, "vcad_infer_begin"));
    if (e->c.num_views > 0) { vc_set_error("vcad_infer_begin: incremental inference has no multiview input (nor has the reference's sequential_inference)"); return VC_ERR_UNSUPPORTED; }
    if (B < 1 || Tmax < 1 || Tmax > 1024 || Tmax > e->c.max_ep_len) { vc_set_error("vcad_infer_begin: bad B=%d Tmax=%d (1 <= Tmax <= min(1024, max_ep_len))", B, Tmax); return VC_ERR_ARG; }
    if (!e->ws) { vc_set_error("vcad_infer_begin: no workspace"); return VC_ERR_WORKSPACE; }
    { vcad_engine tmp = *e; const size_t need = infer_plan(&tmp, B, Tmax, nullptr);
      if (need > e->ws_bytes) { vc_set_error("vcad_infer_begin: workspace %zu < %zu bytes", e->ws_bytes, need); return VC_ERR_WORKSPACE; } }
    infer_plan(e, B, Tmax, e->ws);
    e->B = B; e->T = 1; e->planned_ws = e->ws; e->fwd_valid = false;
    e->infer_T = Tmax; e->infer_t = 0; e->infer_u8 = u8; e->in_u8 = u8; e->in_cad = cad;
    const float keep_p = e->drop_p; e->drop_p = 0.f;       // inference = model.eval()
    vc_stream_t s = (vc_stream_t)stream; Ctx cx{e, s}; const vcad_config& c = e->c;
    const int H = c.hidden_size, D = c.vit_dim;
    int rc = cx.refresh_q8();                                 // (VCAD_FP8: quantised weight copies, no-op otherwise)
    // the re-plan moved the persistent GEMM's ticket counters (Lane::claim) onto workspace bytes that held activations: zero them before any
    // ticket-drawn launch (VCAD_GEMM_DYNAMIC stays on after data-parallel training) — every launch leaves them at zero again
    for (int ln = 0; ln < 2 && !rc; ++ln) rc = vc_memset_async(e->lane[ln].claim, 0, 64 * sizeof(int), s);
    if (!rc) rc = vit_forward(cx, 1, cad, 1, (long)c.image_size * c.image_size);
    if (!rc) { Epi ep; ep.bias = cx.Pf(e->o_ei_b); rc = cx.lin_fwd(cx.AT(e->va[1].e, D), cx.W(e->o_ei_w, D), cx.AT(e->cadE, H), B, H, D, ep); }
    if (!rc && c.enable_past_actions && c.enable_past_states) {
        Epi ep; ep.bias = cx.Pf(e->o_ip_b);
        rc = cx.lin_fwd(cx.AT(e->cadE, H), cx.W(e->o_ip_w + H, 2 * H), cx.A32(e->cadterm, H), B, H, H, ep);
    }
    e->drop_p = keep_p;
    if (rc) return rc;
    if (vc_last_launch_error()) { vc_set_error("vcad_infer_begin: kernel launch failed"); return VC_ERR_LAUNCH; }
    return 0;
}
int vcad_infer_begin(vcad_engine* e, const float* cad, int B, int Tmax, void* stream) { return infer_begin_any(e, cad, 0, B, Tmax, stream); }
int vcad_infer_begin_u8(vcad_engine* e, const uint8_t* cad, int B, int Tmax, void* stream) {
    if ((uintptr_t)cad & 3) { vc_set_error("vcad_infer_begin_u8: cad must be 4-byte aligned"); return VC_ERR_ARG; }
    return infer_begin_any(e, cad, 1, B, Tmax, stream);
}

// one cached attention: the single query row of step t against rows [lo(t) .. t] of this layer's cache
static int infer_attn(const Ctx& cx, const void* q, void* kv_cache, void* o, float* lse, int t, int window) {
    vcad_engine* e = cx.e; const vcad_config& c = e->c; const int H = c.hidden_size;
    AttnParams p; memset(&p, 0, sizeof(p));
    p.q = q; p.ldq = H; p.k = kv_cache; p.v = (char*)kv_cache + (size_t)H * e->esz; p.ldk = p.ldv = 2 * H; p.o = o; p.ldo = H; p.lse = lse;
    p.B = e->B; p.H = c.nhead; p.Tq = 1; p.Tk = t + 1; p.window = window; p.causal = 1; p.qpos = t; p.kv_rows = e->infer_T;
    const int hd = H / c.nhead; p.scale = 1.0f / sqrtf((float)hd);
    return vc_attn_fwd(e->dt, hd, p, cx.s);
}

int vcad_infer_step(vcad_engine* e, int t, const void* frame, int64_t frame_bstride, const float* action_norm, float* cmds_out, float* pars_out,
                    void* stream) {
    if (!e->infer_T || e->planned_ws != e->ws || e->T != 1) { vc_set_error("vcad_infer_step: call vcad_infer_begin first"); return VC_ERR_ARG; }
    if (t != e->infer_t || t >= e->infer_T) { vc_set_error("vcad_infer_step: step %d out of order (next is %d of %d)", t, e->infer_t, e->infer_T); return VC_ERR_ARG; }
    const vcad_config& c = e->c;
    const bool pa = c.enable_past_actions, ps = c.enable_past_states;
    if (ps && !frame) { vc_set_error("vcad_infer_step: frame is null"); return VC_ERR_ARG; }
    if (pa && !action_norm) { vc_set_error("vcad_infer_step: action is null"); return VC_ERR_ARG; }
    if (e->infer_u8 && (((uintptr_t)frame | (uintptr_t)frame_bstride) & 3)) { vc_set_error("vcad_infer_step: uint8 frames must be 4-byte aligned"); return VC_ERR_ARG; }
    vc_stream_t s = (vc_stream_t)stream; Ctx cx{e, s};
    const int B = e->B, H = c.hidden_size, D = c.vit_dim, ff = c.dim_feedforward, Tmax = e->infer_T; const long M = B;
    const size_t es = e->esz;
    const float keep_p = e->drop_p; e->drop_p = 0.f;
    const float* ts = c.enable_timestep_embedding ? cx.Pf(e->o_ts) + (long)t * H : nullptr;     // row t of the timestep table
    e->in_frames = frame; e->in_fbstride = frame_bstride; e->in_actions = action_norm; e->in_u8 = e->infer_u8;
    auto body = [&]() -> int {
        if (ps) {
            CK(cx.refresh_q8());
            CK(vit_forward(cx, 0, frame, 1, frame_bstride));
            Epi ep; ep.bias = cx.Pf(e->o_es_b); ep.rowadd = ts; ep.rdiv = 1; ep.rmod = 1; ep.ldrow = H; ep.act = VC_ACT_TANH;
            CK(cx.lin_fwd(cx.AT(e->va[0].e, D), cx.W(e->o_es_w, D), cx.A32(e->ui, H), B, H, D, ep));
        }
        if (pa && ps) {
            Epi ep; ep.rowadd = e->cadterm; ep.rdiv = 1; ep.rmod = 0; ep.ldrow = H; ep.act = VC_ACT_TANH;
            CK(cx.lin_fwd(cx.A32(e->ui, H), cx.W(e->o_ip_w, 2 * H), cx.A32(e->mem, H), B, H, H, ep));
        } else {
            CK(vc_bcast_tanh(e->dt, e->cadE, e->mem, M, H, 1, s));
        }
        if (pa) CK(vc_embed_action(VC_F32, action_norm, cx.Pf(e->o_ea_w), cx.Pf(e->o_ea_b), ts, e->act, nullptr, M, H, c.act_dim, 1, s));
        const float* x = pa ? e->act : (ps ? e->ui : e->mem);
        const int sa_window = pa ? Tmax : c.window_size;
        const long cld = (long)Tmax * 2 * H;                                   // cache row stride between clips, in elements
        for (int L = 0; L < c.num_decoder_layers; ++L) {
            const DecW& w = e->wd[L]; DecLayerActs& d = e->da[L];
            char* kvs = (char*)e->ic_kv_s[L]; char* kvc = (char*)e->ic_kv_c[L];
            { Epi ep; ep.bias = cx.Pf(w.sa_b); CK(cx.lin_fwd(cx.A32(x, H), cx.W(w.sa_w, H), cx.AT(d.qkv_s, H), B, H, H, ep)); }
            { Epi ep; ep.bias = cx.Pf(w.sa_b + H);
              CK(cx.lin_fwd(cx.A32(x, H), cx.W(w.sa_w + (long)H * H, H), cx.AT(kvs + (size_t)t * 2 * H * es, cld), B, 2 * H, H, ep)); }
            CK(infer_attn(cx, d.qkv_s, kvs, d.ao_s, d.lse_s, t, sa_window));
            { Epi ep; ep.bias = cx.Pf(w.sa_ob); ep.residual = x; ep.ldr = H; CK(cx.lin_fwd(cx.AT(d.ao_s, H), cx.W(w.sa_ow, H), cx.A32(d.s1, H), B, H, H, ep)); }
            CK(cx.ln_fwd(VC_F32, d.s1, H, w.n1w, w.n1b, d.x1, H, nullptr, 0, d.st1, M, H));
            { Epi ep; ep.bias = cx.Pf(w.ca_b); CK(cx.lin_fwd(cx.A32(d.x1, H), cx.W(w.ca_w, H), cx.AT(d.q_c, H), B, H, H, ep)); }
            { Epi ep; ep.bias = cx.Pf(w.ca_b + H);
              CK(cx.lin_fwd(cx.A32(e->mem, H), cx.W(w.ca_w + (long)H * H, H), cx.AT(kvc + (size_t)t * 2 * H * es, cld), B, 2 * H, H, ep)); }
            CK(infer_attn(cx, d.q_c, kvc, d.ao_c, d.lse_c, t, c.window_size));
            { Epi ep; ep.bias = cx.Pf(w.ca_ob); ep.residual = d.x1; ep.ldr = H; CK(cx.lin_fwd(cx.AT(d.ao_c, H), cx.W(w.ca_ow, H), cx.A32(d.s2, H), B, H, H, ep)); }
            CK(cx.ln_fwd(VC_F32, d.s2, H, w.n2w, w.n2b, d.x2, H, nullptr, 0, d.st2, M, H));
            { Epi ep; ep.bias = cx.Pf(w.b1); ep.act = VC_ACT_RELU; CK(cx.lin_fwd(cx.A32(d.x2, H), cx.W(w.w1, H), cx.AT(d.f1, ff), B, ff, H, ep)); }
            { Epi ep; ep.bias = cx.Pf(w.b2); ep.residual = d.x2; ep.ldr = H; CK(cx.lin_fwd(cx.AT(d.f1, ff), cx.W(w.w2, ff), cx.A32(d.s3, H), B, H, ff, ep)); }
            CK(cx.ln_fwd(VC_F32, d.s3, H, w.n3w, w.n3b, d.x3, H, nullptr, 0, d.st3, M, H));
            x = d.x3;
        }
        const int n5 = c.num_classes, n6 = c.num_params * c.num_params_values;
        { Epi ep; ep.bias = cx.Pf(e->o_h5_b); CK(cx.lin_fwd(cx.A32(x, H), cx.W(e->o_h5_w, H), cx.A32(cmds_out, n5), B, n5, H, ep)); }
        { Epi ep; ep.bias = cx.Pf(e->o_h6_b); CK(cx.lin_fwd(cx.A32(x, H), cx.W(e->o_h6_w, H), cx.A32(pars_out, n6), B, n6, H, ep)); }
        return 0;
    };
    const int rc = body();
    e->drop_p = keep_p;
    if (rc) return rc;
    if (vc_last_launch_error()) { vc_set_error("vcad_infer_step: kernel launch failed"); return VC_ERR_LAUNCH; }
    e->infer_t = t + 1;
    return 0;
}

int vcad_optimizer_step(vcad_engine* e, float lr, float b1, float b2, float eps, float max_norm, int step, float gscale,
                        float* norm_out, void* stream) {
    float lrs[NB_BUCKETS]; for (int i = 0; i < NB_BUCKETS; ++i) lrs[i] = lr;
    return vcad_optimizer_step_groups(e, lrs, b1, b2, eps, max_norm, step, gscale, norm_out, stream);
}
int vcad_optimizer_step_groups(vcad_engine* e, const float* lr_bucket, float b1, float b2, float eps, float max_norm, int step, float gscale,
                               float* norm_out, void* stream) {
    if (!lr_bucket) { vc_set_error("vcad_optimizer_step_groups: lr_per_bucket is null"); return VC_ERR_ARG; }
    if (!e->P || !e->G || !e->Mm || !e->Vv) { vc_set_error("vcad_optimizer_step: buffers not bound"); return VC_ERR_ARG; }
    if (!e->ws) { vc_set_error("vcad_optimizer_step: no workspace"); return VC_ERR_WORKSPACE; }
    if (step < 1) { vc_set_error("vcad_optimizer_step: step must be >= 1"); return VC_ERR_ARG; }
    CK(need_device(e, "vcad_optimizer_step"));
    if (e->B == 0) plan(e, 1, 1, e->ws);
    vc_stream_t s = (vc_stream_t)stream;
    // (deferred unscale, fp16 engines: the last backward left scale x gradient in G — exact division by a power of two, here instead of in five passes over G)
    const float unscale = e->grads_scaled_by != 0.0f ? 1.0f / e->grads_scaled_by : 0.0f;     // (Adam writes g / scale back: the buffer holds true gradients again after the step)
    if (unscale != 0.0f) { gscale *= unscale; e->grads_scaled_by = 0.0f; }
    // the norm the reference clips against is that of the (already averaged) gradients
    CK(vc_grad_norm(e->G, e->ptotal, max_norm, gscale, e->norm_part, e->norm_out, s));
    // one launch per run of buckets that share a learning rate (the reference's `frozen` mode, trainer.py:237-251, gives the CAD ViT,
    // the frame ViT and everything else their own lr: exactly buckets 2, 3-4 and 0-1 of the flat layout); the clip norm stays global
    for (int b0 = 0; b0 < NB_BUCKETS;) {
        int b1i = b0; while (b1i + 1 < NB_BUCKETS && lr_bucket[b1i + 1] == lr_bucket[b0]) ++b1i;
        const long lo = e->buckets[b0].first, hi = e->buckets[b1i].second;
        AdamParams a; memset(&a, 0, sizeof(a));
        a.p = e->P + lo; a.g = e->G + lo; a.m = e->Mm + lo; a.v = e->Vv + lo; a.n = hi - lo; a.lr = lr_bucket[b0]; a.beta1 = b1; a.beta2 = b2; a.eps = eps;
        a.bc1 = 1.0f - powf(b1, (float)step); a.bc2 = 1.0f - powf(b2, (float)step);
        a.clip = max_norm > 0.f ? e->norm_out + 1 : nullptr; a.finite = e->c.dtype == VCAD_F16 ? e->norm_out + 2 : nullptr; a.gscale = gscale; a.shadow = e->S ? e->S + lo : nullptr; a.shadow_pk = e->Spk ? e->Spk + lo : nullptr;
        if (unscale != 0.0f) { a.gw = e->G + lo; a.gw_mul = unscale; }
        CK(vc_adam(a, s));
        b0 = b1i + 1;
    }
    e->wT_fresh = false; e->pe_fresh[0] = e->pe_fresh[1] = false;                  // the bf16 shadow just changed: its transposed copies are rebuilt before the next backward
    e->q8_fresh = false;                  // ... and the fp8 copies before the next forward
    if (norm_out) CK(vc_memcpy_d2d_async(norm_out, e->norm_out, 2 * 4, s));
    return 0;
}

}  // extern "C"
