/* vcad.h — C ABI of libvcad_hip.so: the MI355X-native hot path of VideoCAD's behaviour-cloning train step.
 *
 * The reference (ghadinehme/VideoCAD) has NO FFI/plugin layer: its boundary is the Python protocol
 *   ModelFactory.create_model(...)                      reference model/model_factory.py:15-36
 *   AutoRegressiveTransformer.forward(inputs)           reference model/autoregressive_transformer.py:121-220
 *   BaseTrainer._process_batch / compute_loss           reference trainer.py:480-496, 935-1063
 * so this header defines the binding a maintainer adds underneath those classes (see INTEGRATION.md for the
 * ctypes stub).  Each entry point names the reference code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; the caller (PyTorch caching allocator) owns
 *     every buffer including the workspace.  The library allocates no device MEMORY; the device objects it owns are, per engine, one
 *     non-blocking side stream + four events (created on first use, destroyed with the engine: the CAD ViT runs there beside the frame ViT)
 *     and, per process, the HIP-event profiler's event list between vcad_profile_begin / _end (off by default)
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*); nothing here synchronises the device
 *   - return value 0 = ok; non-zero = error, text via vcad_last_error() (thread-local)
 *   - dtype: VCAD_F32 = exact-fp32 parity mode (f32 MFMA), VCAD_BF16 = bf16 MFMA with fp32 accumulate,
 *     fp32 residual stream, fp32 master weights + bf16 weight shadow; VCAD_BF16X3 = fp32 tensors everywhere (like VCAD_F32; the
 *     shadow holds the weights pre-split into hi | lo bf16 words), every Linear on the bf16 matrix cores with hi/lo operand splits (three MFMAs per product, fp32 accumulate): the
 *     in-tolerance throughput mode (logits within 1e-3 of the fp32 reference, reference main.py:28 allows TF32 there)
 */
#ifndef VCAD_H
#define VCAD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { VCAD_F32 = 0, VCAD_BF16 = 1, VCAD_BF16X3 = 2, VCAD_F16 = 3 };

/* The 16-bit storage format is a build-time property of the kernel library: libvcad_hip.so stores bf16 (VCAD_BF16, VCAD_BF16X3, VCAD_F32 engines),
 * libvcad_hip_f16.so — the same sources compiled with -DVC_H16, the same entry points — stores IEEE fp16 (VCAD_F16 engines): the kernels,
 * tensors and MFMA rate of VCAD_BF16 with 10 mantissa bits instead of 7 (logits within 1e-3 of the fp32 reference), and a gradient scale that
 * keeps the backward inside fp16's exponent range (vcad_set_grad_scale).  Each library rejects the other's dtypes.  "bf16" / "f16": */
const char* vcad_storage_format(void);

typedef struct vcad_config {
    /* AutoRegressiveTransformer.__init__ kwargs (reference model/autoregressive_transformer.py:11-35) */
    int hidden_size, nhead, num_decoder_layers, dim_feedforward, window_size, act_dim;
    int num_classes, num_params, num_params_values, max_ep_len;
    /* vit_pytorch.ViT(...) constructor call at reference model/trajectory_model.py:54-65 */
    int vit_dim, vit_depth, vit_heads, vit_dim_head, vit_mlp, image_size, patch_size;
    int dtype;                     /* VCAD_F32 | VCAD_BF16 | VCAD_BF16X3 (libvcad_hip.so) | VCAD_F16 (libvcad_hip_f16.so) */
    /* wiring flags of forward (reference model/autoregressive_transformer.py:149-213) */
    int enable_past_actions, enable_past_states, enable_timestep_embedding;
    int num_views;          /* multiview branch (reference model/autoregressive_transformer.py:72-74,167-170): 0 = off (every final_experiments.json entry) */
} vcad_config;

typedef struct vcad_engine vcad_engine;

const char* vcad_last_error(void);
const char* vcad_version(void);

/* ---- engine lifetime (host-only bookkeeping; replaces AutoRegressiveTransformer.__init__ parameter creation) */
int vcad_engine_create(const vcad_config* cfg, vcad_engine** out);
void vcad_engine_destroy(vcad_engine* e);

/* ---- flat parameter layout (state_dict keys of SURVEY Appendix B, live parameters only, reverse-execution order) */
int64_t vcad_param_total(const vcad_engine* e);                   /* floats in the flat buffer (incl. alignment pads) */
int vcad_param_count(const vcad_engine* e);                       /* number of named tensors */
/* name -> host char buffer; shape[4] padded with 0; offset/numel in floats */
int vcad_param_info(const vcad_engine* e, int index, char* name, size_t name_cap, int64_t* offset, int64_t* numel,
                    int64_t shape[4], int* ndim);
/* number of DDP buckets (5) and the [begin,end) float range of bucket i; bucket i's gradients are final when
 * vcad_backward_stage(i) has been enqueued: 0 = heads + decoder, 1 = stem, 2 = CAD ViT, 3 = frame ViT upper half, 4 = lower half +
 * patch embed */
int vcad_bucket_count(const vcad_engine* e);
int vcad_bucket_range(const vcad_engine* e, int bucket, int64_t* begin, int64_t* end);

/* params/grads/m/v: fp32 [vcad_param_total]; shadow: VCAD_BF16 -> bf16 [vcad_param_total] (required); VCAD_BF16X3 -> uint32
 * [vcad_param_total] of pre-split hi | lo words (optional: NULL = the GEMMs split the fp32 weights while staging, same results, slower);
 * VCAD_F32 -> NULL.  The optimiser step keeps the shadow current. */
int vcad_bind(vcad_engine* e, float* params, float* grads, float* adam_m, float* adam_v, void* shadow);
/* refresh the weight shadow from the fp32 master weights (after load_state_dict / external optimiser) */
int vcad_sync_shadow(vcad_engine* e, void* stream);

size_t vcad_workspace_bytes(const vcad_engine* e, int B, int T);
int vcad_set_workspace(vcad_engine* e, void* workspace, size_t bytes);

/* ---- dropout (nn.Dropout / attention dropout at the sites of the reference's train-mode forward: vit-pytorch emb /
 * attention / to_out / feed-forward dropouts, TransformerDecoderLayer dropout, dropout1-3 and attention dropout).
 * p = 0 disables (model.eval()).  Masks are a stateless hash of (seed, site, element index): set a fresh seed before every
 * training forward; the backward of that forward regenerates the same masks.  No mask tensors are stored. */
/* Gradient scale of the backward.  VCAD_F16 engines start in AUTOMATIC mode (scale = 0): each plan sets the scale to 2 x the next power of two >= B * T (the loss
 * is a mean over B * T rows, so the scaled dlogits have the magnitudes of a one-row batch whatever the batch size: 4096 at 32 x 64), floor 1024, cap 2^20.
 * Every other engine: 1 = off.  A value names the scale: a power of two in [1, 2^24].  vcad_backward* multiplies the incoming dlogits by it (into a private
 * copy) and divides each gradient bucket by it — exactly — when the bucket is complete, before the bucket callback: the gradient buffer, the callback and
 * vcad_optimizer_step* only ever see true gradients.  A gradient that overflows fp16 anyway shows up as a non-finite
 * gradient norm: vcad_optimizer_step* then leaves weights and moments untouched (norm_out[0] is inf / NaN); halve the scale and go on.
 * External dlogits handed to vcad_backward* need 4-byte alignment only (a 16-byte aligned pointer takes the vector copy).
 * Re-plans the workspace: call before the next forward.  vcad_grad_scale: the value in force (after a plan in automatic mode: that plan's). */
int vcad_set_grad_scale(vcad_engine* e, float scale);
float vcad_grad_scale(const vcad_engine* e);
/* r06, the native train step (forward, vcad_loss, vcad_backward*, vcad_optimizer_step* with nothing reading gradients or dlogits in between): with on = 1 the gradient
 * buckets stay multiplied by the gradient scale until the optimiser — its norm pass and Adam divide, exactly, and Adam writes the true gradient back, so the buffer
 * holds true gradients again once the step is through — instead of five passes over the gradient buffer, and vcad_loss writes scale x dlogits straight into the
 * backward's private copy (the tensors at vcad_dlogits_offsets are then NOT written).  Bucket callbacks and data-parallel exchanges would see scaled gradients in
 * this mode; the default (0) keeps true gradients in the buffer after every stage. */
int vcad_set_defer_unscale(vcad_engine* e, int on);

/* VCAD_FP8 forward mode (bf16 engines): the four Linear layers of every full ViT layer run on the block-scaled fp8 matrix cores
 * (MXFP8: e4m3 elements, one E8M0 scale per 32 k-values, fp32 accumulate); weights are re-quantised from the fp32 master after every
 * optimiser step, activations right before each GEMM; the backward pass is the bf16 one.  Re-plans the workspace (query
 * vcad_workspace_bytes again).  Replaces nothing in the reference: BASELINE configs[4]'s "fp8 MFMA" variant. */
int vcad_set_fp8(vcad_engine* e, int on);
int vcad_set_dropout(vcad_engine* e, float p, uint64_t seed);
/* multiview images of the NEXT vcad_forward* call (engines with num_views > 0): [B][num_views] gray planes in the CAD image's pixel format, contiguous */
int vcad_set_multiview(vcad_engine* e, const void* images);
/* keep-multipliers of one site, recomputed from (seed, site, index); a pure function of the engine's dropout setting (the parity tests hand them to the oracle)
 *  (module 1 = frame ViT, 2 = CAD ViT, 3 = decoder; kind ids in engine.hip) -> HOST buffer */
int vcad_dropout_mask(const vcad_engine* e, int module, int layer, int kind, int64_t n, float* host_out);
/* same, elements first .. first + n - 1 of the site's index space */
int vcad_dropout_mask_range(const vcad_engine* e, int module, int layer, int kind, int64_t first, int64_t n, float* host_out);

/* ---- AutoRegressiveTransformer.forward (reference model/autoregressive_transformer.py:121-220)
 * frames: fp32, frame (b,t) at frames + b*frame_bstride + t*S*S  (so batch['frames'][:, :-1] needs no copy)
 * actions_norm: fp32 [B,T,7] already normalised (reference trainer.py:800-804); cad: fp32 [B,1,S,S]
 * cmds_out fp32 [B,T,num_classes], params_out fp32 [B,T,num_params*num_params_values]
 * Limits: 1 <= T <= min(max_ep_len, 1024) (r04; the dataset's maximum horizon is 186, reference README.md:40, the reference's max_ep_len 1000:
 * bf16 mode streams the decoder attention over 64-step blocks, the fp32 / bf16x3 modes walk up to sixteen 64-key pieces per query) and
 * B*T*50*3072 < 2^32 (dropout indices). */
int vcad_forward(vcad_engine* e, const float* frames, int64_t frame_bstride, const float* actions_norm, const float* cad,
                 int B, int T, float* cmds_out, float* params_out, void* stream);

/* The same with the loader's native pixels: frames / cad are uint8 GRAYSCALE (PIL 'L' of the stored frames, reference
 * data_loader.py:441-447; cv2 BGR2GRAY of the CAD render, :471-476) and torchvision's ToTensor + Normalize(0.5, 0.5) (reference
 * main.py:103-108) is applied inside the patchify kernels with the same fp32 operations — bit-identical patch vectors, a quarter
 * of the PCIe / HBM bytes.  frame_bstride in BYTES(= pixels); pointers and stride 4-byte aligned. */
int vcad_forward_u8(vcad_engine* e, const uint8_t* frames, int64_t frame_bstride, const float* actions_norm, const uint8_t* cad,
                    int B, int T, float* cmds_out, float* params_out, void* stream);

/* The same with the dataset's STORED frames: uint8 RGB, interleaved [B][S][H][W][3] (pkl `frames uint8 [N,224,224,3]`, reference
 * data_loader/sequence_retriver.py:25-33); PIL's `convert('L')` integer luma (reference data_loader.py:441-447, torchvision Grayscale at main.py:105),
 * ToTensor and Normalize(0.5, 0.5) all happen inside the patchify kernels — bit-identical to the fp32 path.  cad: uint8 GRAY [B,1,S,S] as in
 * vcad_forward_u8 (one image per clip; cv2's BGR2GRAY stays on the host).  frame_bstride in PIXELS (multiple of 4), pointers 4-byte aligned. */
int vcad_forward_rgb8(vcad_engine* e, const uint8_t* frames_rgb, int64_t frame_bstride, const float* actions_norm, const uint8_t* cad,
                      int B, int T, float* cmds_out, float* params_out, void* stream);

/* ---- MultiClassesTrainer.compute_loss (reference trainer.py:935-1063, flexible_cross_entropy :853-917)
 * targets: fp32 [B*T,7] = raw batch['actions'][:, 1:]
 * label_weights: HOST pointer, 5 floats = class_weights.json["Label"] as the caller read it from ./class_weights.json
 *   (reference trainer.py:822-825: the command-CE class weights AND, through [0,0,1,1,2,3], the six per-head multipliers :962)
 * class_weights: DEVICE fp32 [6][1000] per-class CE weights of the six parameter heads (use_mse = 0) or NULL (use_mse = 1)
 * loss_out: fp32 [8] = total, cmd, param0..5;  metrics_out: int32 [32] (slots in loss.h)
 * also leaves d(loss)/d(logits) in the workspace for vcad_backward(NULL, NULL) */
int vcad_loss(vcad_engine* e, const float* cmds, const float* params, const float* targets, int B, int T, int use_mse,
              const float* label_weights, const float* class_weights, float* loss_out, int32_t* metrics_out, void* stream);

/* byte offsets (inside the caller's workspace) of the d(loss)/d(logits) tensors vcad_loss wrote: fp32 [B*T,5], [B*T,6000] */
int vcad_dlogits_offsets(const vcad_engine* e, size_t* off_cmds, size_t* off_params);

/* ---- autograd backward of forward (+ what `loss.backward()` does at reference trainer.py:492)
 * dcmds/dparams: fp32 gradients of the two outputs, or both NULL to use the ones vcad_loss left in the workspace.
 * Gradients are WRITTEN (not accumulated) into the bound flat grad buffer. */
int vcad_backward(vcad_engine* e, const float* dcmds, const float* dparams, void* stream);
/* the same, one DDP bucket at a time (stage 0 .. vcad_bucket_count-1, in order) so the caller can overlap RCCL */
int vcad_backward_stage(vcad_engine* e, int stage, const float* dcmds, const float* dparams, void* stream);
/* Stage vcad_side_stage() (= 2: the CAD ViT's backward, independent of the frame ViT's stages 3-4) on the library's side stream: the caller
   goes on with stages 3 and 4 and their all-reduces; vcad_join_side(stream) makes `stream` (the caller's, or its communication stream) wait
   for that stage, after which bucket 2 may be reduced.  When the engine cannot fork (enable_past_states off, vcad_set_side_stream(0), the
   HIP-event profiler recording) the stage runs on the caller's stream in line and vcad_join_side is a no-op: a communication stream must
   then wait for the caller's stream itself (videocad_amd/trainer.py: GradSync does both). */
int vcad_side_stage(const vcad_engine* e);
/* Data parallelism for a binder WITHOUT PyTorch (r04): a per-bucket hook of the single-call backward.  vcad_backward calls fn(user, bucket, grads, count,
 * stream) right after it has enqueued the last launch that writes bucket `bucket` (grads = device pointer to its `count` fp32 gradients inside the bound
 * buffer, stream = the hipStream_t those launches are on — the caller's stream, or the library's side stream for the CAD ViT's bucket and, in train mode,
 * for bucket 0 whose deferred weight gradients run there).  The hook must not block: record an event on `stream`, make its communication stream wait for it
 * and enqueue ncclAllReduce(grads, grads, count, ncclFloat, ncclSum, comm, comm_stream) (RCCL); before vcad_optimizer_step the caller makes ITS stream
 * wait for the communication stream and passes grad_scale = 1 / world.  The library never touches a bucket again after its hook ran.  NULL = off.
 * (The reference's DDP wrap at experiment.py:104-109; the Python trainer does the same through vcad_backward_stage + torch.distributed.) */
typedef int (*vcad_bucket_ready_fn)(void* user, int bucket, float* grads, int64_t count, void* stream);
int vcad_set_bucket_callback(vcad_engine* e, vcad_bucket_ready_fn fn, void* user);
int vcad_backward_stage_side(vcad_engine* e, int stage, const float* dcmds, const float* dparams, void* stream);
int vcad_join_side(vcad_engine* e, void* stream);

/* ---- half-precision WIRE FORMAT for the gradient exchange (r05; optional — the reference's DDP buckets are fp32, experiment.py:104-109):
 * a finished range [lo, hi) of the flat fp32 gradient buffer (lo % 4 == 0) is copied into a caller-owned buffer of 16-bit elements in THIS LIBRARY'S
 * storage format (vcad_storage_format(): bf16, or IEEE half in libvcad_hip_f16.so), summed over ranks there (ncclAllReduce / ReduceScatter with
 * ncclBfloat16 / ncclHalf) and copied back: half the xGMI bytes.  bf16 has fp32's exponent range: pass amax = NULL.  fp16 does not:
 *   vcad_wire_amax   writes max |g| of the range to amax_out[0] (amax_out: DEVICE, 1 + 1024 floats — the rest is reduction scratch; a NaN
 *                    gradient gives 3.4e38); the caller all-reduces that one float with MAX over the ranks;
 *   vcad_wire_pack   multiplies by 2^floor(log2(32768 / (world * amax))) — the sum over `world` ranks cannot overflow — while converting;
 *   vcad_wire_unpack divides by the same power of two (recomputed from the same device scalar) while converting back.
 * Each rank's addend is rounded to 8 (bf16) / 11 (fp16) significant bits and the collective accumulates in that format: expect ~4e-3 / ~5e-4
 * norm-wise on the summed bucket (tests/test_boundary_cpu.py bounds it).  All three only enqueue kernels on `stream`. */
int vcad_wire_amax(vcad_engine* e, int64_t lo, int64_t hi, float* amax_out, void* stream);
int vcad_wire_pack(vcad_engine* e, int64_t lo, int64_t hi, void* wire16, const float* amax, int world, void* stream);
int vcad_wire_unpack(vcad_engine* e, int64_t lo, int64_t hi, const void* wire16, const float* amax, int world, void* stream);

/* ---- clip_grad_norm_(max_norm) + Adam.step (reference trainer.py:493-494); step = 1-based Adam step count;
 * grad_scale multiplies gradients first (1/world_size after an all-reduce SUM); norm_out: fp32 [2] = |g|, clip coef.  VCAD_F16 engines only: a non-finite |g|
 * (the scaled backward overflowed fp16) skips the update — weights, moments and shadow stay as they are; the caller should then not count the
 * step (pass the same `step` again next time) and lower the scale (vcad_set_grad_scale).  Every other dtype does what the reference's
 * clip_grad_norm_ + Adam do with a non-finite gradient: it propagates into the weights and the run fails loudly. */
int vcad_optimizer_step(vcad_engine* e, float lr, float beta1, float beta2, float eps, float max_norm, int step,
                        float grad_scale, float* norm_out, void* stream);

/* per-bucket learning rates (lr_per_bucket: HOST, vcad_bucket_count() floats): the reference's `frozen` mode (trainer.py:237-251)
 * gives cad_embedding_model, state_embedding_model and the rest their own lr = buckets 2, 3-4 and 0-1; the clip norm stays global */
int vcad_optimizer_step_groups(vcad_engine* e, const float* lr_per_bucket, float beta1, float beta2, float eps, float max_norm,
                               int step, float grad_scale, float* norm_out, void* stream);

/* ---- AutoRegressiveTransformer.sequential_inference (reference model/autoregressive_transformer.py:222-275), incrementally:
 * the reference re-encodes the whole prefix at every step (O(T^2) ViT passes per clip); the model is causal, so each step here
 * encodes ONE new frame per clip and attends to cached per-layer keys / values.  eval mode (dropout off) regardless of
 * vcad_set_dropout.  The caller's workspace must hold vcad_infer_workspace_bytes(B, Tmax).
 *   vcad_infer_begin : CAD ViT + embed_image (+ its image_projection term), resets the caches; cad fp32 [B,1,S,S] (or uint8, _u8)
 *   vcad_infer_step  : steps t = 0, 1, ... in order; frame = frame t of every clip (clip b at frame + b*frame_bstride, same pixel
 *                      type as begin), action_norm fp32 [B,7] = the (normalised) action fed at position t (zeros, or the masked
 *                      arg-max of step t-1, reference :249-263); cmds_out fp32 [B,5], params_out fp32 [B,6000] = logits of step t */
size_t vcad_infer_workspace_bytes(const vcad_engine* e, int B, int Tmax);
int vcad_infer_begin(vcad_engine* e, const float* cad, int B, int Tmax, void* stream);
int vcad_infer_begin_u8(vcad_engine* e, const uint8_t* cad, int B, int Tmax, void* stream);
int vcad_infer_step(vcad_engine* e, int t, const void* frame, int64_t frame_bstride, const float* action_norm,
                    float* cmds_out, float* params_out, void* stream);

/* ---- optional HIP-event profiler: per kernel family (8 categories: gemm fwd/dgrad/wgrad, attention, norm, loss,
 * optimiser, other) elapsed ms, algorithmic FLOPs, algorithmic bytes, launches — measured on the launch stream. */
void vcad_profile_begin(void);
int vcad_profile_end(double ms[8], double flops[8], double bytes[8], int launches[8]);
/* totals of the last vcad_profile_end for one kernel family: 1 = persistent DMA-fed GEMM (gemm_dma_kernel), 2 = register-staged GEMM,
 * 3 = six-stage ring GEMM, 4 = grouped GEMM; out = {ms, flops, bytes, launches} */
int vcad_profile_kernel(int family, double out[4]);

/* ---- kernel-selection flags (0 = automatic).  The shipped library keeps NO process-global switches (the profiler scope above is the
 * only process-wide state): a test that wants a small problem on the kernel the C2 shapes take passes these per call (vcad_op_gemm) or per
 * engine (vcad_set_gemm_flags); every kernel they choose between computes the same result.  The product path never sets them. */
#define VCAD_GEMM_TILE64 1u        /* register-staged kernel: 64 x 64 block tile */
#define VCAD_GEMM_TILE128 2u       /* register-staged kernel: 128 x 128 block tile */
#define VCAD_GEMM_DMA_NEVER 4u     /* persistent DMA-fed kernel: never / whenever legal */
#define VCAD_GEMM_DMA_ALWAYS 8u
#define VCAD_GEMM_WIDE_NEVER 16u   /* its 256 x 256 tile (plain epilogues): never / whenever legal */
#define VCAD_GEMM_WIDE_ALWAYS 32u
#define VCAD_GEMM_MID_NEVER 64u    /* six-stage DMA-ring kernel for mid-size problems: never / whenever legal */
#define VCAD_GEMM_MID_ALWAYS 128u
#define VCAD_GEMM_DYNAMIC (1u << 16)   /* the persistent kernel claims its items with tickets instead of static per-workgroup lists: robust when other kernels (RCCL) hold CUs.
                                         * The one flag the product sets: the data-parallel trainer turns it on for world > 1 (vcad_set_gemm_flags); vcad_op_gemm carves the
                                         * counters from its scratch buffer */
#define VCAD_GEMM_RESERVE_CUS(n8) (((uint32_t)(n8) & 15u) << 17)   /* with VCAD_GEMM_DYNAMIC: the persistent kernel launches on 256 - 8 n8 CUs and leaves the rest to other streams'
                                         * kernels (the data-parallel trainer: RCCL's) — a persistent workgroup waiting for a taken CU stalls the dispatch of every other stream */
#define VCAD_GEMM_MINI_NEVER (1u << 21)  /* persistent kernel: never cut the rows of a mostly empty last round into 64-row mini tiles (default: automatic — 800 tiles on 256 CUs run as
                                         * 3 rounds + one mini per workgroup instead of 4 rounds) */
#define VCAD_GEMM_MINI_ALWAYS (1u << 22) /* tests: mini tiles from the last full tile row on, whatever the grid */
#define VCAD_GEMM_NGROUP(n) ((uint32_t)(n) << 12)    /* register-staged kernel: tile columns per sweep over all tile rows (0 automatic: only when B overflows an XCD's L2) */
#define VCAD_GEMM_XCD_COLS(n) ((uint32_t)(n) << 8)   /* XCD column groups of the persistent kernel's forward launches: 0 automatic, 1 never, 2 / 4 / 8 forced */
int vcad_set_gemm_flags(vcad_engine* e, uint32_t flags);
/* launches so far per kernel family (1 = persistent DMA-fed GEMM, 2 = register-staged, 3 = six-stage ring, 4 = grouped) by this engine */
int64_t vcad_kernel_launches(const vcad_engine* e, int family);
/* 0: the CAD tower and the deferred weight gradients stay on the caller's stream instead of the engine's side stream (1 = default) */
int vcad_set_side_stream(vcad_engine* e, int on);

#ifdef VCAD_AB
/* ---- A/B build only (`make -C videocad_amd/csrc ab` -> tools/_bin/libvcad_ab.so; csrc/ab.h): process-global selectors of the
 * measurement scripts under tools/, including the slower kernel variants that are not compiled into libvcad_hip.so. */
void vcad_debug_force_gemm_tile(int tile);   /* 64 / 128 / 0 */
void vcad_debug_gemm_dma(int mode);          /* -1 automatic, 0 never, 1 whenever legal */
void vcad_debug_gemm_wide(int mode);
void vcad_debug_gemm_mid(int mode);
void vcad_debug_gemm_xcd_cols(int xn);       /* -1 automatic, 0 never, 2 / 4 / 8 forced */
void vcad_debug_attn_variant(int v);         /* 0 = current attention kernels, 1 = r01 kernels */
void vcad_debug_gemm_waves(int n);           /* 256-wide tile of the persistent kernel: 8 waves (64 x 128 each) or 4 waves (128 x 128 each) */
void vcad_debug_split_gelu(int on);          /* ViT MLP activation as its own pass behind a plain GEMM (1, default) or fused into the GEMM epilogue (0) */
void vcad_debug_no_side_stream(int on);
void vcad_debug_gemm_policy(int bits);       /* dispatcher rules of r02: 1 = activation epilogues stay off the persistent kernel, 2 = small-tile-count wgrads too */
void vcad_debug_gemm_epilogue(int mode);     /* persistent kernel, k-contiguous B: -1 automatic, 0 row-per-lane (r01), 1 column-per-lane */
void vcad_debug_gemm_variant(int v);         /* 0 = lockstep persistent kernel, 1 = ping-pong wave groups (slower) */
void vcad_debug_gemm_stagger(int n);         /* ablation (tools/gemm_ablate*.py) */
void vcad_debug_gemm_skip(int mask);
void vcad_debug_res_in_ln(int on);           /* 16-bit ViT layers: residual add + branch dropout inside the LayerNorm pass behind to_out / net.4 (1, default) or in the GEMM epilogue (0: r04) */
void vcad_debug_wgrad_bk32(int on);          /* 256-wide weight-gradient kernel: 0 (default) = two 64-deep ring stages; 1 = r05's experiment, four 32-deep stages — measured slower (172 -> 211 us, profiles/r05_wgrad_bk32_ab.txt) */
void vcad_debug_cls_path(int on);            /* 16-bit engines, last ViT layer: class-token attention on (q W_k, normalised tokens) (1, default: r06, csrc/attn_cls.h) or K / V projections of all tokens (0: r05) */
void vcad_debug_dec_h16(int on);             /* 16-bit engines: decoder LayerNorms also emit 16-bit copies for the Linears / deferred weight gradients behind them (1, default: r06) or not (0: r05) */
void vcad_debug_pe_fold(int on);             /* 16-bit engines: patch-embedding LayerNorm affine folded into its Linear (1, default: r06) or applied to the patches with a dgrad + LayerNorm backward for its gradients (0: r05) */
void vcad_debug_splitk_r06(int on);       /* register-staged GEMM, k-slice rule: 0 (default: r04's) or 1 (r06 experiment: a k-tile priced at the 1.4 / 3 us it lasts in the model, up to 512 tiles — measured slower, profiles/r06_splitk_rule_ab.txt) */
void vcad_debug_batch_wgrad(int on);      /* 16-bit engines, train mode, full ViT layers: net.4 / net.0 / to_out weight gradients in ONE launch of the persistent kernel (1, default: r06) or three (0: r05) */
void vcad_debug_attn_prefetch(int frames); /* ViT attention backward: 0 (default) or: every workgroup warms the L2 for the (frame + frames, head) pair of a later workgroup on its XCD — r06 experiment, measured slower (profiles/r06_attn_prefetch_ab.txt) */
void vcad_debug_frame_first(int on);         /* whole backward with the side stream forked: frame tower's upper stage enqueued before the CAD tower's stage (1, default: r06) or after (0: r05) */
#endif

/* ---- single-op entry points (used by the parity tests; same kernels the engine launches) */
int vcad_op_gemm(int ct, int sa, int sb, int to, int tra, int trb, const void* A, const void* B, void* C,
                 int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, const float* bias, int act,
                 const float* residual, int64_t ldr, float alpha, float* scratch, size_t scratch_bytes, uint32_t flags /* VCAD_GEMM_* */,
                 int* kernel_out /* optional: kernel family that ran */, void* stream);
/* r06: n (<= 4) weight gradients dW_i[N_i, K_i] (fp32) = dY_i[tok, N_i]^T X_i[tok, K_i] (16-bit, compact rows) over the SAME tok rows in one launch of the persistent kernel —
 * what the engine does with a full ViT layer's net.4 / net.0 / to_out weight gradients.  N_i % 8 == 0, K_i % 256 == 0, tok % 64 == 0. */
int vcad_op_wgrad_batched(int n, const void* const* dY, const void* const* X, float* const* dW, const int* N, const int* K, int tok,
                          float* scratch, size_t scratch_bytes, uint32_t flags, void* stream);
/* y[i] = (RNE bf16(x[i]) << 16) | RNE bf16(x[i] - hi): the pre-split operand word of the bf16x3 GEMMs (storage type 3 as vcad_op_gemm's `sb`,
 * forward / dgrad layouts; bit-identical results to splitting inside the kernel, which is what an fp32 `sb` does) */
int vcad_op_pack_x3(const float* x, void* y, int64_t n, void* stream);
/* MXFP8 (VCAD_FP8 mode): x [rows, cols] fp32 / bf16 (tx) -> q [rows, cols] OCP e4m3 bytes + scales [rows, cols/32] E8M0 bytes (one
 * power-of-two scale per 32 consecutive elements); C (type `to`) = act(A8 B8^T + bias) + residual on the block-scaled fp8 MFMA */
int vcad_op_quant_mx8(int tx, const void* x, int64_t ldx, void* q, void* scales, int64_t rows, int cols, void* stream);
int vcad_op_gemm_mx8(int to, const void* A8, const void* sa, const void* B8, const void* sb, void* C, int M, int N, int K, int64_t ldc,
                     const float* bias, int act, const float* residual, int64_t ldr, void* stream);
int vcad_op_layernorm_fwd(int tx, int ty, int C, const void* x, int64_t ldx, const float* gamma, const float* beta,
                          float* y32, void* yt, float* stats, int64_t rows, float eps, void* stream);
int vcad_op_layernorm_bwd(int td, int ty, int C, const void* dy, const float* x, int64_t ldx, const float* stats,
                          const float* gamma, const float* add_in, float* dx32, void* dxt, float* dgamma, float* dbeta,
                          int64_t rows, float* scratch, size_t scratch_bytes, void* stream);
int vcad_op_attention_fwd(int t, int D, const void* q, const void* k, const void* v, void* o, int64_t ldq, int64_t ldk,
                          int64_t ldv, int64_t ldo, float* lse, int B, int H, int Tq, int Tk, int window, int causal,
                          float scale, void* stream);
int vcad_op_attention_bwd(int t, int D, const void* q, const void* k, const void* v, const void* dout, int64_t ldq,
                          int64_t ldk, int64_t ldv, int64_t lddo, const float* lse, float* delta, void* dq, void* dk,
                          void* dv, int64_t lddq, int64_t lddk, int64_t lddv, int B, int H, int Tq, int Tk, int window,
                          int causal, float scale, void* stream);
/* class-token attention of the LAST ViT layer, re-associated (csrc/attn_cls.h; restates vit-pytorch Attention for the pooled row, ctor call reference
 * model/trajectory_model.py:54-65): per frame n and head h, c = softmax_j(scale * g . ha_j) ha, with g = q W_k — no projected keys / values.  16-bit
 * storage tensors: ha [N * P1][512] (ld_ha), g / c / dc / dg [N][H][512]; lse [N][H] fp32; dha [N * P1][512]; r0 (optional) [N][512] fp32 = row 0 of dha. */
int vcad_op_cls_attention_fwd(const void* ha, int64_t ld_ha, const void* g, void* c, float* lse, int N, int H, int P1, float scale, void* stream);
int vcad_op_cls_attention_bwd(const void* ha, int64_t ld_ha, const void* g, const void* dc, const float* lse, void* dg, void* dha, int64_t ld_dha,
                              float* r0, int N, int H, int P1, float scale, void* stream);
/* same, with the saved forward output o passed along (the kernels recompute D_i from P and dP; o is accepted for callers that hold it) */
int vcad_op_attention_bwd_o(int t, int D, const void* q, const void* k, const void* v, const void* o, int64_t ldo, const void* dout,
                            int64_t ldq, int64_t ldk, int64_t ldv, int64_t lddo, const float* lse, float* delta, void* dq, void* dk,
                            void* dv, int64_t lddq, int64_t lddk, int64_t lddv, int B, int H, int Tq, int Tk, int window,
                            int causal, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
