// ops.h — host-side launchers for the hot-path kernels (used by engine.cpp and by the per-op C ABI).
#pragma once
#include "vc_rt.h"
#include "gemm.h"
#include "norm.h"
#include "attn.h"
#include "attn_cls.h"
#include "loss.h"
#include "optim.h"
#include "ab.h"

enum { VC_F32 = 0, VC_BF16 = 1, VC_X3 = 2 /* GEMM compute type only: bf16x3 on fp32 tensors (gemm.h) */,
       VC_PK = 3 /* storage type of a bf16x3 GEMM's B operand only: pre-split hi | lo words (gemm.h vc_pk) */ };
enum { VC_OK = 0, VC_ERR_ARG = 1, VC_ERR_UNSUPPORTED = 2, VC_ERR_LAUNCH = 3, VC_ERR_WORKSPACE = 4 };

void vc_set_error(const char* fmt, ...);

// ---- optional HIP-event profiler (off by default; bench.py turns it on for ONE extra step after the timed region).
// Each launcher opens a scope: events are recorded on the launch stream around the kernel(s); vcad_profile_end sums
// elapsed time, algorithmic FLOPs and algorithmic bytes per category.
enum { VC_CAT_GEMM_FWD = 0, VC_CAT_GEMM_DGRAD = 1, VC_CAT_GEMM_WGRAD = 2, VC_CAT_ATTN = 3, VC_CAT_NORM = 4, VC_CAT_LOSS = 5,
       VC_CAT_OPTIM = 6, VC_CAT_OTHER = 7, VC_NCAT = 8 };
// kernel family of a scope (vcad_profile_kernel): the bench's roofline is quoted on the dominant one, the persistent DMA-fed GEMM
enum { VC_TAG_NONE = 0, VC_TAG_GEMM_DMA = 1, VC_TAG_GEMM_REG = 2, VC_TAG_GEMM_MID = 3, VC_TAG_GEMM_GROUPED = 4, VC_NTAG = 5 };
struct ProfScope {
    void* rec;
    ProfScope(int cat, double flops, double bytes, vc_stream_t s, int tag = VC_TAG_NONE);
    ~ProfScope();
};
const char* vc_get_error();

// per-call kernel-selection flags (0 = automatic): how the tests put small problems on the tile size / kernel family the big shapes take.
// Plain arguments — the library has no process-global switches.
enum { VC_GF_TILE64 = 1, VC_GF_TILE128 = 2, VC_GF_DMA_NEVER = 4, VC_GF_DMA_ALWAYS = 8, VC_GF_WIDE_NEVER = 16, VC_GF_WIDE_ALWAYS = 32,
       VC_GF_MID_NEVER = 64, VC_GF_MID_ALWAYS = 128, VC_GF_XCD_COLS_SHIFT = 8 /* bits 8-11, XCD column groups of the persistent kernel: 0 automatic, 1 never, 2 / 4 / 8 forced */,
       VC_GF_NGROUP_SHIFT = 12 /* bits 12-15, register-staged kernel: tile columns per sweep (gemm.h n_group): 0 automatic, 1-15 forced */,
       VC_GF_DYNAMIC = 1 << 16 /* vcad_op_gemm: persistent kernel claims its items dynamically (counters carved from the scratch buffer) */,
       VC_GF_RESERVE_SHIFT = 17 /* bits 17-20: the persistent kernel launches on 256 - 8 n CUs (n = 0..15), leaving the rest to other streams' kernels (data-parallel runs) */,
       VC_GF_MINI_NEVER = 1 << 21, VC_GF_MINI_ALWAYS = 1 << 22 /* persistent kernel: mini tiles for the rows of a mostly empty last round (gemm_dma.h GdMini): automatic / never / forced (tests) */ };
struct GemmCall {
    int ct, sa, sb, to;         // compute / A-source / B-source / output dtypes
    int tra, trb;
    int role;                   // profiler category: 0 = infer from the layout, else VC_CAT_GEMM_* + 1 (a dgrad through W^T has the forward's layout)
    GemmParams p;               // vecA/vecB/k_per_split/partial are filled by vc_gemm
    unsigned flags;             // VC_GF_*
    int* kernel_out;            // optional: receives the VC_TAG_* of the kernel family that ran
    int* claim;                 // optional: 16 zeroed ints of device memory private to the launching stream — the persistent kernel then claims its items
                                // dynamically (gemm_dma.h) and leaves them zeroed; null = static item lists
};
// scratch: fp32 workspace for split-K partial slabs (may be null -> no split)
bool vc_profile_on();       // the HIP-event profiler is recording: per-kernel times must not overlap, so no side stream
int vc_gemm(GemmCall c, float* scratch, size_t scratch_bytes, vc_stream_t s);
int vc_gemm_prepare(GemmCall& c);                                   // validation + per-problem legality flags (ops_gemm.hip)
int vc_gemm_dma_launch(GemmCall c, int nsplit, int BN, vc_stream_t s);   // persistent DMA-fed kernel (ops_gemm_dma.hip), tile 256 x BN; c already prepared
int vc_gemm_dma_wgrad_batched(GemmCall* calls, int n, float* scratch, size_t scratch_bytes, vc_stream_t s);   // (ops_gemm_dma.hip) up to 4 plain 16-bit weight gradients over the same token rows in one persistent launch
int vc_gemm_mid_launch(GemmCall c, vc_stream_t s);          // gemm_mid.h (ops_gemm_mid.hip)
int vc_gemm_mid_tile_m(int trb); int vc_gemm_mid_tile_n(int trb);
int vc_gemm_mid_batched(GemmCall c, int batch, long bsa, long bsb, long bsc, vc_stream_t s);   // (ops_gemm_mid.hip) batch problems of one shape in one grid
// grouped launch of many same-signature problems in one grid (see ops_gemm.hip)
int vc_gemm_grouped_prepare(GemmCall* calls, int n, GemmParams* probs, int* tile_start, int tile = 128);       // tile: 128 (every signature) or 64 (16-bit weight gradients)
bool vc_gemm_grouped_has_forward();
int vc_gemm_grouped_launch(const GemmCall& sig, const GemmParams* probs, const int* tile_start, int n, int total_tiles, double flops, vc_stream_t s, int tile = 128);

int vc_ln_fwd(int tx, int ty, int C, int mode, LnFwdParams p, vc_stream_t s);
// partial_ws: >= ln_bwd_blocks(rows) * 2 * C floats; dgamma/dbeta written (not accumulated)
long vc_ln_bwd_blocks(long rows);
int vc_ln_bwd(int td, int tx, int ty, int C, int mode, LnBwdParams p, float* partial_ws, float* dgamma, float* dbeta,
              float* colsum_ws, vc_stream_t s, float* dsum_out = nullptr);   // dsum_out: column sums of the emitted gradient (norm.h LnBwdParams::dsum)
// out[b][c] (=|+=) sum_r x[b][r][c];  ws >= batch * nchunk(rows) * cols floats
long vc_colsum_chunks(long rows);
int vc_colsum_seg(const float* x, long ld, long rows, int seg, int nseg, float* out0, float* out1, float* out2, vc_stream_t s);
int vc_colsum(int tx, const void* x, long ld, long rows, int cols, float* out, int accumulate,
              int batch, long bstride_x, long bstride_out, float* ws, vc_stream_t s);
// out (fp32 or T) = in (fp32) * dropout mask; cols % 4 == 0, 16-byte-aligned rows
int vc_dropout_mul(int ty, const float* in, long ld_in, void* out, long ld_out, long rows, int cols, vc_drop d, vc_stream_t s);
int vc_dtanh(int ty, const float* d, const float* y, float* out32, void* outt, long n, vc_stream_t s);
int vc_embed_action(int ty, const float* a, const float* W, const float* b, const float* ts, float* y32, void* yt,
                    long M, int H, int K, int T, vc_stream_t s);
int vc_bcast_tanh(int ts, const void* src, float* out, long M, int H, int T, vc_stream_t s);
int vc_add_inplace(float* a, const float* b, long n, vc_stream_t s);
int vc_scale(const float* x, float* y, long n, float alpha, vc_stream_t s);       // y = alpha * x (y may be x); 16-byte aligned pointers take the vector kernel, anything else one element per thread
// gradient wire format (norm.h): 16-bit copies of a range of the fp32 gradient buffer in the library's storage format; amax = device scalar (max |g|
// over all ranks) that fixes the fp16 build's power-of-two scale, or null (scale 1: bf16 has fp32's range).  amax_out: 1 + 1024 floats.
int vc_wire_amax(const float* g, long n, float* amax_out, vc_stream_t s);
int vc_wire_pack(const float* g, void* wire, long n, const float* amax, int world, vc_stream_t s);
int vc_wire_unpack(const void* wire, float* g, long n, const float* amax, int world, vc_stream_t s);
int vc_zero_cols(void* p, long ld_bytes, long rows, long width_bytes, vc_stream_t s);      // strided memset: 16-byte aligned rows / width
int vc_cast(int ty, const float* x, void* y, long n, vc_stream_t s);
int vc_pack_x3(const float* x, uint32_t* y, long n, vc_stream_t s);       // y[i] = hi bf16(x[i]) << 16 | lo bf16(x[i] - hi)
// g = dropout(act(z)) / dz = (dz * dropmask) * act'(z): compact bf16 [rows, cols], masks indexed like the fused GEMM epilogue
int vc_act_fwd_bf16(const void* z, void* g, long rows, int cols, int act, vc_drop d, vc_stream_t s);
int vc_dact_bwd_bf16(void* dz, const void* z, long rows, int cols, int kind, vc_drop d, vc_stream_t s, float* colsum_out = nullptr, float* partial_ws = nullptr,
                     size_t partial_bytes = 0, float* colsum_ws = nullptr, bool defer_reduce = false);      // defer_reduce: leave the [vc_dact_bwd_blocks][cols] partial rows in partial_ws (the caller sums them later); fails if the fused form does not apply
long vc_dact_bwd_blocks(long rows, int cols);
bool vc_dact_bwd_fused_ok(int cols);           // the activation-derivative pass can reduce its output over rows (partial-row form): cols / 8 divides 256
// grouped column sums: jobs / partial are device pointers; max_chunks = ceil(max rows / 128), strips = total 256-column strips
int vc_colsum_grouped(const ColsumJob* jobs, int njobs, int strips, int max_chunks, float* partial, vc_stream_t s);
int vc_transpose_bf16(const vc_bf16* src, vc_bf16* dst, int rows, int cols, vc_stream_t s);
// n matrices of one source / destination buffer in one grid (element offsets; 32 jobs per launch)
int vc_transpose_bf16_batched(const vc_bf16* S, vc_bf16* D, const long* src_off, const long* dst_off, const int* rows, const int* cols, int n, vc_stream_t s);
// LayerNorm affine folded into the Linear behind it (norm.h): Wf = 16-bit(W diag(gamma)), bf = b + W beta;  backward: dW, dgamma, dbeta from the folded Linear's dWf and db
int vc_pe_fold(const float* W, const float* b, const float* gamma, const float* beta, void* Wf, float* bf, int D, int K, vc_stream_t s);
int vc_pe_fold_bwd(const float* dWf, const float* S, const float* W, const float* gamma, const float* beta, float* dW, float* dgamma_dbeta, int D, int K,
                   float* partial_ws /* >= 32 * 2 K floats */, float* colsum_ws, vc_stream_t s);       // dgamma | dbeta contiguous [2 K]     // dst[c][r] = src[r][c]

// MXFP8 (gemm_mx8.h): quantise [rows, cols] (fp32 / bf16) to e4m3 + E8M0 block scales; C = epilogue(A8 B8^T)
struct Mx8Params { GemmParams g; const uint8_t* sa; long ldsa; const uint8_t* sb; long ldsb; };     // fp8 operands + their E8M0 scale matrices [rows][K / 32]
int vc_mx8_quant(int tx, const void* x, long ld, uint8_t* q, uint8_t* sc, long rows, int cols, vc_stream_t s);
int vc_gemm_mx8(Mx8Params q, int to, vc_stream_t s);

int vc_attn_fwd(int t, int D, AttnParams p, vc_stream_t s);
int vc_attn_bwd(int t, int D, AttnParams p, vc_stream_t s);
// class-token attention of the last ViT layer on (g = q W_k, normalised tokens) instead of projected keys / values (attn_cls.h); 16-bit storage only
bool vc_cls_attn_ok(int D, int H, int P1, int dim_head);
int vc_cls_attn_fwd(ClsAttnParams p, vc_stream_t s);
int vc_cls_attn_bwd(ClsAttnParams p, vc_stream_t s);

int vc_loss_fwd(LossParams p, vc_stream_t s);          // rows + finalize
int vc_loss_bwd(LossParams p, vc_stream_t s);          // dlogits

// norm_out[0] = |g|, norm_out[1] = clip coef.  partial >= 1024 floats.
int vc_grad_norm(const float* g, long n, float max_norm, float gscale, float* partial, float* norm_out, vc_stream_t s);
int vc_adam(AdamParams a, vc_stream_t s);
