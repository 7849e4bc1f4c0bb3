"""Per-kernel parity on a real MI355X through the C ABI (libvcad_hip.so) against plain PyTorch fp32/fp64 math."""
import pytest
import torch

import oputil as U
from videocad_amd import lib as L

pytestmark = pytest.mark.gpu
F32, BF16 = torch.float32, torch.bfloat16
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    return L.load()


@pytest.fixture(params=[64, 128], autouse=True)
def gemm_tile(request, hip):
    """every test in this file runs with both GEMM block tiles (64x64 and 128x128)"""
    if request.param == 64 and "gemm" not in request.node.name:
        pytest.skip("the block tile only matters to the GEMM tests (run once, with the 128 tile)")
    U.GEMM_FLAGS = L.GEMM_TILE64 if request.param == 64 else L.GEMM_TILE128
    yield request.param
    U.GEMM_FLAGS = 0


@pytest.mark.parametrize("tra,trb", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_f32_layouts(hip, tra, trb):
    U.check_gemm(hip, DEV, 300, 200, 260, F32, tra=tra, trb=trb, pad=4, bias=True, residual=True, splitk=False)
    U.check_gemm(hip, DEV, 77, 41, 53, F32, tra=tra, trb=trb, pad=3, bias=True, act=1, splitk=False)


def test_gemm_f32_big_and_splitk(hip):
    U.check_gemm(hip, DEV, 2048, 512, 1024, F32, bias=True, act=1)
    U.check_gemm(hip, DEV, 512, 384, 8192, F32, tra=1, trb=1)              # wgrad-shaped: split-K
    U.check_gemm(hip, DEV, 16, 5, 1024, F32, bias=True)                    # cmd head


@pytest.mark.parametrize("tra,trb", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("sa,to", [(BF16, BF16), (BF16, F32), (F32, BF16), (F32, F32)])
def test_gemm_bf16(hip, tra, trb, sa, to):
    if tra == 1 and to == BF16:
        pytest.skip("wgrad always writes fp32")
    U.check_gemm(hip, DEV, 520, 264, 392, BF16, sa=sa, to=to, tra=tra, trb=trb, pad=8, bias=True, act=2, splitk=False)
    U.check_gemm(hip, DEV, 33, 7, 100, BF16, sa=sa, to=to, tra=tra, trb=trb, pad=1, splitk=False)


@pytest.mark.parametrize("tra,trb", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_bf16x3(hip, tra, trb):
    """VCAD_BF16X3 compute type: fp32 operands split into hi / lo bf16 planes while staging, three MFMAs per product; checked against
    the exact product of the unrounded fp32 operands (tolerance 1.5e-5 relative: ~2^-17 per operand + the dropped lo*lo term)"""
    U.check_gemm(hip, DEV, 520, 264, 392, U.X3, tra=tra, trb=trb, pad=4, bias=True, act=2, residual=True, splitk=False, pack_b=not tra)
    U.check_gemm(hip, DEV, 33, 7, 100, U.X3, tra=tra, trb=trb, pad=1, splitk=False, pack_b=not tra)


def test_gemm_bf16x3_big(hip):
    U.check_gemm(hip, DEV, 5000, 3072, 512, U.X3, bias=True, pack_b=True)                # ViT QKV shape
    U.check_gemm(hip, DEV, 3072, 512, 20000, U.X3, tra=1, trb=1)            # ViT QKV wgrad, split-K
    U.check_gemm(hip, DEV, 4096, 512, 1024, U.X3, trb=1, residual=True, act=1, bias=True, pack_b=True)


def test_gemm_bf16_wgrad_f32_sources(hip):
    for sa in (BF16, F32):
        for sb in (BF16, F32):
            U.check_gemm(hip, DEV, 260, 136, 1000, BF16, sa=sa, sb=sb, to=F32, tra=1, trb=1, pad=4)


def test_gemm_bf16_big(hip):
    U.check_gemm(hip, DEV, 5000, 3072, 512, BF16, to=BF16)                 # ViT QKV shape
    U.check_gemm(hip, DEV, 3072, 512, 20000, BF16, sa=BF16, to=F32, tra=1, trb=1)   # ViT QKV wgrad, split-K
    U.check_gemm(hip, DEV, 4096, 512, 1024, BF16, sa=F32, to=F32, trb=1, residual=True)


def test_gemm_dma_kernel(hip):
    """the persistent DMA-fed kernel (forced: the automatic choice keeps some of these shapes on the register-staged kernel) at
    sizes where every workgroup streams several tiles — ragged M tail, k-slices with a short last slice, fused bias / GELU /
    residual — three times each: a mis-counted vmcnt shows up as sporadic wrong tiles"""
    dma = dict(flags=L.GEMM_DMA_ALWAYS, kernel=L.KERNEL_GEMM_DMA)
    for rep in range(3):
        U.check_gemm(hip, DEV, 20040, 512, 512, BF16, to=BF16, bias=True, act=1, seed=rep, **dma)
        U.check_gemm(hip, DEV, 20040, 512, 1024, BF16, to=F32, bias=True, residual=True, seed=rep, **dma)
        U.check_gemm(hip, DEV, 20040, 512, 3072, BF16, to=BF16, trb=1, seed=rep, **dma)
        U.check_gemm(hip, DEV, 3072, 512, 20032, BF16, to=F32, tra=1, trb=1, seed=rep, **dma)
        U.check_gemm(hip, DEV, 512, 512, 40000 - 64, BF16, to=F32, tra=1, trb=1, seed=rep, **dma)
    U.check_gemm(hip, DEV, 300, 256, 192, BF16, to=F32, bias=True, residual=True, pad=8, **dma)      # small problems, same kernel
    U.check_gemm(hip, DEV, 264, 128, 64, BF16, to=F32, tra=1, trb=1, pad=8, **dma)
    U.check_gemm(hip, DEV, 20040, 512, 1024, BF16, to=F32, bias=True, residual=True, kernel=L.KERNEL_GEMM_DMA)   # automatic choice: DMA kernel


def test_gemm_dma_kernel_mini_tiles(hip):
    """r06: mini tiles for the rows of a mostly empty last round (gemm_dma.h GdMini).  Automatic at the benchmark shapes (102 400 rows x 512 / 1 024 columns:
    800 / 1 600 tiles on 256 CUs), forced on a ragged small problem; static lists and ticket-drawn items; three times each (a mis-counted vmcnt or a
    surplus wave reading a stage shows up as sporadic wrong tiles).  The same calls with VCAD_GEMM_MINI_NEVER must give bit-identical results
    (same tile program, same k order) — check_gemm compares against the fp32 reference within the 16-bit tolerance either way."""
    dma = dict(kernel=L.KERNEL_GEMM_DMA)
    for rep in range(3):
        for fl in (L.GEMM_DMA_ALWAYS, L.GEMM_DMA_ALWAYS | L.GEMM_DYNAMIC, L.GEMM_DMA_ALWAYS | L.GEMM_MINI_NEVER):
            U.check_gemm(hip, DEV, 102400, 512, 512, BF16, to=BF16, bias=True, seed=rep, flags=fl, **dma)
            U.check_gemm(hip, DEV, 102400, 512, 1024, BF16, to=BF16, bias=True, seed=rep, flags=fl, **dma)
        U.check_gemm(hip, DEV, 102400, 1024, 512, BF16, to=BF16, seed=rep, flags=L.GEMM_DMA_ALWAYS, **dma)
        U.check_gemm(hip, DEV, 102400, 512, 3072, BF16, to=BF16, seed=rep, flags=L.GEMM_DMA_ALWAYS, **dma)
        for fl in (0, L.GEMM_DYNAMIC):
            U.check_gemm(hip, DEV, 790, 512, 192, BF16, to=F32, bias=True, residual=True, pad=8, seed=rep, flags=L.GEMM_DMA_ALWAYS | L.GEMM_MINI_ALWAYS | L.GEMM_WIDE_NEVER | fl, **dma)
            U.check_gemm(hip, DEV, 790, 512, 640, BF16, to=BF16, bias=True, pad=8, seed=rep, flags=L.GEMM_DMA_ALWAYS | L.GEMM_MINI_ALWAYS | L.GEMM_WIDE_ALWAYS | fl, **dma)
            U.check_gemm(hip, DEV, 20040, 512, 3072, BF16, to=BF16, trb=1, seed=rep, flags=L.GEMM_DMA_ALWAYS | L.GEMM_MINI_ALWAYS | fl, **dma)


def test_gemm_dma_kernel_batched_weight_gradients(hip):
    """r06: a full ViT layer's three small weight gradients (512 x 512, 512 x 512, 512 x 1024 over 102 400 token rows) in ONE launch of the persistent kernel
    (gemm_dma.h GdBatch: 16 tiles x 16 k-slices), static lists and ticket-drawn items, three times each; a ragged / four-problem case"""
    for rep in range(3):
        for fl in (0, L.GEMM_DYNAMIC):
            U.check_wgrad_batched(hip, DEV, 102400, [(512, 512), (512, 512), (512, 1024)], seed=rep, flags=fl)
        U.check_wgrad_batched(hip, DEV, 20032, [(512, 256), (256, 512), (264, 256), (512, 512)], seed=rep + 7)


def test_gemm_mid_kernel(hip):
    """six-stage DMA-ring kernel (gemm_mid.h) at the decoder's shapes: forward (k-contiguous W) and dgrad (row-contiguous W) layouts, fused
    epilogues, ragged M (B=16, T=186 -> 2976 rows), three times each (a mis-counted vmcnt shows up as sporadic wrong tiles); and the
    automatic choice takes it for a 2048 x 1024 x 1024 Linear"""
    mid = dict(flags=L.GEMM_DMA_NEVER | L.GEMM_MID_ALWAYS, kernel=L.KERNEL_GEMM_MID)
    for rep in range(3):
        U.check_gemm(hip, DEV, 2048, 1024, 1024, BF16, to=F32, bias=True, residual=True, seed=rep, **mid)
        U.check_gemm(hip, DEV, 2976, 3072, 1024, BF16, to=BF16, bias=True, seed=rep, **mid)
        U.check_gemm(hip, DEV, 2976, 1024, 1024, BF16, to=BF16, bias=True, act=2, seed=rep, **mid)
        U.check_gemm(hip, DEV, 2048, 1024, 3072, BF16, to=F32, trb=1, residual=True, seed=rep, **mid)
        U.check_gemm(hip, DEV, 2976, 1024, 2048, BF16, to=BF16, trb=1, seed=rep, **mid)
    U.check_gemm(hip, DEV, 300, 256, 192, BF16, to=F32, bias=True, residual=True, pad=8, **mid)
    U.check_gemm(hip, DEV, 300, 256, 64, BF16, to=BF16, trb=1, pad=8, **mid)
    U.check_gemm(hip, DEV, 2048, 1024, 1024, BF16, to=F32, bias=True, residual=True, kernel=L.KERNEL_GEMM_MID)   # automatic choice



def test_gemm_dma_kernel_wide_tile(hip):
    """256 x 256 tile (automatic for the plain big GEMMs): QKV-forward-, dgrad-through-W^T- and wgrad-like problems, three times each
    (a mis-counted vmcnt of the two-stage ring shows up as sporadic wrong tiles), ragged M tail, short last k-slice"""
    wide = dict(flags=L.GEMM_DMA_ALWAYS | L.GEMM_WIDE_ALWAYS, kernel=L.KERNEL_GEMM_DMA)
    for rep in range(3):
        U.check_gemm(hip, DEV, 20040, 3072, 512, BF16, to=BF16, bias=True, seed=rep, **wide)
        U.check_gemm(hip, DEV, 20040, 512, 3072, BF16, to=BF16, seed=rep, **wide)
        U.check_gemm(hip, DEV, 20040, 512, 1024, BF16, to=F32, bias=True, seed=rep, **wide)
        U.check_gemm(hip, DEV, 3072, 512, 20032, BF16, to=F32, tra=1, trb=1, seed=rep, **wide)
        U.check_gemm(hip, DEV, 512, 512, 40000 - 64, BF16, to=F32, tra=1, trb=1, seed=rep, **wide)
    U.check_gemm(hip, DEV, 300, 256, 192, BF16, to=F32, bias=True, pad=8, **wide)
    U.check_gemm(hip, DEV, 20040, 512, 512, BF16, to=BF16, bias=True, act=1, **wide)      # fused bias + GELU on the 256-wide tile (MLP-1 forward)
    for xn in (2, 4, 8):                                             # XCD column groups, both tile widths
        narrow = L.GEMM_DMA_ALWAYS | L.GEMM_WIDE_NEVER | L.gemm_xcd_cols(xn)
        U.check_gemm(hip, DEV, 20040, 3072, 512, BF16, to=BF16, bias=True, seed=xn, flags=wide["flags"] | L.gemm_xcd_cols(xn), kernel=L.KERNEL_GEMM_DMA)
        U.check_gemm(hip, DEV, 20040, 3072, 512, BF16, to=BF16, bias=True, seed=xn, flags=narrow, kernel=L.KERNEL_GEMM_DMA)
        U.check_gemm(hip, DEV, 20040, 1024, 512, BF16, to=F32, bias=True, residual=True, seed=xn, flags=narrow, kernel=L.KERNEL_GEMM_DMA)


@pytest.mark.parametrize("C_,dt", [(512, F32), (1024, F32), (512, BF16), (1024, BF16)])
def test_layernorm(hip, C_, dt):
    U.check_layernorm(hip, DEV, 5003, C_, dt)


@pytest.mark.parametrize("dt", [F32, BF16])
def test_attention_vit(hip, dt):
    U.check_attention(hip, DEV, 6, 16, 50, 64, window=50, causal=0, dt=dt)


@pytest.mark.parametrize("dt", [F32, BF16])
def test_attention_single_query(hip, dt):
    """cls-only last ViT layer (one query per image and head, 50 keys): bf16 takes the dedicated kernel, fp32 the wave-per-row one"""
    U.check_attention_single_query(hip, DEV, 5, 16, 50, dt)
    U.check_attention_single_query(hip, DEV, 3, 2, 64, dt, seed=3)


def test_cls_attention(hip):
    """attn_cls.h (r06): the last ViT layer's class-token attention on (q W_k, normalised tokens) — forward, backward and the fp32 class row"""
    U.check_cls_attention(hip, DEV, 37, 16, 50, BF16)
    U.check_cls_attention(hip, DEV, 5, 6, 64, BF16, seed=5, pad=8)
    U.check_cls_attention(hip, DEV, 3, 1, 7, BF16, seed=9)


@pytest.mark.parametrize("T,window,causal,D", [(64, 64, 1, 256), (37, 10, 1, 256), (33, 1, 1, 128), (5, 3, 1, 64), (50, 50, 0, 64), (64, 64, 0, 64), (31, 31, 0, 128)])
def test_attention_f32_mfma(hip, T, window, causal, D):
    """fp32 tensors, Tq == Tk <= 64 (attn_f32.h): one wave per 32-query / 32-key block on the f32 matrix cores, forward + both backward kernels"""
    U.check_attention(hip, DEV, 5, 3, T, D, window=window, causal=causal, dt=F32)


@pytest.mark.parametrize("T", [8, 64, 186])
def test_attention_decoder_causal(hip, T):
    U.check_attention(hip, DEV, 2, 4, T, 256, window=T, causal=1, dt=F32)


@pytest.mark.parametrize("T,window", [(64, 64), (50, 50), (64, 10), (33, 10), (64, 1)])
def test_attention_decoder_mfma(hip, T, window):
    U.check_attention(hip, DEV, 4, 4, T, 256, window=window, causal=1, dt=BF16)


@pytest.mark.parametrize("T,window", [(70, 70), (186, 186), (186, 10), (128, 1), (192, 100), (200, 200), (321, 64), (500, 500), (1000, 10)])
def test_attention_decoder_mfma_long(hip, T, window):
    U.check_attention(hip, DEV, 3, 4, T, 256, window=window, causal=1, dt=BF16)


@pytest.mark.parametrize("window", [1, 5, 10])
def test_attention_band(hip, window):
    U.check_attention(hip, DEV, 2, 4, 64, 256, window=window, causal=1, dt=F32)
    U.check_attention(hip, DEV, 2, 4, 186, 256, window=window, causal=1, dt=BF16)


@pytest.mark.parametrize("T,window,dt", [(64, 64, BF16), (64, 10, BF16), (186, 186, BF16), (186, 10, BF16), (186, 186, F32), (64, 10, F32)])
def test_attention_head_dim_128(hip, T, window, dt):
    """nhead = 8 at hidden 1024 (reference final_experiments.json / *_large configs)"""
    U.check_attention(hip, DEV, 3, 8, T, 128, window=window, causal=1, dt=dt)


@pytest.mark.parametrize("to,bias,act,residual", [(F32, True, 0, True), (BF16, True, 1, False), (BF16, False, 0, False)])
def test_gemm_mx8(hip, to, bias, act, residual):
    """VCAD_FP8 building blocks on the hardware: MXFP8 quantiser bit-exact against the OCP MX rule (incl. the non-saturating
    v_cvt_pk_fp8_f32 behind a clamp), block-scaled fp8 MFMA GEMM against its dequantised operands; ragged M, several k-tiles"""
    U.check_mx8(hip, DEV, 5000, 512, 1024, to, bias=bias, act=act, residual=residual)
    U.check_mx8(hip, DEV, 300, 3072, 512, to, bias=bias, act=act, residual=residual, seed=5)


@pytest.mark.parametrize("T", [50, 64, 7])
def test_attention_vit_bf16x3(hip, T):
    """bf16x3 mode's ViT attention (attn_x3.h): fp32 tensors, hi / lo split operands on the bf16 matrix cores — forward and backward at the GEMMs' error level"""
    U.check_attention(hip, DEV, 3, 2, T, 64, window=T, causal=0, dt=F32, x3=True)


def test_layernorm_pre_split_outputs(hip):
    """bf16x3 mode, r04: the typed outputs of the LayerNorm kernels (forward y, backward dx) written as pre-split hi | lo words for the GEMMs that
    consume them == vcad_op_pack_x3 of the fp32 outputs, word for word"""
    import ctypes as C_
    dev = DEV
    rows, Cn = 37, 512
    x = U.rnd((rows, Cn), dev, seed=1, scale=2.0) + 0.5; g = U.rnd((Cn,), dev, seed=2, scale=0.2) + 1.0; b = U.rnd((Cn,), dev, seed=3, scale=0.1)
    y32 = torch.empty(rows, Cn, device=dev); yp = torch.zeros(rows, Cn, dtype=torch.int32, device=dev); stats = torch.empty(rows, 2, device=dev)
    st = U.stream_of(dev)
    L.check(hip, hip.vcad_op_layernorm_fwd(L.VCAD_F32, 3, Cn, U.ptr(x), Cn, U.ptr(g), U.ptr(b), U.ptr(y32), U.ptr(yp), U.ptr(stats), rows, 1e-5, st), "ln_fwd pk")
    want = torch.empty_like(yp); L.check(hip, hip.vcad_op_pack_x3(U.ptr(y32), U.ptr(want), y32.numel(), st), "pack")
    assert torch.equal(yp.cpu(), want.cpu())
    dy = U.rnd((rows, Cn), dev, seed=4); add = U.rnd((rows, Cn), dev, seed=5)
    dx = torch.empty(rows, Cn, device=dev); dxp = torch.zeros(rows, Cn, dtype=torch.int32, device=dev); dg = torch.empty(Cn, device=dev); db = torch.empty(Cn, device=dev)
    scratch = torch.empty(4 << 20, dtype=torch.float32, device=dev)
    L.check(hip, hip.vcad_op_layernorm_bwd(L.VCAD_F32, 3, Cn, U.ptr(dy), U.ptr(x), Cn, U.ptr(stats), U.ptr(g), U.ptr(add), U.ptr(dx), U.ptr(dxp), U.ptr(dg), U.ptr(db),
                                             rows, U.ptr(scratch), scratch.numel() * 4, st), "ln_bwd pk")
    L.check(hip, hip.vcad_op_pack_x3(U.ptr(dx), U.ptr(want), dx.numel(), st), "pack")
    assert torch.equal(dxp.cpu(), want.cpu())


@pytest.mark.parametrize("ct", [F32, U.X3, BF16])
@pytest.mark.parametrize("trb", [0, 1])
def test_gemm_ragged_rows_with_aligned_operands(hip, ct, trb):
    """r04: a tile that is ragged in the row dimension only, on 16-byte-aligned operands with whole k-tiles, stages through vector loads on clamped
    row indices (A rows past M, B rows past N) and takes the row-wise epilogue with a row bound — the decoder's 2 976-row Linears at the maximum
    horizon (46.5 tiles of 64 rows).  Both tile sizes, fused epilogue, bf16 and fp32 outputs; nothing may be written outside [M, N]."""
    for flags in (L.GEMM_TILE64 | L.GEMM_DMA_NEVER | L.GEMM_MID_NEVER, L.GEMM_TILE128 | L.GEMM_DMA_NEVER | L.GEMM_MID_NEVER):
        U.check_gemm(hip, DEV, 200, 136 if not trb else 192, 128, ct, trb=trb, bias=True, act=2, residual=True, splitk=False, flags=flags, pack_b=(ct == U.X3))
        if ct == BF16:
            U.check_gemm(hip, DEV, 75, 128, 192, ct, to=F32, sa=F32, trb=trb, residual=True, splitk=False, flags=flags)      # fp32 activations, fp32 output (the decoder's residual-stream Linears)


@pytest.mark.parametrize("T,window,D", [(200, 200, 256), (333, 333, 128), (260, 10, 256)])
def test_attention_f32_beyond_192_keys(hip, T, window, D):
    """r04: the fp32 modes' wave-per-row kernels with up to sixteen 64-key pieces per query (horizons up to 1 024; the reference's max_ep_len is 1 000)"""
    U.check_attention(hip, DEV, 1, 2, T, D, window=window, causal=1, dt=F32)
