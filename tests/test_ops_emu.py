"""CPU checks of the HIP kernels' indexing logic: the SAME kernel sources compiled with -DVC_EMU run under the
fiber emulator (tests/emu/emu_rt.cpp, test infrastructure only) and are compared with PyTorch fp32 math.
Sizes are tiny — the emulator executes every lane as a fiber."""
import pytest
import torch

import oputil as U
from videocad_amd import lib as L

F32, BF16 = torch.float32, torch.bfloat16


@pytest.fixture(scope="module")
def emu():
    return U.load_emu()


@pytest.fixture(params=[64, 128], autouse=True)
def gemm_tile(request, emu):
    """every test in this file runs with both GEMM block tiles (64x64 and 128x128)"""
    if request.param == 64 and "gemm" not in request.node.name:
        pytest.skip("the block tile only matters to the GEMM tests (run once, with the 128 tile)")
    U.GEMM_FLAGS = L.GEMM_TILE64 if request.param == 64 else L.GEMM_TILE128
    yield request.param
    U.GEMM_FLAGS = 0


@pytest.mark.parametrize("tra,trb", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_f32_layouts(emu, tra, trb):
    U.check_gemm(emu, "cpu", 70, 40, 50, F32, tra=tra, trb=trb, pad=3, bias=True, residual=True, splitk=False)


@pytest.mark.parametrize("ct", [F32, U.X3, BF16])
def test_gemm_column_group_sweep(emu, gemm_tile, ct):
    """register-staged kernel, tile order in column groups (gemm.h n_group; automatic only when B overflows an L2): forced group widths that do and do
    not divide the tile-column count, ragged edges — every tile must still be computed exactly once"""
    if gemm_tile != 64:
        pytest.skip("one tile size is enough (5 x 3 tiles of 64)")
    for g in (1, 2, 3, 4):
        U.check_gemm(emu, "cpu", 170, 300, 40, ct, trb=(g & 1), pad=4, bias=True, splitk=False, flags=L.gemm_ngroup(g), kernel=L.KERNEL_GEMM_REG)


def test_gemm_f32_multi_tile_and_act(emu):
    U.check_gemm(emu, "cpu", 130, 136, 37, F32, act=1, bias=True, pad=4)
    U.check_gemm(emu, "cpu", 16, 5, 64, F32, act=3, bias=True, pad=0)


def test_gemm_f32_splitk(emu):
    U.check_gemm(emu, "cpu", 24, 20, 640, F32, tra=1, trb=1, bias=True, residual=True)


@pytest.mark.parametrize("tra,trb", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("sa,to", [(BF16, BF16), (F32, F32)])
def test_gemm_bf16(emu, tra, trb, sa, to):
    if tra == 1 and to == BF16:
        pytest.skip("wgrad always writes fp32")
    U.check_gemm(emu, "cpu", 72, 48, 136, BF16, sa=sa, to=to, tra=tra, trb=trb, pad=8, bias=True, act=2, splitk=False)


@pytest.mark.parametrize("tra,trb", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_bf16x3_layouts(emu, tra, trb):
    """VCAD_BF16X3: fp32 operands split into hi / lo bf16 planes while staging, three MFMAs per product (gemm.h) — checked against the
    exact product of the UNROUNDED operands"""
    U.check_gemm(emu, "cpu", 70, 40, 100, U.X3, tra=tra, trb=trb, pad=4, bias=True, residual=True, splitk=False, pack_b=not tra)
    U.check_gemm(emu, "cpu", 33, 7, 50, U.X3, tra=tra, trb=trb, pad=1, act=2, splitk=False, pack_b=not tra)       # unaligned rows: element-wise staging path


def test_gemm_bf16x3_multi_tile_splitk(emu):
    U.check_gemm(emu, "cpu", 130, 136, 70, U.X3, act=1, bias=True, pad=4, pack_b=True)
    U.check_gemm(emu, "cpu", 24, 20, 640, U.X3, tra=1, trb=1, bias=True, residual=True)


def test_gemm_bf16_wgrad_f32_b(emu):
    U.check_gemm(emu, "cpu", 40, 24, 70, BF16, sa=BF16, sb=F32, to=F32, tra=1, trb=1, pad=4, splitk=False)


def test_gemm_bf16_unaligned_tail(emu):
    U.check_gemm(emu, "cpu", 33, 7, 100, BF16, sa=F32, to=F32, tra=1, trb=1, pad=1, splitk=False)


@pytest.mark.parametrize("tra,trb,to", [(0, 0, BF16), (0, 0, F32), (0, 1, BF16), (0, 1, F32), (1, 1, F32)])
def test_gemm_dma_kernel(emu, gemm_tile, tra, trb, to):
    """the persistent DMA-fed kernel (gemm_dma.h): M tail, several tiles per workgroup stream, fused epilogue, k-slices"""
    if gemm_tile != 128:
        pytest.skip("tile-size fixture does not apply to the DMA kernel")
    dma = dict(flags=L.GEMM_DMA_ALWAYS, kernel=L.KERNEL_GEMM_DMA)
    M = 264 if tra else 300                       # row-contiguous A needs M % 8 == 0; 300 leaves a ragged last tile
    U.check_gemm(emu, "cpu", M, 256, 192, BF16, sa=BF16, sb=BF16, to=to, tra=tra, trb=trb, pad=8, bias=True, act=(0 if tra else 1),
                 residual=(to == F32), splitk=bool(tra), **dma)
    U.check_gemm(emu, "cpu", M, 256, 192, BF16, sa=BF16, sb=BF16, to=to, tra=tra, trb=trb, pad=8, bias=True, residual=(to == F32), splitk=bool(tra), **dma)
    U.check_gemm(emu, "cpu", M, 128, 64, BF16, sa=BF16, sb=BF16, to=to, tra=tra, trb=trb, pad=0, splitk=False, **dma)      # one k-tile per item
    if not tra and not trb:      # XCD column groups: 4 tile columns over 4 / 2 XCD columns, 9 tile rows over 2 / 4 XCD rows
        for xn in (4, 2):
            U.check_gemm(emu, "cpu", 2100, 512, 128, BF16, sa=BF16, sb=BF16, to=to, pad=8, bias=True, residual=(to == F32), splitk=False,
                         flags=L.GEMM_DMA_ALWAYS | L.gemm_xcd_cols(xn), kernel=L.KERNEL_GEMM_DMA)


@pytest.mark.parametrize("tra,trb,to", [(0, 0, BF16), (0, 0, F32), (1, 1, F32)])
def test_gemm_dma_kernel_wide_tile(emu, gemm_tile, tra, trb, to):
    """the 256 x 256 tile of the persistent kernel (two 64 KiB stages, eight waves of 64 x 128): plain epilogues — bias, k-slice slabs"""
    if gemm_tile != 128:
        pytest.skip("tile-size fixture does not apply to the DMA kernel")
    wide = dict(flags=L.GEMM_DMA_ALWAYS | L.GEMM_WIDE_ALWAYS, kernel=L.KERNEL_GEMM_DMA)
    M = 520 if tra else 600                       # three rows of items, ragged last one
    U.check_gemm(emu, "cpu", M, 512, 192, BF16, sa=BF16, sb=BF16, to=to, tra=tra, trb=trb, pad=8, bias=not tra, splitk=bool(tra), **wide)
    U.check_gemm(emu, "cpu", 264, 256, 64, BF16, sa=BF16, sb=BF16, to=to, tra=tra, trb=trb, pad=0, splitk=False, **wide)      # one k-tile per item
    if not tra:      # fused epilogue without side inputs (bias + GELU: the MLP's first Linear) — column-per-lane form
        U.check_gemm(emu, "cpu", 600, 512, 128, BF16, sa=BF16, sb=BF16, to=to, pad=8, bias=True, act=1, splitk=False, **wide)
        # XCD column groups (forward layout): 2 x 4 XCD grid / automatic over 3 tile rows x 2 tile columns
        for xn in (2, 0):
            U.check_gemm(emu, "cpu", 2100, 512, 128, BF16, sa=BF16, sb=BF16, to=to, pad=8, bias=True, splitk=False,
                         flags=wide["flags"] | L.gemm_xcd_cols(xn), kernel=L.KERNEL_GEMM_DMA)


@pytest.mark.parametrize("wide", [0, 1])
@pytest.mark.parametrize("tra,trb,to", [(0, 0, BF16), (0, 0, F32), (0, 1, BF16), (1, 1, F32)])
def test_gemm_dma_kernel_dynamic_item_claiming(emu, gemm_tile, tra, trb, to, wide):
    """persistent kernel with ticket-drawn items (gemm_dma.h `claim`; what the engine always uses): under the emulator workgroups run one after the
    other, so the first one draws every item of its XCD's range and then steals the other XCDs' — the multi-item stream, the lazy claim in front
    of the prefetch cursor's last k-tile (1, 2, 3 and 10 k-tiles per item), the exhausted-counter path and the counter reset (two launches on the
    same scratch) in one go"""
    if gemm_tile != 128 or (wide and trb and not tra):
        pytest.skip("tile-size fixture does not apply / no wide tile for the tr-read B forward layout")
    fl = L.GEMM_DMA_ALWAYS | L.GEMM_DYNAMIC | (L.GEMM_WIDE_ALWAYS if wide else L.GEMM_WIDE_NEVER)
    for K in (64, 128, 192, 640):
        M = 776 if tra else 790                        # 4 tile rows (ragged last one) x 2-4 tile columns
        U.check_gemm(emu, "cpu", M, 512, K, BF16, sa=BF16, sb=BF16, to=to, tra=tra, trb=trb, pad=8, bias=not tra, residual=(to == F32 and not wide and not tra),
                     splitk=True, flags=fl, kernel=L.KERNEL_GEMM_DMA)
    if not tra and not trb:                            # XCD column groups keep their own counters (no stealing)
        U.check_gemm(emu, "cpu", 2100, 512, 128, BF16, sa=BF16, sb=BF16, to=to, pad=8, bias=True, splitk=True, flags=fl | L.gemm_xcd_cols(2), kernel=L.KERNEL_GEMM_DMA)


@pytest.mark.parametrize("dyn", [0, 1])
@pytest.mark.parametrize("trb,to,wide", [(0, BF16, 1), (0, F32, 1), (0, BF16, 0), (0, F32, 0), (1, BF16, 0)])
def test_gemm_dma_kernel_mini_tiles(emu, gemm_tile, trb, to, wide, dyn):
    """r06: the rows of a mostly empty last round as 64-row mini tiles of the same tile program (gemm_dma.h GdMini) — forced here from the last full tile row on:
    a ragged last piece (790 = 3 x 256 + 22: pieces of 64, 64, 64, 64, 22 rows behind two full tile rows), 1 / 3 / 10 k-tiles per item, plain and fused
    (bias + residual, 128-wide tile) epilogues, static lists and ticket-drawn items, and a problem shorter than one tile (everything is minis)"""
    if gemm_tile != 128:
        pytest.skip("tile-size fixture does not apply to the DMA kernel")
    fl = L.GEMM_DMA_ALWAYS | L.GEMM_MINI_ALWAYS | (L.GEMM_DYNAMIC if dyn else 0) | (L.GEMM_WIDE_ALWAYS if wide else L.GEMM_WIDE_NEVER)
    for K in (64, 192, 640):
        U.check_gemm(emu, "cpu", 790, 512, K, BF16, sa=BF16, sb=BF16, to=to, trb=trb, pad=8, bias=True, residual=(to == F32 and not wide), splitk=False,
                     flags=fl, kernel=L.KERNEL_GEMM_DMA)
    U.check_gemm(emu, "cpu", 200, 256, 128, BF16, sa=BF16, sb=BF16, to=to, trb=trb, pad=0, bias=True, splitk=False, flags=fl, kernel=L.KERNEL_GEMM_DMA)


@pytest.mark.parametrize("dyn", [0, 1])
def test_gemm_dma_kernel_batched_weight_gradients(emu, gemm_tile, dyn):
    """r06: several weight gradients over the same token rows in ONE launch of the persistent kernel (gemm_dma.h GdBatch): a ViT layer's shapes at a small token count
    (512 x 512, 512 x 512, 512 x 1024: 16 tiles), a ragged output height (264 rows: two tile rows, the second with 8 valid rows), a single problem, one / several k-slices"""
    if gemm_tile != 128:
        pytest.skip("tile-size fixture does not apply to the DMA kernel")
    fl = L.GEMM_DYNAMIC if dyn else 0
    U.check_wgrad_batched(emu, "cpu", 1024, [(512, 512), (512, 512), (512, 1024)], flags=fl)      # 16 tiles x 2 k-slices
    U.check_wgrad_batched(emu, "cpu", 576, [(264, 256), (512, 512)], seed=3, flags=fl)             # ragged M; 9 k-tiles -> one slice
    U.check_wgrad_batched(emu, "cpu", 4096, [(256, 256)], seed=5, flags=fl)                        # one problem, 8 k-slices
    U.check_wgrad_batched(emu, "cpu", 2048, [(512, 256), (256, 512), (264, 256), (512, 512)], seed=7, flags=fl)


@pytest.mark.parametrize("trb,to", [(0, BF16), (0, F32), (1, BF16), (1, F32)])
def test_gemm_mid_kernel(emu, gemm_tile, trb, to):
    """the six-stage DMA-ring kernel for mid-size problems (gemm_mid.h): ragged M tail, fewer k-tiles than stages / exactly / more,
    fused bias + ReLU / residual epilogues, padded leading dimensions"""
    if gemm_tile != 128:
        pytest.skip("tile-size fixture does not apply to the DMA-ring kernel")
    for K in (64, 192, 384, 640):                 # 1, 3, 6 (= stages) and 10 k-tiles
        U.check_gemm(emu, "cpu", 300, 256, K, BF16, sa=BF16, sb=BF16, to=to, trb=trb, pad=8, bias=True, act=(2 if to == BF16 else 0),
                     residual=(to == F32), splitk=False, flags=L.GEMM_DMA_NEVER | L.GEMM_MID_ALWAYS, kernel=L.KERNEL_GEMM_MID)


@pytest.mark.parametrize("C_,dt", [(512, F32), (1024, F32), (512, BF16)])
def test_layernorm(emu, C_, dt):
    U.check_layernorm(emu, "cpu", 11, C_, dt)


@pytest.mark.parametrize("dt", [F32, BF16])
def test_attention_single_query(emu, dt):
    """cls-only last ViT layer (one query per image and head, 50 keys): bf16 takes the dedicated kernel, fp32 the wave-per-row one"""
    U.check_attention_single_query(emu, "cpu", 5, 16, 50, dt)
    U.check_attention_single_query(emu, "cpu", 3, 2, 64, dt, seed=3)


@pytest.mark.parametrize("dt", [F32, BF16])
def test_attention_vit_shape(emu, dt):
    U.check_attention(emu, "cpu", 2, 2, 50, 64, window=50, causal=0, dt=dt)


def test_attention_causal_d256(emu):
    U.check_attention(emu, "cpu", 1, 2, 70, 256, window=70, causal=1, dt=F32)     # > 64 keys: NPASS = 3 path


@pytest.mark.parametrize("T,window", [(64, 64), (37, 37), (37, 10), (64, 1), (5, 3)])
def test_attention_decoder_mfma(emu, T, window):
    """bf16, head dim 256, causal (+ band), T <= 64: the matrix-core decoder kernels (attn_mfma.h), forward and backward"""
    U.check_attention(emu, "cpu", 2, 2, T, 256, window=window, causal=1, dt=BF16)


@pytest.mark.parametrize("T,window", [(70, 70), (186, 186), (186, 10), (130, 1), (192, 100), (200, 200), (321, 64), (257, 10)])
def test_attention_decoder_mfma_long(emu, T, window):
    """T > 64, any horizon (the reference allows max_ep_len = 1000): the block-streaming kernels (online-softmax forward, D_i sweep + dQ, dK/dV)"""
    U.check_attention(emu, "cpu", 1, 2, T, 256, window=window, causal=1, dt=BF16)


@pytest.mark.parametrize("T,window,causal,D", [(64, 64, 1, 256), (37, 10, 1, 256), (33, 1, 1, 128), (5, 3, 1, 64), (50, 50, 0, 64), (64, 64, 0, 64), (31, 31, 0, 128)])
def test_attention_f32_mfma(emu, T, window, causal, D):
    """fp32 tensors, Tq == Tk <= 64 (attn_f32.h): one wave per 32-query / 32-key block on the f32 matrix cores, forward + both backward kernels"""
    U.check_attention(emu, "cpu", 2, 2, T, D, window=window, causal=causal, dt=F32)


def test_attention_band(emu):
    U.check_attention(emu, "cpu", 2, 1, 23, 256, window=10, causal=1, dt=F32)
    U.check_attention(emu, "cpu", 1, 1, 9, 256, window=1, causal=1, dt=F32)


@pytest.mark.parametrize("T,window,dt", [(37, 37, BF16), (37, 10, BF16), (70, 70, BF16), (130, 10, BF16), (23, 10, F32), (70, 70, F32)])
def test_attention_head_dim_128(emu, T, window, dt):
    """nhead = 8 at hidden 1024 (reference final_experiments.json / the *_large configs): head dim 128 = two 64-wide chunks in the
    matrix-core decoder kernels, DPL = 2 in the wave-per-row kernels"""
    U.check_attention(emu, "cpu", 1, 2, T, 128, window=window, causal=1, dt=dt)


@pytest.mark.parametrize("to,bias,act,residual", [(F32, True, 0, True), (BF16, True, 1, False), (BF16, False, 0, False)])
def test_gemm_mx8(emu, gemm_tile, to, bias, act, residual):
    """MXFP8 quantiser (bit-exact against the OCP MX rule) + block-scaled fp8 GEMM (against its dequantised operands)"""
    if gemm_tile != 128:
        pytest.skip("tile-size fixture does not apply")
    U.check_mx8(emu, "cpu", 200, 256, 384, to, bias=bias, act=act, residual=residual)


@pytest.mark.parametrize("T", [50, 64, 7])
def test_attention_vit_bf16x3(emu, T):
    """bf16x3 mode's ViT attention (attn_x3.h): fp32 tensors, hi / lo split operands on the bf16 matrix cores — forward and backward at the GEMMs' error level"""
    U.check_attention(emu, "cpu", 3, 2, T, 64, window=T, causal=0, dt=F32, x3=True)


def test_layernorm_pre_split_outputs(emu):
    """bf16x3 mode, r04: the typed outputs of the LayerNorm kernels (forward y, backward dx) written as pre-split hi | lo words for the GEMMs that
    consume them == vcad_op_pack_x3 of the fp32 outputs, word for word"""
    import ctypes as C_
    dev = "cpu"
    rows, Cn = 37, 512
    x = U.rnd((rows, Cn), dev, seed=1, scale=2.0) + 0.5; g = U.rnd((Cn,), dev, seed=2, scale=0.2) + 1.0; b = U.rnd((Cn,), dev, seed=3, scale=0.1)
    y32 = torch.empty(rows, Cn, device=dev); yp = torch.zeros(rows, Cn, dtype=torch.int32, device=dev); stats = torch.empty(rows, 2, device=dev)
    st = U.stream_of(dev)
    L.check(emu, emu.vcad_op_layernorm_fwd(L.VCAD_F32, 3, Cn, U.ptr(x), Cn, U.ptr(g), U.ptr(b), U.ptr(y32), U.ptr(yp), U.ptr(stats), rows, 1e-5, st), "ln_fwd pk")
    want = torch.empty_like(yp); L.check(emu, emu.vcad_op_pack_x3(U.ptr(y32), U.ptr(want), y32.numel(), st), "pack")
    assert torch.equal(yp.cpu(), want.cpu())
    dy = U.rnd((rows, Cn), dev, seed=4); add = U.rnd((rows, Cn), dev, seed=5)
    dx = torch.empty(rows, Cn, device=dev); dxp = torch.zeros(rows, Cn, dtype=torch.int32, device=dev); dg = torch.empty(Cn, device=dev); db = torch.empty(Cn, device=dev)
    scratch = torch.empty(4 << 20, dtype=torch.float32, device=dev)
    L.check(emu, emu.vcad_op_layernorm_bwd(L.VCAD_F32, 3, Cn, U.ptr(dy), U.ptr(x), Cn, U.ptr(stats), U.ptr(g), U.ptr(add), U.ptr(dx), U.ptr(dxp), U.ptr(dg), U.ptr(db),
                                             rows, U.ptr(scratch), scratch.numel() * 4, st), "ln_bwd pk")
    L.check(emu, emu.vcad_op_pack_x3(U.ptr(dx), U.ptr(want), dx.numel(), st), "pack")
    assert torch.equal(dxp.cpu(), want.cpu())


@pytest.mark.parametrize("ct", [F32, U.X3, BF16])
@pytest.mark.parametrize("trb", [0, 1])
def test_gemm_ragged_rows_with_aligned_operands(emu, ct, trb):
    """r04: a tile that is ragged in the row dimension only, on 16-byte-aligned operands with whole k-tiles, stages through vector loads on clamped
    row indices (A rows past M, B rows past N) and takes the row-wise epilogue with a row bound — the decoder's 2 976-row Linears at the maximum
    horizon (46.5 tiles of 64 rows).  Both tile sizes, fused epilogue, bf16 and fp32 outputs; nothing may be written outside [M, N]."""
    for flags in (L.GEMM_TILE64 | L.GEMM_DMA_NEVER | L.GEMM_MID_NEVER, L.GEMM_TILE128 | L.GEMM_DMA_NEVER | L.GEMM_MID_NEVER):
        U.check_gemm(emu, "cpu", 200, 136 if not trb else 192, 128, ct, trb=trb, bias=True, act=2, residual=True, splitk=False, flags=flags, pack_b=(ct == U.X3))
        if ct == BF16:
            U.check_gemm(emu, "cpu", 75, 128, 192, ct, to=F32, sa=F32, trb=trb, residual=True, splitk=False, flags=flags)      # fp32 activations, fp32 output (the decoder's residual-stream Linears)


@pytest.mark.parametrize("T,window,D", [(200, 200, 256), (333, 333, 128), (260, 10, 256)])
def test_attention_f32_beyond_192_keys(emu, T, window, D):
    """r04: the fp32 modes' wave-per-row kernels with up to sixteen 64-key pieces per query (horizons up to 1 024; the reference's max_ep_len is 1 000)"""
    U.check_attention(emu, "cpu", 1, 2, T, D, window=window, causal=1, dt=F32)


def test_cls_attention(emu):
    """attn_cls.h under the emulator: the canonical head / token counts and a short model (H not a multiple of the heads-per-wave groups, 64 tokens)"""
    U.check_cls_attention(emu, "cpu", 2, 16, 50, BF16)
    U.check_cls_attention(emu, "cpu", 1, 6, 64, BF16, seed=5, pad=8)
