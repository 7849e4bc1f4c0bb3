"""The fp16-storage build of the kernel library (libvcad_hip_f16.so: the same sources compiled with -DVC_H16, include/vcad.h vcad_storage_format):
per-kernel parity on a real MI355X against plain PyTorch fp64 math — the checks of test_ops_gpu.py with fp16 tensors, at tolerances scaled by the
rounding step (oputil.EPS16: fp16 rounds 8x finer than bf16)."""
import pytest
import torch

import oputil as U
from videocad_amd import lib as L

pytestmark = pytest.mark.gpu
F32, F16 = torch.float32, torch.float16
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    lib = L.load("f16")
    assert lib.vcad_storage_format() == b"f16" and L.load("bf16").vcad_storage_format() == b"bf16"
    return lib


@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("tra,trb", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("sa,to", [(F16, F16), (F16, F32), (F32, F16), (F32, F32)])
def test_gemm_f16(hip, tile, tra, trb, sa, to):
    if tra == 1 and to == F16:
        pytest.skip("wgrad always writes fp32")
    U.GEMM_FLAGS = L.GEMM_TILE64 if tile == 64 else L.GEMM_TILE128
    try:
        U.check_gemm(hip, DEV, 520, 264, 392, F16, sa=sa, to=to, tra=tra, trb=trb, pad=8, bias=True, act=2, splitk=False)
        U.check_gemm(hip, DEV, 33, 7, 100, F16, sa=sa, to=to, tra=tra, trb=trb, pad=1, splitk=False)
    finally:
        U.GEMM_FLAGS = 0


def test_gemm_f16_big_and_f32_sources(hip):
    U.check_gemm(hip, DEV, 5000, 3072, 512, F16, to=F16)
    U.check_gemm(hip, DEV, 3072, 512, 20000, F16, sa=F16, to=F32, tra=1, trb=1)
    U.check_gemm(hip, DEV, 4096, 512, 1024, F16, sa=F32, to=F32, trb=1, residual=True)
    for sa in (F16, F32):
        for sb in (F16, F32):
            U.check_gemm(hip, DEV, 260, 136, 1000, F16, sa=sa, sb=sb, to=F32, tra=1, trb=1, pad=4)


def test_gemm_f16_persistent_kernels(hip):
    """both tiles of the persistent DMA-fed kernel and the six-stage ring kernel: the operand path is format-blind (bytes through the LDS-DMA ring), the
    MFMA is v_mfma_f32_32x32x16_f16 and the epilogues convert with v_cvt_pk_f16_f32 / v_cvt_f32_f16"""
    dma = dict(flags=L.GEMM_DMA_ALWAYS, kernel=L.KERNEL_GEMM_DMA)
    wide = dict(flags=L.GEMM_DMA_ALWAYS | L.GEMM_WIDE_ALWAYS, kernel=L.KERNEL_GEMM_DMA)
    mid = dict(flags=L.GEMM_DMA_NEVER | L.GEMM_MID_ALWAYS, kernel=L.KERNEL_GEMM_MID)
    for rep in range(2):
        U.check_gemm(hip, DEV, 20040, 512, 512, F16, to=F16, bias=True, act=1, seed=rep, **dma)
        U.check_gemm(hip, DEV, 20040, 512, 1024, F16, to=F32, bias=True, residual=True, seed=rep, **dma)
        U.check_gemm(hip, DEV, 20040, 512, 3072, F16, to=F16, trb=1, seed=rep, **dma)
        U.check_gemm(hip, DEV, 3072, 512, 20032, F16, to=F32, tra=1, trb=1, seed=rep, **dma)
        U.check_gemm(hip, DEV, 20040, 3072, 512, F16, to=F16, bias=True, seed=rep, **wide)
        U.check_gemm(hip, DEV, 20040, 512, 3072, F16, to=F16, seed=rep, **wide)
        U.check_gemm(hip, DEV, 512, 512, 40000 - 64, F16, to=F32, tra=1, trb=1, seed=rep, **wide)
        U.check_gemm(hip, DEV, 2048, 1024, 1024, F16, to=F32, bias=True, residual=True, seed=rep, **mid)
        U.check_gemm(hip, DEV, 2976, 3072, 1024, F16, to=F16, bias=True, seed=rep, **mid)
        U.check_gemm(hip, DEV, 2048, 1024, 3072, F16, to=F32, trb=1, residual=True, seed=rep, **mid)


@pytest.mark.parametrize("C_", [512, 1024])
def test_layernorm_f16(hip, C_):
    U.check_layernorm(hip, DEV, 5003, C_, F16)


def test_attention_vit_f16(hip):
    U.check_attention(hip, DEV, 6, 16, 50, 64, window=50, causal=0, dt=F16)
    U.check_attention_single_query(hip, DEV, 5, 16, 50, F16)
    U.check_cls_attention(hip, DEV, 37, 16, 50, F16)
    U.check_cls_attention(hip, DEV, 5, 6, 64, F16, seed=5, pad=8)


@pytest.mark.parametrize("T,window,D", [(64, 64, 256), (64, 10, 256), (33, 10, 256), (186, 186, 256), (186, 10, 256), (500, 500, 256), (64, 64, 128), (186, 10, 128)])
def test_attention_decoder_f16(hip, T, window, D):
    U.check_attention(hip, DEV, 3, 4, T, D, window=window, causal=1, dt=F16)


def test_bf16_only_modes_are_rejected(hip):
    """bf16x3 and the fp8 forward live in the bf16 build; each library refuses the other's engine dtypes"""
    from videocad_amd.engine import make_config
    from oracle import restatement as O
    import ctypes as C
    keys = ("hidden_size", "nhead", "num_decoder_layers", "dim_feedforward", "window_size", "act_dim", "num_classes", "num_params",
            "num_params_values", "max_ep_len", "vit_dim", "vit_depth", "vit_heads", "vit_dim_head", "vit_mlp", "image_size", "patch_size")
    for lib, bad in ((hip, L.VCAD_BF16), (hip, L.VCAD_BF16X3), (hip, L.VCAD_F32), (L.load("bf16"), L.VCAD_F16)):
        cfg = make_config(dtype=bad, **{k: O.CANONICAL_CONFIG[k] for k in keys})
        h = C.c_void_p()
        assert lib.vcad_engine_create(C.byref(cfg), C.byref(h)) != 0
        assert b"build" in lib.vcad_last_error()
