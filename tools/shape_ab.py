"""Isolated timings of single GEMM shapes of the C2 step through the product library (vcad_op_gemm), under the per-call kernel-selection flags:
which kernel family / tile a shape SHOULD take.  Each variant is timed interleaved (A, B, C, A, B, C, ...), median of the rounds, operands
re-randomised between rounds so nothing stays in L2 by accident of the loop.   python tools/shape_ab.py [name ...]"""
import ctypes as C
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocad_amd import lib as L

lib = L.load()
dev = torch.device("cuda:0")
BF, F32 = L.VCAD_BF16, L.VCAD_F32
R = 2048 * 50
# name: (M, N, K, tra, trb, out dtype)   — bf16 operands
SHAPES = {
    "mlp_wgrad 512x512xR": (512, 512, R, 1, 1, F32),
    "ow_wgrad 512x1024xR": (512, 1024, R, 1, 1, F32),
    "qkv_wgrad 3072x512xR": (3072, 512, R, 1, 1, F32),
    "dec_fwd 2048x1024x1024": (2048, 1024, 1024, 0, 0, BF),
    "dec_dgrad 2048x1024x1024 (W row-major)": (2048, 1024, 1024, 0, 1, BF),
    "dec_fwd_qkv 2048x3072x1024": (2048, 3072, 1024, 0, 0, BF),
}
VARIANTS = {"auto": 0, "dma_never": L.GEMM_DMA_NEVER | L.GEMM_MID_NEVER, "dma_always": L.GEMM_DMA_ALWAYS, "dma_wide": L.GEMM_DMA_ALWAYS | L.GEMM_WIDE_ALWAYS,
            "dma_narrow": L.GEMM_DMA_ALWAYS | L.GEMM_WIDE_NEVER, "mid_always": L.GEMM_MID_ALWAYS | L.GEMM_DMA_NEVER, "tile128": L.GEMM_TILE128 | L.GEMM_DMA_NEVER | L.GEMM_MID_NEVER,
            "tile64": L.GEMM_TILE64 | L.GEMM_DMA_NEVER | L.GEMM_MID_NEVER}
FAM = {0: "-", 1: "dma", 2: "reg", 3: "mid", 4: "grouped"}


def run(name):
    M, N, K, tra, trb, to = SHAPES[name]
    a_shape = (K, M) if tra else (M, K)
    b_shape = (K, N) if trb else (N, K)
    A = torch.randn(a_shape, device=dev).bfloat16(); B = torch.randn(b_shape, device=dev).bfloat16()
    Cc = torch.empty(M, N, device=dev, dtype=torch.float32 if to == F32 else torch.bfloat16)
    scratch = torch.empty(64 << 18, device=dev)          # 64 MB
    big = torch.empty(512 << 20, device=dev, dtype=torch.uint8)      # L2 / MALL flusher
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = {}
    for rnd in range(7):
        for vn, fl in VARIANTS.items():
            big.fill_(rnd)
            tag = C.c_int(0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.vcad_op_gemm(BF, BF, BF, to, tra, trb, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cc.data_ptr()), M, N, K,
                                  a_shape[1], b_shape[1], N, None, 0, None, 0, 1.0, C.c_void_p(scratch.data_ptr()), scratch.numel() * 4, fl, C.byref(tag), st)
            e1.record(); torch.cuda.synchronize()
            if rc:
                res.setdefault(vn, []).append((float("nan"), "err")); continue
            res.setdefault(vn, []).append((e0.elapsed_time(e1) * 1e3, FAM.get(tag.value, "?")))
    gf = 2.0 * M * N * K / 1e9
    print(f"## {name}  ({gf:.1f} GFLOP)")
    for vn, v in res.items():
        ts = sorted(t for t, _ in v[1:])
        med = ts[len(ts) // 2]
        print(f"  {vn:11s} {v[0][1]:8s} {med:8.1f} us   {gf / med * 1e3 if med == med else 0:7.1f} TF/s")


for n in (sys.argv[1:] or SHAPES):
    run(n)
