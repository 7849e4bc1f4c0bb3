"""A few launches of selected GEMM shapes through both kernels (register-staged / persistent DMA) — run under rocprofv3 --pmc.
Usage: python tools/gemm_pmc.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocad_amd import lib as L

lib = L.load_ab()
dev = "cuda:0"
BF, F32 = torch.bfloat16, torch.float32
TD = {F32: 0, BF: 1}
scratch = torch.empty(64 << 20, dtype=torch.uint8, device=dev)


def run(M, N, K, to=BF, tra=0, trb=0, bias=False, iters=3):
    A = torch.randn((K, M) if tra else (M, K), device=dev).to(BF)
    B = torch.randn((K, N) if trb else (N, K), device=dev).to(BF)
    Cm = torch.empty(M, N, dtype=to, device=dev)
    bias_t = torch.randn(N, device=dev) if bias else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for mode in (0, -1):
        lib.vcad_debug_gemm_dma(mode)
        for _ in range(iters):
            rc = lib.vcad_op_gemm(1, 1, 1, TD[to], tra, trb, p(A), p(B), p(Cm), M, N, K, A.shape[1], B.shape[1], N, p(bias_t), 0,
                                  None, N, 1.0, p(scratch), scratch.numel(), 0, None, st)
            assert rc == 0, lib.vcad_last_error()
        torch.cuda.synchronize()


R = 104000
run(R, 3072, 512, bias=True)                       # QKV forward
run(R, 512, 3072)                                  # dqkv dgrad through the transposed weight shadow (k-contiguous)
run(3072, 512, R, to=F32, tra=1, trb=1)            # QKV wgrad
run(8192, 8192, 8192)                              # reference point
