"""Occupied-CU A/B of the persistent GEMM's item assignment: a background kernel pins N CUs (one 1024-thread / 160 KiB-LDS
workgroup each, A/B build: vcad_debug_hog) on a second stream while the headline GEMM shapes run on the first — static item lists (r02: the
workgroups that find no CU run as a second round) against ticket-drawn items (r03).  Usage: python tools/gemm_hog_ab.py [n_hog ...]"""
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
scratch = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
hog_stream = torch.cuda.Stream()


def run(name, M, N, K, to=BF, tra=0, trb=0, bias=False, dynamic=False, n_hog=0, iters=10):
    A = torch.randn((K, M) if tra else (M, K), device=dev).to(BF)
    B = torch.randn((K, N) if trb else (N, K), device=dev).to(BF)
    Cm = torch.empty(M, N, dtype=to, device=dev)
    bias_t = torch.randn(N, device=dev) if bias else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl = L.GEMM_DMA_ALWAYS | (L.GEMM_DYNAMIC if dynamic else 0)

    def call():
        rc = lib.vcad_op_gemm(1, 1, 1, TD[to], tra, trb, p(A), p(B), p(Cm), M, N, K, A.shape[1], B.shape[1], N, p(bias_t), 0, None, N, 1.0,
                              p(scratch), scratch.numel(), fl, None, st)
        assert rc == 0, lib.vcad_last_error()
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    if n_hog:       # the hog outlives the timed launches
        assert lib.vcad_debug_hog(n_hog, 60000, p(sink), C.c_void_p(hog_stream.cuda_stream)) == 0      # 60 ms
        import time; time.sleep(0.003)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    torch.cuda.synchronize()
    print(f"{name:22s} hog={n_hog:3d} CUs  {'tickets' if dynamic else 'static ':8s} {ms*1e3:8.1f} us  {2.0*M*N*K/(ms*1e-3)/1e12:7.1f} TF/s", flush=True)
    return ms


R = 104000
hogs = [int(a) for a in sys.argv[1:]] or [0, 16, 32]
for name, dims, kw in [("qkv fwd", (R, 3072, 512), dict(bias=True)), ("dqkv dgrad W^T", (R, 512, 3072), {}), ("qkv wgrad", (3072, 512, R), dict(to=F32, tra=1, trb=1)),
                       ("out fwd (128 tile)", (R, 512, 1024), dict(to=F32, bias=True))]:
    for n in hogs:
        for rnd in range(2):
            for dyn in (False, True):
                run(name, *dims, dynamic=dyn, n_hog=n, **kw)
