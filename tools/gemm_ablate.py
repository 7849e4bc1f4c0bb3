"""Ablation of the GEMM K-loop on hardware: which of {global loads, LDS stores, MFMA} bounds it (results are WRONG numerically
by construction; timing only).  python tools/gemm_ablate.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocad_amd import lib as L
lib = L.load_ab()
dev = "cuda:0"; BF = torch.bfloat16
def run(M, N, K, iters=10):
    A = torch.randn(M, K, device=dev).to(BF); B = torch.randn(N, K, device=dev).to(BF); Cm = torch.empty(M, N, dtype=BF, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.vcad_op_gemm(1, 1, 1, 1, 0, 0, p(A), p(B), p(Cm), M, N, K, K, K, N, None, 0, None, N, 1.0, None, 0, 0, None, st)
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
lib.vcad_debug_gemm_stagger.argtypes = [C.c_int]
for shape in [(104000, 3072, 512), (104000, 512, 512), (104000, 512, 1024), (8192, 8192, 8192)]:
    print(shape, "stagger:", " | ".join(f"{n}: {(lib.vcad_debug_gemm_stagger(n), run(*shape))[1]:.0f}us" for n in (0, 1, 2, 3, 4, 6)), flush=True)
lib.vcad_debug_gemm_stagger(0)
for shape in [(104000, 3072, 512), (104000, 512, 512)]:
    for st in (0, 4):
        lib.vcad_debug_gemm_stagger(st)
        print(shape, "stagger", st, " | ".join(f"{name}: {(lib.vcad_debug_gemm_skip(m), run(*shape))[1]:.0f}us" for m, name in ((0, "lds-epilogue"), (16, "direct-epilogue"), (7, "lds-epi only"), (23, "direct-epi only"))), flush=True)
lib.vcad_debug_gemm_skip(0); lib.vcad_debug_gemm_stagger(0)
for shape in []:
    out = []
    for mask, name in [(0, "full"), (8, "no-epilogue"), (7, "barriers+epilogue only"), (15, "barriers only"), (11, "mfma+ldsread only"), (3, "no-gload/store")]:
        lib.vcad_debug_gemm_skip(mask); out.append(f"{name} {run(*shape):.0f}us")
    lib.vcad_debug_gemm_skip(0)
    print(shape, " | ".join(out), flush=True)
