"""Ablations of the persistent DMA GEMM (debug_skip bits kept in the kernel: 32 = conservative vmcnt (epilogue stores not counted as
younger), 64 = no epilogue).  The no-DMA / no-MFMA bits used for the decomposition quoted in DESIGN.md §4 (63 + 84 + 210 + 155 us on the
QKV shape) were removed from the k-tile loop again: every branch there costs issue slots."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocad_amd import lib as L
lib = L.load_ab(); dev = "cuda:0"
BF, F32 = torch.bfloat16, torch.float32
TD = {F32: 0, BF: 1}
scratch = torch.empty(64 << 20, dtype=torch.uint8, device=dev)

def run(name, M, N, K, to=BF, tra=0, trb=0, bias=False, iters=10, masks=(0, 32, 64)):
    A = torch.randn((K, M) if tra else (M, K), device=dev).to(BF)
    B = torch.randn((K, N) if trb else (N, K), device=dev).to(BF)
    Cm = torch.empty(M, N, dtype=to, device=dev)
    bias_t = torch.randn(N, device=dev) if bias else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for mask in masks:
        lib.vcad_debug_gemm_skip(mask)
        def call():
            rc = lib.vcad_op_gemm(1, 1, 1, TD[to], tra, trb, p(A), p(B), p(Cm), M, N, K, A.shape[1], B.shape[1], N, p(bias_t), 0, None, N, 1.0, p(scratch), scratch.numel(), 0, None, st)
            assert rc == 0
        for _ in range(2): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): call()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:16s} skip={mask:3d}  {e0.elapsed_time(e1)/iters*1e3:8.1f} us", flush=True)
    lib.vcad_debug_gemm_skip(0)

R = 104000
run("qkv fwd", R, 3072, 512, bias=True)
run("dqkv dgrad", R, 512, 3072, trb=1)
run("qkv wgrad", 3072, 512, R, to=F32, tra=1, trb=1)
