"""A few launches of the bf16x3 GEMM shapes and the fp32 ViT attention — run under rocprofv3 --pmc (tools/x3_pmc.sh)."""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocad_amd import lib as L
lib = L.load(); dev = "cuda:0"
scratch = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def gemm(M, N, K, tra=0, trb=0, iters=2):
    A = torch.randn((K, M) if tra else (M, K), device=dev); B = torch.randn((K, N) if trb else (N, K), device=dev); Cm = torch.empty(M, N, device=dev)
    for _ in range(iters):
        assert lib.vcad_op_gemm(2, 0, 0, 0, tra, trb, p(A), p(B), p(Cm), M, N, K, A.shape[1], B.shape[1], N, None, 0, None, N, 1.0, p(scratch), scratch.numel(), 0, None, st) == 0
    torch.cuda.synchronize()


R = 102400
gemm(R, 3072, 512); gemm(R, 512, 512); gemm(R, 512, 3072, trb=1); gemm(3072, 512, R, tra=1, trb=1)
B, H, T, D = 2048, 16, 50, 64
qkv = torch.randn(B, T, 3, H, D, device=dev); ld = 3 * H * D; o = torch.empty(B, T, H, D, device=dev); lse = torch.empty(B, H, T, device=dev)
delta = torch.empty(B, H, T, device=dev); do = torch.randn(B, T, H, D, device=dev); dqkv = torch.zeros_like(qkv); base, db, es = qkv.data_ptr(), dqkv.data_ptr(), 4
for _ in range(2):
    lib.vcad_op_attention_fwd(0, D, C.c_void_p(base), C.c_void_p(base + H * D * es), C.c_void_p(base + 2 * H * D * es), p(o), ld, ld, ld, H * D, p(lse), B, H, T, T, T, 0, 0.125, st)
    lib.vcad_op_attention_bwd_o(0, D, C.c_void_p(base), C.c_void_p(base + H * D * es), C.c_void_p(base + 2 * H * D * es), p(o), H * D, p(do), ld, ld, ld, H * D, p(lse), p(delta),
                                C.c_void_p(db), C.c_void_p(db + H * D * es), C.c_void_p(db + 2 * H * D * es), ld, ld, ld, B, H, T, T, T, 0, 0.125, st)
torch.cuda.synchronize()
