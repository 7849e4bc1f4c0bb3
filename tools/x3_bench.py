"""Per-shape timing of the bf16x3 (VCAD_BF16X3) and f32 GEMMs through the C ABI at the shapes of one C2 train step, and of the fp32
attention kernels.  Usage: python tools/x3_bench.py [gemm|attn]"""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocad_amd import lib as L

if os.environ.get("VCAD_LIB"):          # an experiment build of the library (timing only)
    L._lib = L.declare(C.CDLL(os.environ["VCAD_LIB"]))
lib = L.load()
dev = "cuda:0"
scratch = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def timeit(call, iters=10):
    for _ in range(2):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def gemm(name, M, N, K, ct=2, tra=0, trb=0, bias=False, res=False, pk=False):
    A = torch.randn((K, M) if tra else (M, K), device=dev)
    B = torch.randn((K, N) if trb else (N, K), device=dev)
    Cm = torch.empty(M, N, device=dev)
    bias_t = torch.randn(N, device=dev) if bias else None
    res_t = torch.randn(M, N, device=dev) if res else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    sb = 0
    if pk:        # B pre-split into hi | lo words (what the engine's weight shadow holds)
        Bp = torch.empty(B.shape, dtype=torch.int32, device=dev)
        assert lib.vcad_op_pack_x3(p(B), p(Bp), B.numel(), st) == 0
        B, sb = Bp, 3

    def call():
        rc = lib.vcad_op_gemm(ct, 0, sb, 0, tra, trb, p(A), p(B), p(Cm), M, N, K, A.shape[1], B.shape[1], N, p(bias_t), 0, p(res_t), N, 1.0,
                              p(scratch), scratch.numel(), 0, None, st)
        assert rc == 0, lib.vcad_last_error()
    ms = timeit(call)
    print(f"{name:28s} ct={('x3pk' if pk else 'x3') if ct == 2 else 'f32'} M={M:6d} N={N:5d} K={K:6d} tra={tra} trb={trb} {ms*1e3:8.1f} us  {2.0*M*N*K/(ms*1e-3)/1e12:7.1f} TF/s", flush=True)


def attn(name, B, H, T, D, window, causal):
    qkv = torch.randn(B, T, 3, H, D, device=dev); ld = 3 * H * D
    o = torch.empty(B, T, H, D, device=dev); lse = torch.empty(B, H, T, device=dev); delta = torch.empty(B, H, T, device=dev)
    do = torch.randn(B, T, H, D, device=dev); dqkv = torch.zeros_like(qkv)
    base, db = qkv.data_ptr(), dqkv.data_ptr(); es = 4
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    sc = 1.0 / math.sqrt(D)
    f = lambda: lib.vcad_op_attention_fwd(0, D, C.c_void_p(base), C.c_void_p(base + H * D * es), C.c_void_p(base + 2 * H * D * es), p(o), ld, ld, ld, H * D,
                                           p(lse), B, H, T, T, window, causal, sc, st)
    g = lambda: lib.vcad_op_attention_bwd_o(0, D, C.c_void_p(base), C.c_void_p(base + H * D * es), C.c_void_p(base + 2 * H * D * es), p(o), H * D, p(do), ld, ld, ld,
                                             H * D, p(lse), p(delta), C.c_void_p(db), C.c_void_p(db + H * D * es), C.c_void_p(db + 2 * H * D * es), ld, ld, ld,
                                             B, H, T, T, window, causal, sc, st)
    mf, mb = timeit(f), timeit(g)
    fl = 4.0 * B * H * T * T * D
    print(f"{name:28s} B={B} H={H} T={T} D={D} w={window}: fwd {mf*1e3:8.1f} us ({fl/(mf*1e-3)/1e12:5.1f} TF/s)  bwd {mb*1e3:8.1f} us ({2.5*fl/(mb*1e-3)/1e12:5.1f} TF/s)", flush=True)


R = 2048 * 50
what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("gemm", "all"):
    for nm, M, N, K, kw in (("QKV fwd", R, 3072, 512, {}), ("out-proj fwd (+res)", R, 512, 1024, dict(bias=True, res=True)), ("MLP fwd", R, 512, 512, dict(bias=True)),
                            ("dqkv dgrad", R, 512, 3072, dict(trb=1)), ("dao dgrad", R, 1024, 512, dict(trb=1))):
        gemm(nm, M, N, K, 2, pk=True, **kw)
    for ct in (2, 0):
        gemm("QKV fwd", R, 3072, 512, ct)
        gemm("out-proj fwd (+res)", R, 512, 1024, ct, bias=True, res=True)
        gemm("MLP fwd", R, 512, 512, ct, bias=True)
        gemm("dqkv dgrad", R, 512, 3072, ct, trb=1)
        gemm("dao dgrad", R, 1024, 512, ct, trb=1)
        gemm("QKV wgrad", 3072, 512, R, ct, tra=1, trb=1)
        gemm("MLP wgrad", 512, 512, R, ct, tra=1, trb=1)
        gemm("decoder Linear", 2048, 1024, 1024, ct, bias=True)
        gemm("heads", 2048, 6000, 1024, ct, bias=True)
if what in ("attn", "all"):
    attn("ViT", 2048, 16, 50, 64, 50, 0)
    attn("decoder self", 32, 4, 64, 256, 64, 1)
    attn("decoder cross (band 10)", 32, 4, 64, 256, 10, 1)
