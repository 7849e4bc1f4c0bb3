"""r06 probe: what does the GPU do with an HBM-bound pass on one stream while the persistent GEMM holds every CU on another?

The frame ViT's launches are a chain of MFMA / LDS-bound GEMMs (2.7 TB/s of HBM) and HBM-bound passes (LayerNorm, attention, activation: 5-6 TB/s,
no matrix work).  Two half-batches on two streams would put a pass of one half beside a GEMM of the other — if the hardware co-schedules them
(229 VGPRs x 2 waves per SIMD leave 48 registers per lane: ln_fwd's 44 fit, ln_bwd's 88 do not).  This script times, at the HALF-batch shapes
(R = 51 200 token rows): each op alone, the pair back to back on one stream, and the pair on two streams.      python tools/overlap_ops.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocad_amd import lib as L

lib = L.load()
dev = "cuda:0"
BF, F32 = torch.bfloat16, torch.float32
TD = {F32: 0, BF: 1}
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
R = int(os.environ.get("R", 51200))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def mk_gemm(M, N, K, to=BF, tra=0, trb=0, bias=False):
    A = torch.randn((K, M) if tra else (M, K), device=dev).to(BF)
    B = torch.randn((K, N) if trb else (N, K), device=dev).to(BF)
    Cm = torch.empty(M, N, dtype=to, device=dev)
    bias_t = torch.randn(N, device=dev) if bias else None
    scratch = torch.empty(128 << 20, dtype=torch.uint8, device=dev)

    def call(st):
        rc = lib.vcad_op_gemm(1, 1, 1, TD[to], tra, trb, p(A), p(B), p(Cm), M, N, K, A.shape[1], B.shape[1], N, p(bias_t), 0, None, N, 1.0,
                              p(scratch), scratch.numel(), 0, None, C.c_void_p(st.cuda_stream))
        assert rc == 0, lib.vcad_last_error()
    return call


def mk_ln_fwd(rows, Cc=512):
    x = torch.randn(rows, Cc, device=dev); g = torch.ones(Cc, device=dev); b = torch.zeros(Cc, device=dev)
    y = torch.empty(rows, Cc, dtype=BF, device=dev); st_ = torch.empty(rows, 2, device=dev)

    def call(st):
        rc = lib.vcad_op_layernorm_fwd(0, 1, Cc, p(x), Cc, p(g), p(b), None, p(y), p(st_), rows, 1e-5, C.c_void_p(st.cuda_stream))
        assert rc == 0, lib.vcad_last_error()
    return call


def mk_ln_bwd(rows, Cc=512):
    x = torch.randn(rows, Cc, device=dev); g = torch.ones(Cc, device=dev); dy = torch.randn(rows, Cc, device=dev).to(BF)
    st_ = torch.rand(rows, 2, device=dev) + 0.5; add = torch.randn(rows, Cc, device=dev); dx = torch.empty(rows, Cc, device=dev)
    dxt = torch.empty(rows, Cc, dtype=BF, device=dev); dg = torch.empty(Cc, device=dev); db = torch.empty(Cc, device=dev)
    scr = torch.empty(64 << 20, dtype=torch.uint8, device=dev)

    def call(st):
        rc = lib.vcad_op_layernorm_bwd(1, 1, Cc, p(dy), p(x), Cc, p(st_), p(g), p(add), p(dx), p(dxt), p(dg), p(db), rows, p(scr), scr.numel(), C.c_void_p(st.cuda_stream))
        assert rc == 0, lib.vcad_last_error()
    return call


def mk_attn(frames, bwd):
    H, Dh, Tk = 16, 64, 50
    inner = H * Dh
    qkv = torch.randn(frames * Tk, 3 * inner, device=dev).to(BF); o = torch.empty(frames * Tk, inner, dtype=BF, device=dev)
    lse = torch.zeros(frames * H * Tk, device=dev); delta = torch.empty(frames * H * Tk, device=dev)
    do = torch.randn(frames * Tk, inner, device=dev).to(BF); dqkv = torch.empty_like(qkv)
    es = 2
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + inner * es, qkv.data_ptr() + 2 * inner * es
    dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + inner * es, dqkv.data_ptr() + 2 * inner * es
    vp = C.c_void_p

    def fwd(st):
        rc = lib.vcad_op_attention_fwd(1, Dh, vp(q), vp(k), vp(v), p(o), 3 * inner, 3 * inner, 3 * inner, inner, p(lse), frames, H, Tk, Tk, Tk, 0, 0.125, C.c_void_p(st.cuda_stream))
        assert rc == 0, lib.vcad_last_error()

    def bw(st):
        rc = lib.vcad_op_attention_bwd(1, Dh, vp(q), vp(k), vp(v), p(do), 3 * inner, 3 * inner, 3 * inner, inner, p(lse), p(delta), vp(dq), vp(dk), vp(dv),
                                       3 * inner, 3 * inner, 3 * inner, frames, H, Tk, Tk, Tk, 0, 0.125, C.c_void_p(st.cuda_stream))
        assert rc == 0, lib.vcad_last_error()
    fwd(s1); torch.cuda.synchronize()
    return bw if bwd else fwd


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    # both streams must be done before the closing event
    cur = torch.cuda.current_stream()
    cur.wait_stream(s1); cur.wait_stream(s2)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def pair(name, a, b):
    """a: the GEMM (stream 1), b: the pass (stream 2)"""
    def alone_a():
        s1.wait_stream(torch.cuda.current_stream()); a(s1)
    def alone_b():
        s2.wait_stream(torch.cuda.current_stream()); b(s2)
    def serial():
        s1.wait_stream(torch.cuda.current_stream()); a(s1); b(s1)
    def both():
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream()); a(s1); b(s2)
    ta, tb, ts, tc = timed(alone_a), timed(alone_b), timed(serial), timed(both)
    print(f"{name:44s} gemm {ta:7.1f}  pass {tb:7.1f}  one stream {ts:7.1f}  two streams {tc:7.1f} us   hidden {ts - tc:6.1f} us = {(ts - tc) / tb * 100:5.1f} % of the pass", flush=True)


frames = R // 50
print(f"R = {R} token rows ({frames} frames)")
g_qkv = mk_gemm(R, 3072, 512, bias=True)
g_dqkv = mk_gemm(R, 512, 3072)
g_wqkv = mk_gemm(3072, 512, R, to=F32, tra=1, trb=1)
g_out = mk_gemm(R, 512, 1024, bias=True)
g_w1 = mk_gemm(R, 512, 512, bias=True)
ln_f, ln_b = mk_ln_fwd(R), mk_ln_bwd(R)
at_f, at_b = mk_attn(frames, False), mk_attn(frames, True)
pair("QKV forward   ||  ln_fwd", g_qkv, ln_f)
pair("QKV forward   ||  attention forward", g_qkv, at_f)
pair("to_out forward ||  ln_fwd", g_out, ln_f)
pair("W1 forward    ||  attention forward", g_w1, at_f)
pair("dqkv dgrad    ||  ln_bwd", g_dqkv, ln_b)
pair("dqkv dgrad    ||  attention backward", g_dqkv, at_b)
pair("QKV wgrad     ||  ln_bwd", g_wqkv, ln_b)
pair("QKV wgrad     ||  attention backward", g_wqkv, at_b)
pair("QKV forward   ||  QKV forward (two GEMMs)", g_qkv, g_qkv)
pair("ln_fwd        ||  attention forward (two passes)", ln_f, at_f)
