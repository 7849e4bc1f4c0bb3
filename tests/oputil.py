"""Shared op-level parity checks: the same functions drive the CPU emulator build (tests/emu/libvcad_emu.so,
host pointers) and the real HIP library (device pointers), comparing each kernel with plain PyTorch fp32 math."""
import contextlib
import ctypes as C
import math
import os
import subprocess

import torch

from videocad_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_PATH = os.path.join(ROOT, "tests", "emu", "libvcad_emu.so")
EMU_PATH_F16 = os.path.join(ROOT, "tests", "emu", "libvcad_emu_f16.so")
X3 = "bf16x3"            # GEMM compute type of the VCAD_BF16X3 mode: fp32 tensors, hi/lo-split bf16 MFMAs
TD = {torch.float32: L.VCAD_F32, torch.bfloat16: L.VCAD_BF16, X3: L.VCAD_BF16X3,
      torch.float16: L.VCAD_BF16}      # op-level type code 1 = "the 16-bit storage type of the library": fp16 tensors go to libvcad_hip_f16.so (L.load("f16"))
S16 = (torch.bfloat16, torch.float16)
EPS16 = {torch.bfloat16: 1.0, torch.float16: 0.125}      # rounding step relative to bf16's (the 16-bit tolerances below are bf16's, scaled)


def load_emu(fmt="bf16"):
    """the host-emulator build of the kernel sources (tests/emu/), in either 16-bit storage format; rebuilt when a source is newer"""
    path = EMU_PATH if fmt == "bf16" else EMU_PATH_F16
    if not os.path.exists(path) or os.path.getmtime(path) < max(
            os.path.getmtime(os.path.join(ROOT, "videocad_amd", "csrc", f)) for f in os.listdir(os.path.join(ROOT, "videocad_amd", "csrc"))
            if f.endswith((".h", ".hip"))):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "videocad_amd", "csrc"), "emu"])
    lib = L.declare(C.CDLL(path))
    assert lib.vcad_storage_format().decode() == fmt
    return lib


_EMU = {}


@contextlib.contextmanager
def emulated(fmt="bf16"):
    """Route `videocad_amd.lib.load(fmt)` to the host-emulator build for the duration (tests only — the product has no such
    switch: it loads csrc/libvcad_hip*.so or raises)."""
    if fmt not in _EMU:
        _EMU[fmt] = load_emu(fmt)
    attr = "_lib" if fmt == "bf16" else "_lib_f16"
    old = getattr(L, attr)
    setattr(L, attr, _EMU[fmt])
    try:
        yield _EMU[fmt]
    finally:
        setattr(L, attr, old)


LABEL_W = [0.04332685213392362, 0.02915898563179938, 0.267566828114559, 0.6005346809501417, 0.05941265316957628]
# ^ tests/golden/class_weights.json "Label" (the trainer reads the file; engine-level tests pass the list)


def ensure_hip_lib():
    """the gfx950 product library is git-ignored: cross-compile it (hipcc works without a GPU) when a fresh checkout lacks it"""
    if not os.path.exists(L.LIB_PATH) or not os.path.exists(L.LIB_PATH_F16):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "videocad_amd", "csrc"), "all"])
    return L.LIB_PATH


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream_of(device):
    if torch.device(device).type == "cuda":
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return None


def rnd(shape, device, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(device)


def relerr(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ------------------------------------------------------------------------------------------------ GEMM
GEMM_FLAGS = 0        # OR-ed into every check_gemm call (the tile-size fixtures of test_ops_*.py set it)


def check_gemm(lib, device, M, N, K, ct, sa=None, to=None, sb=None, tra=0, trb=0, bias=False, act=0, residual=False, seed=0,
               pad=0, tol=None, splitk=True, flags=0, kernel=None, pack_b=False):
    """C = act(opA @ opB^T + bias) + residual ; operands stored with `pad` extra leading-dimension elements.
    flags: VCAD_GEMM_* kernel-selection flags of the call; kernel: the kernel family (lib.KERNEL_*) that must have run.
    pack_b (bf16x3): B additionally handed over as pre-split hi | lo words (vcad_op_pack_x3) — the result must be bit-identical."""
    x3 = ct == X3
    st = torch.float32 if x3 else ct
    sa = st if sa is None else sa
    to = st if to is None else to
    sb = st if sb is None else sb
    A_log = rnd((M, K), "cpu", seed=seed + 1)
    B_log = rnd((N, K), "cpu", seed=seed + 2)
    A_q = A_log.to(sa).float() if ct == torch.float32 or x3 else A_log.to(ct).float()
    B_q = B_log.to(sb).float() if ct == torch.float32 or x3 else B_log.to(ct).float()

    def store(x_log, tr, dt):
        x = x_log.t().contiguous() if tr else x_log.contiguous()
        buf = torch.zeros(x.shape[0], x.shape[1] + pad, dtype=dt)
        buf[:, : x.shape[1]] = x.to(dt)
        return buf.to(device), x.shape[1] + pad

    A, lda = store(A_log, tra, sa)
    Bm, ldb = store(B_log, trb, sb)
    bias_t = rnd((N,), device, seed=seed + 3) if bias else None
    res_t = rnd((M, N), device, seed=seed + 4) if residual else None
    Cbuf = torch.full((M, N + pad), 7.0, dtype=to, device=device)
    scratch = torch.empty(8 << 20, dtype=torch.float32, device=device) if splitk else None
    tag = C.c_int(0)
    rc = lib.vcad_op_gemm(TD[ct], TD[sa], TD[sb], TD[to], tra, trb, ptr(A), ptr(Bm), ptr(Cbuf), M, N, K, lda, ldb, N + pad,
                          ptr(bias_t), act, ptr(res_t), N, 1.0, ptr(scratch), (scratch.numel() * 4) if splitk else 0, flags | GEMM_FLAGS, C.byref(tag), stream_of(device))
    L.check(lib, rc, "gemm")
    if pack_b:
        assert x3
        Bp = torch.empty(Bm.shape, dtype=torch.int32, device=device)
        L.check(lib, lib.vcad_op_pack_x3(ptr(Bm), ptr(Bp), Bm.numel(), stream_of(device)), "pack_x3")
        C2 = torch.full((M, N + pad), 7.0, dtype=to, device=device)
        rc = lib.vcad_op_gemm(TD[ct], TD[sa], 3, TD[to], tra, trb, ptr(A), ptr(Bp), ptr(C2), M, N, K, lda, ldb, N + pad,
                              ptr(bias_t), act, ptr(res_t), N, 1.0, ptr(scratch), (scratch.numel() * 4) if splitk else 0, flags | GEMM_FLAGS, None, stream_of(device))
        L.check(lib, rc, "gemm (pre-split B)")
        assert torch.equal(C2.cpu(), Cbuf.cpu()), "pre-split B operand: result differs from the in-kernel split"
    if pack_b and (not tra or trb):
        # r04: A pre-split too, and (forward / dgrad layouts) the output written pre-split — same hi / lo values as the in-kernel split, so the fp32
        # result is bit-identical and the pre-split output holds exactly the words vcad_op_pack_x3 makes of it
        pack = lambda t: (lambda o: (L.check(lib, lib.vcad_op_pack_x3(ptr(t), ptr(o), t.numel(), stream_of(device)), "pack_x3"), o)[1])(torch.empty(t.shape, dtype=torch.int32, device=device))
        Ap = pack(A)
        C3 = torch.full((M, N + pad), 7.0, dtype=to, device=device)
        rc = lib.vcad_op_gemm(TD[ct], 3, 3, TD[to], tra, trb, ptr(Ap), ptr(Bp), ptr(C3), M, N, K, lda, ldb, N + pad,
                              ptr(bias_t), act, ptr(res_t), N, 1.0, ptr(scratch), (scratch.numel() * 4) if splitk else 0, flags | GEMM_FLAGS, None, stream_of(device))
        L.check(lib, rc, "gemm (pre-split A and B)")
        assert torch.equal(C3.cpu(), Cbuf.cpu()), "pre-split A operand: result differs from the in-kernel split"
        if not tra:
            C4 = torch.full((M, N + pad), 0x40E00000, dtype=torch.int32, device=device)          # (bit pattern of 7.0f in the pad columns)
            rc = lib.vcad_op_gemm(TD[ct], 3, 3, 3, tra, trb, ptr(Ap), ptr(Bp), ptr(C4), M, N, K, lda, ldb, N + pad,
                                  ptr(bias_t), act, ptr(res_t), N, 1.0, ptr(scratch), (scratch.numel() * 4) if splitk else 0, flags | GEMM_FLAGS, None, stream_of(device))
            L.check(lib, rc, "gemm (pre-split output)")
            assert torch.equal(C4[:, :N].cpu(), pack(Cbuf)[:, :N].cpu()), "pre-split output differs from pack(fp32 output)"
            if pad:
                assert bool((C4[:, N:] == 0x40E00000).all()), "gemm (pre-split output) wrote outside the N columns"
    if kernel is not None:
        assert tag.value == kernel, f"gemm M{M} N{N} K{K} tra={tra} trb={trb} flags={flags}: ran on kernel family {tag.value}, expected {kernel}"
    ref = A_q.double() @ B_q.double().t()
    if bias:
        ref = ref + bias_t.double().cpu()
    if act == 1:
        ref = torch.nn.functional.gelu(ref)
    elif act == 2:
        ref = torch.relu(ref)
    elif act == 3:
        ref = torch.tanh(ref)
    if residual:
        ref = ref + res_t.double().cpu()
    out = Cbuf[:, :N].float().cpu()
    err = relerr(out, ref)
    if tol is None:
        tol = 2e-6 if (ct == torch.float32) else (6e-3 * EPS16[to] if to in S16 else 2e-5)
        if ct in S16 and to == torch.float32:
            tol = 1e-5
        if x3:
            tol = 1.5e-5          # three-term split: ~2^-17 per operand + the dropped lo*lo term, against the UNROUNDED fp32 operands
    assert err < tol, f"gemm M{M} N{N} K{K} ct={ct} sa={sa} to={to} tra={tra} trb={trb}: rel err {err:.3e} > {tol}"
    if pad:
        assert bool((Cbuf[:, N:].float() == 7.0).all()), "gemm wrote outside the N columns"
    return err


# ------------------------------------------------------------------------------------------------ LayerNorm
def check_wgrad_batched(lib, device, tok, shapes, seed=0, flags=0, dt=torch.bfloat16):
    """vcad_op_wgrad_batched: dW_i = dY_i^T X_i for every (N_i, K_i) of `shapes` over the same `tok` rows, one launch — against fp32 matmuls of the rounded operands"""
    import ctypes as C_
    n = len(shapes)
    dYs = [rnd((tok, N), device, seed=seed + 11 * i).to(dt) for i, (N, K) in enumerate(shapes)]
    Xs = [rnd((tok, K), device, seed=seed + 11 * i + 5).to(dt) for i, (N, K) in enumerate(shapes)]
    dWs = [torch.full((N, K), float("nan"), device=device) for (N, K) in shapes]
    scratch = torch.zeros(64 + 64 * sum(N * K for N, K in shapes), dtype=torch.float32, device=device)
    arr = lambda ts: (C_.c_void_p * n)(*[t.data_ptr() for t in ts])
    ints = lambda xs: (C_.c_int * n)(*xs)
    rc = lib.vcad_op_wgrad_batched(n, arr(dYs), arr(Xs), arr(dWs), ints([s[0] for s in shapes]), ints([s[1] for s in shapes]), tok,
                                   ptr(scratch), scratch.numel() * 4, flags, stream_of(device))
    assert rc == 0, lib.vcad_last_error()
    if device != "cpu":
        torch.cuda.synchronize()
    for dY, X, dW in zip(dYs, Xs, dWs):
        ref = dY.float().t() @ X.float()
        err = relerr(dW, ref)
        assert err < 2e-5, err


def check_layernorm(lib, device, rows, C_, dt, seed=0):
    x = rnd((rows, C_), device, seed=seed, scale=2.0) + 0.5
    g = rnd((C_,), device, seed=seed + 1, scale=0.2) + 1.0
    b = rnd((C_,), device, seed=seed + 2, scale=0.1)
    y32 = torch.empty(rows, C_, device=device)
    yt = torch.empty(rows, C_, dtype=dt, device=device)
    stats = torch.empty(rows, 2, device=device)
    rc = lib.vcad_op_layernorm_fwd(L.VCAD_F32, TD[dt], C_, ptr(x), C_, ptr(g), ptr(b), ptr(y32), ptr(yt), ptr(stats), rows, 1e-5, stream_of(device))
    L.check(lib, rc, "ln_fwd")
    xr = x.double().cpu().requires_grad_(True); gr = g.double().cpu().requires_grad_(True); br = b.double().cpu().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (C_,), gr, br, 1e-5)
    assert relerr(y32, ref) < 2e-6, relerr(y32, ref)
    assert relerr(yt, ref) < (2e-6 if dt == torch.float32 else 4e-3 * EPS16[dt])
    dy = rnd((rows, C_), device, seed=seed + 3)
    dyt = dy.to(dt)
    add = rnd((rows, C_), device, seed=seed + 4)
    dx = torch.empty(rows, C_, device=device); dg = torch.empty(C_, device=device); db = torch.empty(C_, device=device)
    scratch = torch.empty(4 << 20, dtype=torch.float32, device=device)
    rc = lib.vcad_op_layernorm_bwd(TD[dt], TD[dt], C_, ptr(dyt), ptr(x), C_, ptr(stats), ptr(g), ptr(add), ptr(dx), None, ptr(dg), ptr(db),
                                   rows, ptr(scratch), scratch.numel() * 4, stream_of(device))
    L.check(lib, rc, "ln_bwd")
    ref.backward(dyt.double().cpu())
    tol = 5e-6 if dt == torch.float32 else 1e-5
    assert relerr(dx, xr.grad + add.double().cpu()) < tol, relerr(dx, xr.grad + add.double().cpu())
    assert relerr(dg, gr.grad) < tol and relerr(db, br.grad) < tol, (relerr(dg, gr.grad), relerr(db, br.grad))


# ------------------------------------------------------------------------------------------------ attention
def attn_ref(q, k, v, window, causal, scale):
    """q,k,v: [B,T,H,D] double. mask rule of csrc/attn.h."""
    Tq, Tk = q.shape[1], k.shape[1]
    i = torch.arange(Tq)[:, None]; j = torch.arange(Tk)[None, :]
    ok = (j >= i - window + 1) & ((j <= i) if causal else torch.ones_like(j, dtype=torch.bool))
    s = torch.einsum("bihd,bjhd->bhij", q, k) * scale
    s = s.masked_fill(~ok, float("-inf"))
    p = s.softmax(-1)
    return torch.einsum("bhij,bjhd->bihd", p, v), torch.logsumexp(s, -1)


def check_attention(lib, device, B, H, T, D, window, causal, dt, seed=0, packed=True, x3=False):
    """x3: fp32 tensors through the bf16x3 form of the kernel (csrc/attn_x3.h: hi / lo split operands, three bf16 MFMAs per product)"""
    scale = 1.0 / math.sqrt(D)
    tcode = 2 if x3 else TD[dt]
    qkv = rnd((B, T, 3, H, D), device, dt, seed=seed)                      # packed projection [B*T, 3*H*D]
    ld = 3 * H * D
    es = qkv.element_size()
    base = qkv.data_ptr()
    o = torch.empty(B, T, H, D, dtype=dt, device=device)
    lse = torch.empty(B, H, T, device=device)
    st = stream_of(device)
    rc = lib.vcad_op_attention_fwd(tcode, D, C.c_void_p(base), C.c_void_p(base + H * D * es), C.c_void_p(base + 2 * H * D * es), ptr(o),
                                   ld, ld, ld, H * D, ptr(lse), B, H, T, T, window, causal, scale, st)
    L.check(lib, rc, "attn_fwd")
    qr, kr, vr = (qkv[:, :, i].double().cpu().requires_grad_(True) for i in range(3))
    ref, lse_ref = attn_ref(qr, kr, vr, window, causal, scale)
    tol = (2e-5 if x3 else 3e-6) if dt == torch.float32 else 6e-3 * EPS16[dt]
    assert relerr(o, ref) < tol, ("attn fwd", relerr(o, ref))
    assert relerr(lse, lse_ref) < 1e-5
    do = rnd((B, T, H, D), device, dt, seed=seed + 1)
    dqkv = torch.zeros(B, T, 3, H, D, dtype=dt, device=device)
    delta = torch.empty(B, H, T, device=device)
    db_ = dqkv.data_ptr()
    rc = lib.vcad_op_attention_bwd_o(tcode, D, C.c_void_p(base), C.c_void_p(base + H * D * es), C.c_void_p(base + 2 * H * D * es), ptr(o), H * D,
                                     ptr(do), ld, ld, ld, H * D, ptr(lse), ptr(delta), C.c_void_p(db_), C.c_void_p(db_ + H * D * es),
                                     C.c_void_p(db_ + 2 * H * D * es), ld, ld, ld, B, H, T, T, window, causal, scale, st)
    L.check(lib, rc, "attn_bwd")
    ref.backward(do.double().cpu())
    tolb = (5e-5 if x3 else 1e-5) if dt == torch.float32 else 1.5e-2 * EPS16[dt]
    for i, g in enumerate((qr.grad, kr.grad, vr.grad)):
        assert relerr(dqkv[:, :, i], g) < tolb, ("attn bwd", i, relerr(dqkv[:, :, i], g))
    if x3:       # r04: the same kernels on pre-split tensors (what the bf16x3 engine hands them): outputs == pack(fp32-tensor outputs), word for word
        pack = lambda t: (lambda o_: (L.check(lib, lib.vcad_op_pack_x3(ptr(t), ptr(o_), t.numel(), st), "pack_x3"), o_)[1])(torch.empty(t.shape, dtype=torch.int32, device=device))
        qp, dop = pack(qkv), pack(do)
        bp = qp.data_ptr()
        o2 = torch.zeros(B, T, H, D, dtype=torch.int32, device=device); lse2 = torch.empty_like(lse)
        L.check(lib, lib.vcad_op_attention_fwd(3, D, C.c_void_p(bp), C.c_void_p(bp + H * D * 4), C.c_void_p(bp + 2 * H * D * 4), ptr(o2),
                                               ld, ld, ld, H * D, ptr(lse2), B, H, T, T, window, causal, scale, st), "attn_fwd (pre-split)")
        assert torch.equal(o2.cpu(), pack(o).cpu()) and torch.equal(lse2.cpu(), lse.cpu()), "pre-split attention forward differs"
        dq2 = torch.zeros(B, T, 3, H, D, dtype=torch.int32, device=device); d2 = dq2.data_ptr()
        L.check(lib, lib.vcad_op_attention_bwd_o(3, D, C.c_void_p(bp), C.c_void_p(bp + H * D * 4), C.c_void_p(bp + 2 * H * D * 4), ptr(o2), H * D,
                                                 ptr(dop), ld, ld, ld, H * D, ptr(lse), ptr(delta), C.c_void_p(d2), C.c_void_p(d2 + H * D * 4),
                                                 C.c_void_p(d2 + 2 * H * D * 4), ld, ld, ld, B, H, T, T, window, causal, scale, st), "attn_bwd (pre-split)")
        assert torch.equal(dq2.cpu(), pack(dqkv).cpu()), "pre-split attention backward differs"


def check_attention_single_query(lib, device, B, H, Tk, dt, seed=0):
    """the cls-only last ViT layer: ONE query (token 0 of every image) against all Tk keys; q / o addressed like the engine does (row
    stride = one whole image)"""
    D = 64
    scale = 1.0 / math.sqrt(D)
    qkv = rnd((B, Tk, 3, H, D), device, dt, seed=seed)
    ld = 3 * H * D; es = qkv.element_size(); base = qkv.data_ptr()
    o = torch.zeros(B, H, D, dtype=dt, device=device)
    lse = torch.empty(B, H, device=device)
    rc = lib.vcad_op_attention_fwd(TD[dt], D, C.c_void_p(base), C.c_void_p(base + H * D * es), C.c_void_p(base + 2 * H * D * es), ptr(o),
                                   Tk * ld, ld, ld, H * D, ptr(lse), B, H, 1, Tk, Tk, 0, scale, stream_of(device))
    L.check(lib, rc, "attn_fwd (single query)")
    q = qkv[:, 0, 0].double().cpu(); k = qkv[:, :, 1].double().cpu(); v = qkv[:, :, 2].double().cpu()      # [B,H,D], [B,Tk,H,D]
    sc = torch.einsum("bhd,bkhd->bhk", q, k) * scale
    ref = torch.einsum("bhk,bkhd->bhd", torch.softmax(sc, -1), v)
    tol = 3e-6 if dt == torch.float32 else 6e-3
    assert relerr(o, ref) < tol, ("attn fwd single query", relerr(o, ref))
    assert relerr(lse, torch.logsumexp(sc, -1)) < 1e-5


def check_cls_attention(lib, device, N, H, P1, dt, seed=0, pad=0):
    """csrc/attn_cls.h: the last ViT layer's class-token attention on (g = q W_k, normalised tokens): c[n,h] = softmax_j(scale g[n,h] . ha[n,j]) ha[n]
    and its backward, against fp64 autograd on the SAME 16-bit-rounded inputs (ha rows stored with `pad` extra leading-dimension elements)."""
    D = 512
    scale = 0.125
    ld = D + pad
    ha_st = torch.zeros(N * P1, ld, dtype=dt, device=device)
    ha_log = rnd((N * P1, D), "cpu", seed=seed + 1).to(dt)
    ha_st[:, :D] = ha_log.to(device)
    g = rnd((N, H, D), device, dt, seed=seed + 2, scale=0.25)
    dc = rnd((N, H, D), device, dt, seed=seed + 3)
    c = torch.zeros(N, H, D, dtype=dt, device=device); lse = torch.zeros(N, H, device=device)
    st = stream_of(device)
    L.check(lib, lib.vcad_op_cls_attention_fwd(ptr(ha_st), ld, ptr(g), ptr(c), ptr(lse), N, H, P1, scale, st), "cls_attention_fwd")
    ha64 = ha_log.double().view(N, P1, D).requires_grad_(True); g64 = g.double().cpu().requires_grad_(True)
    sc = torch.einsum("nhd,njd->nhj", g64, ha64) * scale
    ref = torch.einsum("nhj,njd->nhd", torch.softmax(sc, -1), ha64)
    tol = 6e-3 * EPS16[dt] + 2e-5
    ref_d = ref.detach()
    assert relerr(c, ref_d) < tol, ("cls attention fwd", relerr(c, ref_d))
    assert relerr(lse, torch.logsumexp(sc, -1).detach()) < 1e-5
    ref.backward(dc.double().cpu())
    dg = torch.zeros(N, H, D, dtype=dt, device=device); dha = torch.zeros(N * P1, ld, dtype=dt, device=device); r0 = torch.zeros(N, D, device=device)
    L.check(lib, lib.vcad_op_cls_attention_bwd(ptr(ha_st), ld, ptr(g), ptr(dc), ptr(lse), ptr(dg), ptr(dha), ld, ptr(r0), N, H, P1, scale, st), "cls_attention_bwd")
    assert relerr(dg, g64.grad) < tol, ("cls attention dg", relerr(dg, g64.grad))
    dha_log = dha[:, :D].float().cpu().view(N, P1, D)
    assert relerr(dha_log, ha64.grad) < tol, ("cls attention dha", relerr(dha_log, ha64.grad))
    assert relerr(r0, ha64.grad[:, 0]) < tol, ("cls attention r0", relerr(r0, ha64.grad[:, 0]))      # (fp32 copy of the class row: the matrix cores' operands — dS, P~ — are rounded, the sum is not)
    if pad:
        assert float(dha[:, D:].float().abs().max()) == 0.0, "cls attention backward wrote into the row padding"


# ------------------------------------------------------------------------------------------------ MXFP8 (csrc/gemm_mx8.h)
def mx8_quant_ref(x):
    """[rows, cols] float -> (e4m3 bytes [rows, cols] uint8, E8M0 scale bytes [rows, cols/32] uint8): the OCP MX rule the kernel
    implements (shared exponent = floor(log2(block amax)) - 8, elements RNE to e4m3, saturating)."""
    r, c = x.shape
    xb = x.float().reshape(r, c // 32, 32)
    am = xb.abs().amax(-1)
    e = torch.where(am > 0, torch.floor(torch.log2(am.double())).float() - 8, torch.full_like(am, -127.0)).clamp(-127, 127)
    q = (xb * torch.exp2(-e).unsqueeze(-1)).clamp(-448, 448).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).reshape(r, c), (e + 127).to(torch.uint8)


def mx8_dequant(q, s):
    r, c = q.shape
    return (q.view(torch.float8_e4m3fn).float().reshape(r, c // 32, 32) * torch.exp2(s.float() - 127).unsqueeze(-1)).reshape(r, c).double()


def check_mx8(lib, device, M, N, K, to, bias=False, act=0, residual=False, seed=0):
    A = rnd((M, K), "cpu", torch.bfloat16, seed=seed + 1); W = rnd((N, K), "cpu", torch.bfloat16, seed=seed + 2, scale=0.05)
    A[0, :32] = 0                                           # an all-zero block
    A[1, 5] = 3000.0                                        # a block dominated by one outlier
    out = {}
    for name, x, dt in (("a", A, torch.bfloat16), ("w", W.float(), torch.float32)):       # both source types of the quantiser
        xd = x.to(device)
        q = torch.empty(x.shape, dtype=torch.uint8, device=device); sc = torch.empty(x.shape[0], K // 32, dtype=torch.uint8, device=device)
        L.check(lib, lib.vcad_op_quant_mx8(TD[dt], ptr(xd), K, ptr(q), ptr(sc), x.shape[0], K, stream_of(device)), "quant_mx8")
        qr, sr = mx8_quant_ref(x)
        assert torch.equal(sc.cpu(), sr), (name, "scales differ")
        assert torch.equal(q.cpu(), qr), (name, "e4m3 bytes differ", int((q.cpu() != qr).sum()))
        out[name] = (q, sc)
    bias_t = rnd((N,), device, seed=seed + 3) if bias else None
    res_t = rnd((M, N), device, seed=seed + 4) if residual else None
    Cbuf = torch.full((M, N), 7.0, dtype=to, device=device)
    L.check(lib, lib.vcad_op_gemm_mx8(TD[to], ptr(out["a"][0]), ptr(out["a"][1]), ptr(out["w"][0]), ptr(out["w"][1]), ptr(Cbuf), M, N, K, N,
                                      ptr(bias_t), act, ptr(res_t), N, stream_of(device)), "gemm_mx8")
    ref = mx8_dequant(out["a"][0].cpu(), out["a"][1].cpu()) @ mx8_dequant(out["w"][0].cpu(), out["w"][1].cpu()).t()
    exact = A.double() @ W.double().t()
    if bias:
        ref = ref + bias_t.double().cpu(); exact = exact + bias_t.double().cpu()
    if act == 1:
        ref = torch.nn.functional.gelu(ref); exact = torch.nn.functional.gelu(exact)
    if residual:
        ref = ref + res_t.double().cpu(); exact = exact + res_t.double().cpu()
    err = relerr(Cbuf.float(), ref)
    # fp32 out: the matrix core's own rounding of a 64-term block dot product (measured 1.5e-5 .. 6e-5 on gfx950; the emulator sums in fp32)
    tol = 2e-4 if to == torch.float32 else 6e-3
    assert err < tol, f"gemm_mx8 M{M} N{N} K{K} to={to}: rel err vs dequantised reference {err:.3e} > {tol}"
    qerr = relerr(Cbuf.float(), exact)                       # what the 8-bit operands cost against the bf16 operands
    assert qerr < 6e-2, qerr
    return err, qerr


def f16_overflow_drill(tr, batch, inject_at=3, factor=1e8, window=2, grow_after=8, max_steps=40):
    """r05: the fp16 gradient-scale machinery of the TRAINER on a live run — a batch whose loss is `factor` times larger is injected
    at step `inject_at` (the step is run by hand through the public engine API with external dlogits = factor x the loss's own: exactly a factor-times
    larger loss); everything else is `trainer.train_step`.  Expected, and asserted: that update is SKIPPED (weights, moments, fp16 shadow untouched; a
    non-finite norm comes back) -> one window later the trainer HALVES the scale and takes the Adam step counter back by one -> training RECOVERS
    (finite norms, weights move again at the lower scale) -> after `grow_after` clean steps the scale CLIMBS BACK to the automatic rule's value and the
    engine is in automatic mode again.  Returns the (step, scale, finite) history."""
    import torch
    from videocad_amd import lib as L
    eng = tr.engine
    assert eng.cfg.dtype == L.VCAD_F16
    tr.OVERFLOW_WINDOW = window; eng.GROW_AFTER = grow_after
    bd = tr.prepare_batch(batch)
    B, T = bd["actions"].shape[0], bd["actions"].shape[1] - 1
    hist, auto_scale, halved_at, regrown_at = [], None, None, None
    for i in range(max_steps):
        before = eng.params.clone()
        if i == inject_at:
            inputs = tr._prepare_model_inputs(bd, False)
            tr.native._arm_dropout()
            cmds, pars = eng.forward(inputs["frames"], inputs["actions"], inputs["cad_image"])
            eng.loss(cmds, pars, bd["actions"][:, 1:], tr._label_w(), use_mse=tr.use_mse, class_weights=tr._class_w())
            dc, dp = eng.dl_views(B, T)
            tr.gradsync.backward(dc * factor, dp * factor)
            m0, v0, s0 = eng.m.clone(), eng.v.clone(), eng.shadow.clone()
            norm = eng.optimizer_step(lr=tr.optimizer.lr, betas=tr.optimizer.betas, eps=tr.optimizer.eps, max_norm=1.0, grad_scale=1.0 / tr.gradsync.world)
            tr.native.mark_shadow_fresh()
            tr._watch_overflow(norm)
            assert not bool(torch.isfinite(norm[0])), "the injected loss did not overflow the scaled backward"
            assert torch.equal(eng.params, before) and torch.equal(eng.m, m0) and torch.equal(eng.v, v0) and torch.equal(eng.shadow, s0), "the overflowed update was not skipped"
        else:
            loss, _ = tr.train_step(bd)
            assert bool(torch.isfinite(loss)) and bool(torch.isfinite(tr._last_norm[0])), i
            assert not torch.equal(eng.params, before), f"step {i}: the weights did not move"
        if auto_scale is None:
            auto_scale = eng.grad_scale
        hist.append((i, eng.grad_scale, i != inject_at))
        if halved_at is None and eng.grad_scale < auto_scale:
            halved_at = i
            assert eng.grad_scale == auto_scale / 2 and eng.skipped_steps == 1 and eng._scale_target == auto_scale
            assert eng.step_count == i + 1 - 1, (eng.step_count, i)            # the skipped update does not count towards Adam's bias correction
        if halved_at is not None and regrown_at is None and eng.grad_scale == auto_scale and getattr(eng, "_scale_target", None) is None:
            regrown_at = i
            break
    assert halved_at is not None and inject_at < halved_at <= inject_at + 2 * window, (halved_at, hist)
    assert regrown_at is not None and regrown_at >= halved_at + grow_after - window, (regrown_at, hist)
    # one more step in automatic mode at the restored scale
    loss, _ = tr.train_step(bd)
    assert bool(torch.isfinite(loss)) and eng.grad_scale == auto_scale and eng.skipped_steps == 1
    return hist
