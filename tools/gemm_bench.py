"""Per-shape GEMM timing through the C ABI (the shapes one C2 train step launches).  Usage: python tools/gemm_bench.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocad_amd import lib as L

if os.environ.get("VCAD_ABL"):      # ablated build (tools/gemm_ablate.sh): timing only, results are garbage
    os.environ["VCAD_AB_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", f"libvcad_abl{os.environ['VCAD_ABL']}.so")
lib = L.load_ab()
dev = "cuda:0"
BF, F32 = torch.bfloat16, torch.float32
TD = {F32: 0, BF: 1}
scratch = torch.empty(64 << 20, dtype=torch.uint8, device=dev)


def run(name, M, N, K, sa=BF, sb=BF, to=BF, tra=0, trb=0, bias=False, res=False, iters=20):
    A = torch.randn((K, M) if tra else (M, K), device=dev).to(sa)
    B = torch.randn((K, N) if trb else (N, K), device=dev).to(sb)
    Cm = torch.empty(M, N, dtype=to, device=dev)
    bias_t = torch.randn(N, device=dev) if bias else None
    res_t = torch.randn(M, N, device=dev) if res else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call():
        rc = lib.vcad_op_gemm(1, TD[sa], TD[sb], TD[to], tra, trb, p(A), p(B), p(Cm), M, N, K, A.shape[1], B.shape[1], N, p(bias_t), 0,
                              p(res_t), N, 1.0, p(scratch), scratch.numel(), 0, None, st)
        assert rc == 0, lib.vcad_last_error()
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    print(f"{name:34s} M={M:6d} N={N:5d} K={K:6d} tra={tra} trb={trb} {str(sa)[6:]:>8s}->{str(to)[6:]:<8s} {ms*1e3:8.1f} us  {tf:7.1f} TF/s", flush=True)


def both(name, *a, **k):
    """old register-staged kernel vs automatic choice (persistent DMA-fed kernel where legal)"""
    lib.vcad_debug_gemm_dma(0); run(name + " [reg]", *a, **k)
    lib.vcad_debug_gemm_dma(-1); run(name + " [auto]", *a, **k)


R = 104000
if len(sys.argv) > 1 and sys.argv[1] == "heads":
    for tile in (128, 64):
        lib.vcad_debug_force_gemm_tile(tile)
        run(f"heads fwd (f32 A, f32 out) tile{tile}", 2080, 6000, 1024, sa=F32, to=F32, bias=True)
        run(f"heads wgrad (f32 A dlogits, f32 B) tile{tile}", 6000, 1024, 2080, sa=F32, sb=F32, to=F32, tra=1, trb=1)
        run(f"heads wgrad (bf16 B) tile{tile}", 6000, 1024, 2080, sa=F32, sb=BF, to=F32, tra=1, trb=1)
        run(f"heads dgrad (f32 A) tile{tile}", 2080, 1024, 6000, sa=F32, to=BF, trb=1)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "nn":
    lib.vcad_debug_gemm_dma(1)
    run("dqkv dgrad as NN (W^T shadow) [dma]", R, 512, 3072)
    run("dqkv dgrad NT [dma]", R, 512, 3072, trb=1)
    run("dao dgrad as NN [dma]", R, 1024, 512)
    run("dao dgrad NT [dma]", R, 1024, 512, trb=1)
    lib.vcad_debug_gemm_dma(0)
    run("dqkv dgrad as NN [reg]", R, 512, 3072)
    run("dqkv dgrad NT [reg]", R, 512, 3072, trb=1)
    sys.exit(0)
def three(name, *a, **k):
    """forced through the persistent kernels: r01 lockstep kernel vs the ping-pong kernel (same launch, same data)"""
    lib.vcad_debug_gemm_dma(1)
    lib.vcad_debug_gemm_variant(0); run(name + " [lockstep]", *a, **k)
    lib.vcad_debug_gemm_variant(1); run(name + " [ping-pong]", *a, **k)
    lib.vcad_debug_gemm_dma(-1)


def wide(name, *a, **k):
    """persistent kernel, 256 x 128 tile vs 256 x 256 tile (same launch, same data)"""
    lib.vcad_debug_gemm_dma(1)
    lib.vcad_debug_gemm_wide(0); run(name + " [256x128]", *a, **k)
    lib.vcad_debug_gemm_wide(1); run(name + " [256x256]", *a, **k)
    lib.vcad_debug_gemm_dma(-1); lib.vcad_debug_gemm_wide(-1)


if len(sys.argv) > 1 and sys.argv[1] == "mx8":
    def run8(name, M, N, K, to=BF, bias=False, res=False, iters=20):
        A = torch.randn(M, K, device=dev).to(BF); W = (torch.randn(N, K, device=dev) * 0.05).to(BF)
        qa = torch.empty(M, K, dtype=torch.uint8, device=dev); sa = torch.empty(M, K // 32, dtype=torch.uint8, device=dev)
        qw = torch.empty(N, K, dtype=torch.uint8, device=dev); sw = torch.empty(N, K // 32, dtype=torch.uint8, device=dev)
        Cm = torch.empty(M, N, dtype=to, device=dev)
        bias_t = torch.randn(N, device=dev) if bias else None; res_t = torch.randn(M, N, device=dev) if res else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.vcad_op_quant_mx8(1, p(W), K, p(qw), p(sw), N, K, st) == 0
        quant = lambda: lib.vcad_op_quant_mx8(1, p(A), K, p(qa), p(sa), M, K, st)
        gemm = lambda: lib.vcad_op_gemm_mx8(TD[to], p(qa), p(sa), p(qw), p(sw), p(Cm), M, N, K, N, p(bias_t), 0, p(res_t), N, st)
        for fn, what in ((quant, "quantise A"), (gemm, "gemm")):
            for _ in range(3):
                assert fn() == 0, lib.vcad_last_error()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            print(f"{name:28s} {what:10s} M={M:6d} N={N:5d} K={K:5d} {ms*1e3:8.1f} us" + (f"  {2.0*M*N*K/(ms*1e-3)/1e12:7.1f} TF/s" if what == "gemm" else f"  {M*K*3.03/(ms*1e-3)/1e9:7.1f} GB/s"), flush=True)
    for rep in range(2):
        run8("vit qkv fwd [mxfp8]", R, 3072, 512)
        run8("vit out fwd +res f32 [mxfp8]", R, 512, 1024, to=F32, bias=True, res=True)
        run8("vit mlp fwd [mxfp8]", R, 512, 512, bias=True)
        run8("square 8192 [mxfp8]", 8192, 8192, 8192)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "waves":
    # 256-wide tile of the persistent kernel: eight waves (64 x 128 each) vs four waves (128 x 128 each), interleaved; with and without epilogue
    lib.vcad_debug_gemm_dma(1)
    shapes = [("qkv fwd", (R, 3072, 512), {}), ("dqkv dgrad W^T", (R, 512, 3072), {}), ("dao dgrad W^T", (R, 1024, 512), {}), ("dh dgrad W^T", (R, 512, 512), {}),
              ("patch embed f32", (101920, 512, 1024), dict(to=F32, bias=True)), ("qkv wgrad", (3072, 512, R), dict(to=F32, tra=1, trb=1)),
              ("out wgrad", (512, 1024, R), dict(to=F32, tra=1, trb=1)), ("square 8192", (8192, 8192, 8192), {})]
    for name, dims, kw in shapes:
        for sk in (0, 64):
            lib.vcad_debug_gemm_skip(sk)
            for rnd in range(2):
                for w in (8, 4):
                    lib.vcad_debug_gemm_waves(w)
                    run(f"{name} [{'no epi' if sk else 'full'} {w} waves #{rnd}]", *dims, **kw)
    lib.vcad_debug_gemm_skip(0); lib.vcad_debug_gemm_waves(8)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "mid":
    # decoder-size problems: register-staged kernel vs the six-stage DMA-ring kernel (gemm_mid.h), interleaved
    shapes = [("dec out fwd +res f32", (2048, 1024, 1024), dict(to=F32, bias=True, res=True)), ("dec q-proj fwd bf16", (2048, 1024, 1024), dict(bias=True)),
              ("dec kv-proj fwd", (2048, 2048, 1024), dict(bias=True)), ("dec in_proj fwd", (2048, 3072, 1024), dict(bias=True)),
              ("dec dgrad W f32+res", (2048, 1024, 1024), dict(to=F32, trb=1, res=True)), ("dec dgrad W bf16", (2048, 1024, 1024), dict(trb=1)),
              ("dec dkv dgrad", (2048, 1024, 2048), dict(to=F32, trb=1, res=True)), ("dec dqkv dgrad", (2048, 1024, 3072), dict(to=F32, trb=1, res=True)),
              ("C4 out fwd +res", (2976, 1024, 1024), dict(to=F32, bias=True, res=True)), ("cad vit qkv", (1600, 3072, 512), {}), ("cad vit out +res", (1600, 512, 1024), dict(to=F32, bias=True, res=True))]
    for name, dims, kw in shapes:
        for rnd in range(2):
            for m in (0, 1):
                lib.vcad_debug_gemm_mid(m)
                run(f"{name} [{'ring' if m else 'reg'} #{rnd}]", *dims, iters=50, **kw)
    lib.vcad_debug_gemm_mid(-1)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "mainloop":
    # main loop only (epilogue dropped): run under VCAD_ABL = 1 (no MFMA) / 2 (no fragment reads) / 4 (no DMA) / combinations
    lib.vcad_debug_gemm_dma(1); lib.vcad_debug_gemm_skip(64)
    tag = os.environ.get("VCAD_ABL", "0")
    for rnd in range(2):
        run(f"qkv fwd [abl {tag}]", R, 3072, 512)
        run(f"dqkv dgrad W^T [abl {tag}]", R, 512, 3072)
        run(f"qkv wgrad [abl {tag}]", 3072, 512, R, to=F32, tra=1, trb=1)
        run(f"square 8192 [abl {tag}]", 8192, 8192, 8192)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "abl":
    # where does an item's time go?  full launch vs the same launch without its epilogue (debug bit 64: accumulators are dropped)
    lib.vcad_debug_gemm_dma(1)
    shapes = [("qkv fwd", (R, 3072, 512), {}), ("dqkv dgrad W^T", (R, 512, 3072), {}), ("dao dgrad W^T", (R, 1024, 512), {}), ("dh dgrad W^T", (R, 512, 512), {}),
              ("patch embed f32", (101920, 512, 1024), dict(to=F32, bias=True)), ("out fwd +res f32", (R, 512, 1024), dict(to=F32, bias=True, res=True)),
              ("mlp2 fwd +res f32", (R, 512, 512), dict(to=F32, bias=True, res=True)), ("qkv wgrad", (3072, 512, R), dict(to=F32, tra=1, trb=1)),
              ("out wgrad", (512, 1024, R), dict(to=F32, tra=1, trb=1)), ("mlp wgrad", (512, 512, R), dict(to=F32, tra=1, trb=1)), ("square 8192", (8192, 8192, 8192), {})]
    for name, dims, kw in shapes:
        for rnd in range(2):
            for sk in (0, 64):
                lib.vcad_debug_gemm_skip(sk)
                run(f"{name} [{'no epilogue' if sk else 'full'} #{rnd}]", *dims, **kw)
    lib.vcad_debug_gemm_skip(0)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "spread":
    # the residual epilogue of the 128-wide tile: full / no epilogue (64) / no epilogue but its loads and stores spread over the main loop (256)
    lib.vcad_debug_gemm_dma(1)
    for name, dims, kw in [("out fwd +res f32", (R, 512, 1024), dict(to=F32, bias=True, res=True)), ("mlp2 fwd +res f32", (R, 512, 512), dict(to=F32, bias=True, res=True)),
                           ("out fwd +res f32 102400", (102400, 512, 1024), dict(to=F32, bias=True, res=True)), ("mlp2 fwd +res f32 102400", (102400, 512, 512), dict(to=F32, bias=True, res=True))]:
        for rnd in range(3):
            for sk, tag in ((0, "full"), (64, "no epilogue"), (256, "spread traffic, no epilogue")):
                lib.vcad_debug_gemm_skip(sk)
                run(f"{name} [{tag} #{rnd}]", *dims, **kw)
    lib.vcad_debug_gemm_skip(0)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "wr":
    # is the epilogue's cost the HBM write stream?  full / no epilogue (64) / the same stores aimed at 256 rows of C that stay in L2 (128)
    lib.vcad_debug_gemm_dma(1)
    for name, dims, kw in [("qkv fwd", (R, 3072, 512), {}), ("dao dgrad W^T", (R, 1024, 512), {}), ("dh dgrad W^T", (R, 512, 512), {}), ("dqkv dgrad W^T", (R, 512, 3072), {})]:
        for rnd in range(2):
            for sk, tag in ((0, "full"), (64, "no epilogue"), (128, "stores stay in L2")):
                lib.vcad_debug_gemm_skip(sk)
                run(f"{name} [{tag} #{rnd}]", *dims, **kw)
    lib.vcad_debug_gemm_skip(0)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "epi2":
    # interleaved A/B (rule: perf deltas come from within-process interleaved rounds): per shape, 3 rounds of row / col
    lib.vcad_debug_gemm_dma(1)
    shapes = [("qkv fwd wide", 1, (R, 3072, 512), {}), ("dao dgrad W^T wide", 1, (R, 1024, 512), {}), ("dh dgrad W^T wide", 1, (R, 512, 512), {}),
              ("dh dgrad W^T narrow", 0, (R, 512, 512), {}), ("dqkv dgrad W^T wide", 1, (R, 512, 3072), {}),
              ("patch embed f32 wide", 1, (101920, 512, 1024), dict(to=F32, bias=True)),
              ("out fwd +res f32 narrow", 0, (R, 512, 1024), dict(to=F32, bias=True, res=True)),
              ("mlp2 fwd +res f32 narrow", 0, (R, 512, 512), dict(to=F32, bias=True, res=True))]
    for name, w, dims, kw in shapes:
        lib.vcad_debug_gemm_wide(w)
        for rnd in range(3):
            for e in (0, 1):
                lib.vcad_debug_gemm_epilogue(e)
                run(f"{name} [{'col' if e else 'row'} #{rnd}]", *dims, **kw)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "epi":
    lib.vcad_debug_gemm_dma(1)
    for rep in range(2):
        for w in (0, 1):
            lib.vcad_debug_gemm_wide(w)
            for e in (0, 1):
                lib.vcad_debug_gemm_epilogue(e)
                tag = f"[wide={w} {'col' if e else 'row'}]"
                run("vit qkv fwd " + tag, R, 3072, 512)
                run("vit dqkv dgrad W^T (bf16 out) " + tag, R, 512, 3072)
                run("vit dao dgrad W^T (bf16 out) " + tag, R, 1024, 512)
                run("vit dh dgrad W^T (bf16 out) " + tag, R, 512, 512)
                run("patch embed fwd (f32 out) " + tag, 101920, 512, 1024, to=F32, bias=True)
                if not w:
                    run("vit out fwd (+res, f32 out) " + tag, R, 512, 1024, to=F32, bias=True, res=True)
                    run("vit mlp2 fwd (+res, f32 out) " + tag, R, 512, 512, to=F32, bias=True, res=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "xcd":
    lib.vcad_debug_gemm_dma(1)
    for rep in range(2):
        for w in (0, 1):
            lib.vcad_debug_gemm_wide(w)
            for xn in (0, 2, 4, 8):
                lib.vcad_debug_gemm_xcd_cols(xn)
                run(f"vit qkv fwd [wide={w} xcd-cols={xn}]", R, 3072, 512)
            for xn in (0, 2, 4):
                lib.vcad_debug_gemm_xcd_cols(xn)
                run(f"vit dao dgrad W^T N=1024 [wide={w} xcd-cols={xn}]", R, 1024, 512)
            for xn in (0, 2):
                lib.vcad_debug_gemm_xcd_cols(xn)
                run(f"vit dqkv dgrad W^T K=3072 [wide={w} xcd-cols={xn}]", R, 512, 3072)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "wide":
    for rep in range(2):
        wide("vit qkv fwd", R, 3072, 512)
        wide("patch embed fwd (f32 out)", 101920, 512, 1024, to=F32, bias=True)
        wide("vit dqkv dgrad via W^T (NN)", R, 512, 3072)
        wide("vit dao dgrad via W^T (NN)", R, 1024, 512)
        wide("vit dz/dh dgrad via W^T (NN)", R, 512, 512)
        wide("vit qkv wgrad", 3072, 512, R, to=F32, tra=1, trb=1)
        wide("vit out wgrad", 512, 1024, R, to=F32, tra=1, trb=1)
        wide("vit mlp wgrad", 512, 512, R, to=F32, tra=1, trb=1)
        wide("square 4096", 4096, 4096, 4096)
        wide("square 8192", 8192, 8192, 8192)
        wide("square 8192 TT", 8192, 8192, 8192, to=F32, tra=1, trb=1)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "pp":
    for rep in range(2):
        three("vit qkv fwd", R, 3072, 512)
        three("vit qkv fwd + bias", R, 3072, 512, bias=True)
        three("patch embed fwd (f32 out)", 101920, 512, 1024, to=F32, bias=True)
        three("vit dqkv dgrad via W^T (NN)", R, 512, 3072)
        three("vit dao dgrad via W^T (NN)", R, 1024, 512)
        three("vit dz/dh dgrad via W^T (NN)", R, 512, 512)
        three("vit dqkv dgrad NT", R, 512, 3072, trb=1)
        three("vit qkv wgrad", 3072, 512, R, to=F32, tra=1, trb=1)
        three("vit out wgrad", 512, 1024, R, to=F32, tra=1, trb=1)
        three("vit mlp wgrad", 512, 512, R, to=F32, tra=1, trb=1)
        three("square 4096", 4096, 4096, 4096)
        three("square 8192", 8192, 8192, 8192)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "dma":
    both("vit qkv fwd", R, 3072, 512, bias=True)
    both("vit out fwd (+res, f32 out)", R, 512, 1024, to=F32, bias=True, res=True)
    both("vit mlp1 fwd", R, 512, 512, bias=True)
    both("vit mlp2 fwd (+res, f32 out)", R, 512, 512, to=F32, bias=True, res=True)
    both("vit dqkv dgrad", R, 512, 3072, trb=1)
    both("vit dao dgrad", R, 1024, 512, trb=1)
    both("vit dz dgrad", R, 512, 512, trb=1)
    both("vit qkv wgrad", 3072, 512, R, to=F32, tra=1, trb=1)
    both("vit out wgrad", 512, 1024, R, to=F32, tra=1, trb=1)
    both("vit mlp wgrad", 512, 512, R, to=F32, tra=1, trb=1)
    both("square 4096", 4096, 4096, 4096)
    both("square 8192", 8192, 8192, 8192)
    both("square 8192 NT", 8192, 8192, 8192, trb=1)
    both("square 8192 TT", 8192, 8192, 8192, to=F32, tra=1, trb=1)
    sys.exit(0)
run("vit qkv fwd", R, 3072, 512)
run("vit out fwd (+res, f32 out)", R, 512, 1024, to=F32, bias=True, res=True)
run("vit mlp1 fwd", R, 512, 512, bias=True)
run("vit mlp2 fwd (+res, f32 out)", R, 512, 512, to=F32, bias=True, res=True)
run("patch embed fwd (f32 out)", 101920, 512, 1024, to=F32, bias=True)
run("vit dqkv dgrad", R, 512, 3072, trb=1)
run("vit dao dgrad (f32 A)", R, 1024, 512, sa=F32, trb=1)
run("vit dz dgrad (f32 A)", R, 512, 512, sa=F32, trb=1)
run("vit qkv wgrad", 3072, 512, R, to=F32, tra=1, trb=1)
run("vit out wgrad (f32 A)", 512, 1024, R, sa=F32, to=F32, tra=1, trb=1)
run("vit mlp wgrad", 512, 512, R, to=F32, tra=1, trb=1)
run("dec in_proj fwd (f32 A)", 2048, 3072, 1024, sa=F32, bias=True)
run("dec 1024x1024 fwd (+res)", 2048, 1024, 1024, to=F32, bias=True, res=True)
run("heads fwd (f32 A, f32 out)", 2048, 6000, 1024, sa=F32, to=F32, bias=True)
run("heads wgrad (f32 A, f32 B)", 6000, 1024, 2048, sa=F32, sb=F32, to=F32, tra=1, trb=1)
run("square 4096 (reference point)", 4096, 4096, 4096)
run("square 8192 (reference point)", 8192, 8192, 8192)
