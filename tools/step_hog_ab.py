"""r06: the closest thing to RCCL contention a 1-GPU lease can produce.  During data-parallel training RCCL's kernels hold a few CUs on
the communication stream while the backward runs; the persistent GEMM launches one workgroup per CU, so with static per-workgroup item lists the workgroups
that find their CU taken run as a second round.  GradSync therefore turns on ticket-drawn items (VCAD_GEMM_DYNAMIC) for world > 1.  This script times the
WHOLE train step (bench.py's step: forward, loss, backward, clip, Adam; dropout 0.1) while a background kernel pins N CUs (one 1024-thread / 160 KiB-LDS
workgroup each: nothing else fits beside it) on another stream for the whole step — static lists against tickets, N = 0 / 8 / 16 / 32.

    make -C videocad_amd/csrc ab && python tools/step_hog_ab.py [--batch 32 --seq 64] [--hogs 0 8 16 32]"""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocad_amd import lib as L

lib = L.load_ab()
from videocad_amd import bench_impl as BI      # noqa: E402  (after load_ab: the package runs on the A/B build in this process)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32); ap.add_argument("--seq", type=int, default=64)
    ap.add_argument("--reserve", type=int, default=2, help="third column: persistent GEMM on 256 - 8 n CUs")
    ap.add_argument("--hogs", type=int, nargs="*", default=[0, 8, 16, 32]); ap.add_argument("--steps", type=int, default=12)
    a = ap.parse_args()
    global RES
    RES = a.reserve
    dev = "cuda:0"
    model, tr = BI.build_trainer("bf16", 0.1, dev, 0)
    bd = BI.synthetic_batch(a.batch, a.seq, 1, dev)
    eng = model._engine
    hog_stream = torch.cuda.Stream()
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    for _ in range(4):
        tr.train_step(bd)
    torch.cuda.synchronize()

    def run(n_hog, dynamic, extra=0):
        eng.set_gemm_flags((L.GEMM_DYNAMIC if dynamic else 0) | extra)
        for _ in range(2):
            tr.train_step(bd)
        torch.cuda.synchronize()
        ms = []
        for _ in range(a.steps):
            if n_hog:       # the hog outlives the step (60 ms); the step starts once it is resident
                assert lib.vcad_debug_hog(n_hog, 60000, C.c_void_p(sink.data_ptr()), C.c_void_p(hog_stream.cuda_stream)) == 0
                import time; time.sleep(0.002)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); tr.train_step(bd); e1.record(); e1.synchronize()
            ms.append(e0.elapsed_time(e1))
            torch.cuda.synchronize()                 # (the hog drains before the next step's hog is launched)
        return statistics.median(ms)

    print(f"# whole train step, B = {a.batch}, T = {a.seq}, bf16, dropout 0.1; median of {a.steps} steps; hog = CUs pinned by a background kernel for the whole step")
    base = None
    for n in a.hogs:
        for rnd in range(2):
            s_ms, d_ms, m_ms = run(n, False), run(n, True), run(n, True, L.gemm_reserve_cus(RES))
            if base is None:
                base = s_ms
            print(f"hog {n:3d} CUs  static {s_ms:7.3f} ms ({(s_ms / base - 1) * 100:+5.1f} %)   tickets {d_ms:7.3f} ms ({(d_ms / base - 1) * 100:+5.1f} %)   "
                  f"tickets on {256 - 8 * RES} CUs {m_ms:7.3f} ms ({(m_ms / base - 1) * 100:+5.1f} %)   (capacity alone: {(256 / (256 - n) - 1) * 100:+5.1f} %)", flush=True)
    if max(a.hogs) > 0:
        breakdown(tr, bd, eng, 8, hog_stream, sink)


def breakdown(tr, bd, eng, n_hog, hog_stream, sink):
    """per-category kernel time of ONE step under the library's HIP-event profiler (single stream), with and without the hog: which kernels pay"""
    import time
    out = {}
    for n in (0, n_hog):
        eng.set_gemm_flags(L.GEMM_DYNAMIC)
        tr.train_step(bd); torch.cuda.synchronize()
        if n:
            assert lib.vcad_debug_hog(n, 90000, C.c_void_p(sink.data_ptr()), C.c_void_p(hog_stream.cuda_stream)) == 0
            time.sleep(0.002)
        roof_unused = BI.profile_step(tr, bd, "bf16", bd["actions"].shape[0], bd["actions"].shape[1] - 1, 1.0)
        torch.cuda.synchronize()
        out[n] = roof_unused[0]
    print(f"# HIP-event profile of one step (single stream, tickets on): ms per category without / with {n_hog} CUs pinned")
    for k in out[0]:
        a, b = out[0][k]["ms"], out[n_hog][k]["ms"]
        print(f"  {k:12s} {a:8.3f} -> {b:8.3f} ms  ({(b / a - 1) * 100 if a > 0 else 0:+6.1f} %)   launches {out[0][k]['launches']}")


if __name__ == "__main__":
    main()
