"""Interleaved in-process A/B of engine GEMM flags on the whole train step (bench.py's step: forward, loss, backward, clip, Adam; dropout 0.1).

    python tools/flag_ab.py [--batch 32 --seq 64] [--dtype bf16] [--rounds 4] [--steps 12] --flags 0 mini_never ...
    python tools/flag_ab.py --knob vcad_debug_attn_prefetch --values 0 64 128        (A/B build)

Each flag set is timed `rounds` times in turn (A B A B ...), median of `steps` steps per leg; one process, one box."""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocad_amd import lib as L               # noqa: E402
if "--knob" in sys.argv:                        # A/B-build knobs (vcad_debug_*): the package runs on tools/_bin/libvcad_ab.so in this process (`make -C videocad_amd/csrc ab`)
    L.load_ab()
from videocad_amd import bench_impl as BI       # noqa: E402

NAMES = {"0": 0, "mini_never": L.GEMM_MINI_NEVER, "dynamic": L.GEMM_DYNAMIC, "dynamic_mini_never": L.GEMM_DYNAMIC | L.GEMM_MINI_NEVER}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32); ap.add_argument("--seq", type=int, default=64); ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=4); ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--flags", nargs="*", default=["mini_never", "0"])
    ap.add_argument("--knob", default=None, help="A/B build: name of a vcad_debug_* setter; the legs are --values instead of --flags")
    ap.add_argument("--attr", default=None, help="trainer attribute set to each of --values in turn (e.g. defer_unscale)")
    ap.add_argument("--values", type=int, nargs="*", default=[0, 1])
    a = ap.parse_args()
    if a.knob or a.attr:
        a.flags = [str(v) for v in a.values]
    dev = "cuda:0"
    model, tr = BI.build_trainer(a.dtype, 0.1, dev, 0)
    bd = BI.synthetic_batch(a.batch, a.seq, 1, dev)
    eng = model._engine
    for _ in range(4):
        tr.train_step(bd)
    torch.cuda.synchronize()
    res = {f: [] for f in a.flags}
    for r in range(a.rounds):
        for f in a.flags:
            if a.attr:
                setattr(tr, a.attr, int(f))
            elif a.knob:
                getattr(L.load_ab(), a.knob)(int(f))
            else:
                eng.set_gemm_flags(NAMES[f])
            for _ in range(2):
                tr.train_step(bd)
            torch.cuda.synchronize()
            ms = []
            for _ in range(a.steps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); tr.train_step(bd); e1.record(); e1.synchronize()
                ms.append(e0.elapsed_time(e1))
            res[f].append(statistics.median(ms))
            print(f"round {r} flags={f}: {res[f][-1]:.3f} ms/step", flush=True)
    base = statistics.median(res[a.flags[0]])
    print(f"# whole train step, B = {a.batch}, T = {a.seq}, {a.dtype}, dropout 0.1; medians over {a.rounds} rounds of the per-leg median of {a.steps} steps")
    for f in a.flags:
        m = statistics.median(res[f])
        print(f"flags={f}: {m:.3f} ms ({(m / base - 1) * 100:+.2f} % vs {a.flags[0]})")


if __name__ == "__main__":
    main()
