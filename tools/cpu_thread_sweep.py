"""Host-thread sweep of the CPU baseline (the oracle restatement, one full train step at B=2, T=64): justifies the thread count
bench.py's `cpu_baseline` uses on the GPU box's host.  Usage: python tools/cpu_thread_sweep.py > profiles/rNN_cpu_thread_sweep.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as O
from videocad_amd import synth

shapes = O.param_shapes()
weights = {k: synth.make_param_torch(k, s, "cpu").numpy() for k, s in shapes.items()}
batch = synth.make_batch(2, 64, seed=1)
print(f"host cores: {os.cpu_count()}; torch {torch.__version__}; one oracle train step (fwd + loss + bwd + clip + Adam), B=2, T=64, fp32")
print("| threads | s / step | frames/s |\n|---|---|---|")
for th in [t for t in (4, 8, 16, 32, 64, 128, 256) if t <= (os.cpu_count() or 1)]:
    torch.set_num_threads(th)
    ot = O.OracleTrainer(weights)
    ot.step(batch)
    t0 = time.time(); n = 0
    while n < 2:
        ot.step(batch); n += 1
    dt = (time.time() - t0) / n
    print(f"| {th} | {dt:.2f} | {2 * 64 / dt:.1f} |", flush=True)
