"""Loss / gradient-norm trajectory of the same synthetic batch stream in two compute modes (same weights, same dropout seeds): does the fp16-storage build train like
the bf16 one, does its gradient scale ever move?   python tools/mode_trajectory.py [steps] [lr]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocad_amd import bench_impl as BI

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5          # the reference's learning rate
dev = torch.device("cuda:0")
B, T = 32, 64
out = {}
for dtype in ("bf16", "f16"):
    model, tr = BI.build_trainer(dtype, 0.1, dev, 0)
    tr.optimizer.param_groups[0]["lr"] = lr
    eng = model._engine
    rows = []
    for i in range(steps):
        bd = BI.synthetic_batch(B, T, 1000 + i % 8, dev)    # eight batches, cycled
        loss, _ = tr.train_step(bd)
        norm = tr._last_norm if dtype == "f16" else None
        rows.append((float(loss), float(norm[0]) if norm is not None else float("nan"), eng.grad_scale))
    out[dtype] = rows
    del tr, model
    torch.cuda.empty_cache()
print(f"# {steps} train steps at B={B}, T={T}, dropout 0.1, lr {lr:g}, eight synthetic batches cycled; same hash-init weights and dropout seeds in both modes")
print("# step | loss bf16 | loss f16 | |g| f16 (pre-clip) | f16 gradient scale")
for i in range(steps):
    if i < 10 or i % 5 == 4:
        print(f"{i + 1:4d} | {out['bf16'][i][0]:9.4f} | {out['f16'][i][0]:9.4f} | {out['f16'][i][1]:9.3f} | {out['f16'][i][2]:.0f}")
fin = all(r[1] == r[1] and abs(r[1]) != float('inf') for r in out['f16'])
print(f"# every fp16 gradient norm finite: {fin}; scale at the end: {out['f16'][-1][2]:.0f}; max |loss_f16 - loss_bf16| / loss: "
      f"{max(abs(a[0] - b[0]) / abs(b[0]) for a, b in zip(out['f16'], out['bf16'])):.3e}")
