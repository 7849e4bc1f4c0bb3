"""Loss / gradient-norm trajectory of ONE synthetic batch stream in two compute modes (same hash-init weights, same batches, same dropout seeds): does the
fp16-storage build train like an fp32-accurate mode, does its gradient scale ever move?

    python tools/mode_trajectory.py [steps] [lr] [ref_mode] [B] [T] [fresh]

    steps     train steps (default 60)
    lr        learning rate (default 1e-5 = the reference's, main.py:80)
    ref_mode  bf16 | bf16x3 | f32 (default bf16x3: fp32 tensors, logits 6.9e-6 from the reference — fp32 accuracy at half the time of f32)
    B, T      batch shape (default 32 64 = BASELINE configs[1])
    fresh     1 (default) = a NEW synthetic batch every step (seed 1000 + i); 0 = eight batches cycled (r04's form)

r05: >= 2 000 steps on a stream of distinct batches at the reference learning rate and at 10x; the summary block at the end is what
profiles/r05_f16_trajectory_*.txt keep.  The two modes run interleaved step by step (both models are resident: 2 x ~16 GB at C2)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocad_amd import bench_impl as BI

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
ref = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
T = int(sys.argv[5]) if len(sys.argv) > 5 else 64
fresh = (int(sys.argv[6]) if len(sys.argv) > 6 else 1) != 0
dev = torch.device("cuda:0")

runs = {}
for dtype in (ref, "f16"):
    model, tr = BI.build_trainer(dtype, 0.1, dev, 0)
    tr.optimizer.param_groups[0]["lr"] = lr
    runs[dtype] = (model, tr)
rows = {ref: [], "f16": []}
scale_hist = []
losses = {ref: [], "f16": []}
norms = []
for i in range(steps):
    bd = BI.synthetic_batch(B, T, 1000 + (i if fresh else i % 8), dev)
    for dtype in (ref, "f16"):
        model, tr = runs[dtype]
        loss, _ = tr.train_step(bd)
        losses[dtype].append(loss.detach())                   # device scalars: no host sync inside the loop
        if dtype == "f16":
            norms.append(tr._last_norm[0].detach().clone())
            scale_hist.append(model._engine.grad_scale)         # host value (changes only when the trainer's watch halves / regrows it)
    del bd
torch.cuda.synchronize()
lr_ = torch.stack(losses[ref]).float().cpu(); lf = torch.stack(losses["f16"]).float().cpu(); gn = torch.stack(norms).float().cpu()
e16 = runs["f16"][0]._engine
print(f"# {steps} train steps at B={B}, T={T}, dropout 0.1, lr {lr:g}; {'a new synthetic batch every step' if fresh else 'eight synthetic batches cycled'}; "
      f"same hash-init weights and dropout seeds in both modes; reference mode = {ref}")
print(f"# step | loss {ref} | loss f16 | rel diff | |g| f16 (pre-clip) | f16 gradient scale")
every = max(1, steps // 40)
for i in range(steps):
    if i < 5 or i % every == every - 1 or i == steps - 1:
        print(f"{i + 1:5d} | {float(lr_[i]):9.4f} | {float(lf[i]):9.4f} | {abs(float(lf[i]) - float(lr_[i])) / abs(float(lr_[i])):8.2e} | {float(gn[i]):9.3f} | {scale_hist[i]:.0f}")
rel = ((lf - lr_).abs() / lr_.abs())
win = max(1, steps // 10)
sm = lambda x: torch.stack([x[j:j + win].mean() for j in range(0, steps - win + 1, win)])
print(f"# every fp16 gradient norm finite: {bool(torch.isfinite(gn).all())}; updates skipped: {getattr(e16, 'skipped_steps', 0)}; "
      f"scale history: {sorted(set(scale_hist))} (end: {scale_hist[-1]:.0f}); Adam step count f16 {e16.step_count} / {ref} {runs[ref][0]._engine.step_count}")
print(f"# |loss_f16 - loss_{ref}| / loss: max {float(rel.max()):.3e}, mean {float(rel.mean()):.3e}, last {win} steps mean {float(rel[-win:].mean()):.3e}")
print(f"# {win}-step mean losses {ref}: " + " ".join(f"{float(v):.4f}" for v in sm(lr_)))
print(f"# {win}-step mean losses f16:    " + " ".join(f"{float(v):.4f}" for v in sm(lf)))
wr = float((runs['f16'][0]._engine.params - runs[ref][0]._engine.params).norm() / runs[ref][0]._engine.params.norm())
print(f"# weights after {steps} steps: |w_f16 - w_{ref}| / |w_{ref}| = {wr:.3e}")
