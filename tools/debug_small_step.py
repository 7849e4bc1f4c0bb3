"""debug: train steps at a small shape with a device sync after every call; argv[1] = library path ('-' = in-tree), argv[2..3] = B T, argv[4] = bf16 | f16"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videocad_amd import lib as L
if len(sys.argv) > 1 and sys.argv[1] != "-":
    L.LIB_PATH = sys.argv[1]
from videocad_amd import synth
from videocad_amd.engine import NativeEngine, make_config
from videocad_amd.bench_impl import CANONICAL, init_weights
from oracle import restatement as O
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
T = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda:0")
DT = sys.argv[4] if len(sys.argv) > 4 else "bf16"
if DT == "f16" and len(sys.argv) > 1 and sys.argv[1] != "-":
    L.LIB_PATH_F16 = sys.argv[1]
eng = NativeEngine(make_config(dtype=L.VCAD_F16 if DT == "f16" else L.VCAD_BF16, **CANONICAL), dev)
init_weights(eng, dev)
eng.set_dropout(0.1, 5)
def sync(tag):
    torch.cuda.synchronize(); print("ok:", tag, flush=True)
b = synth.make_batch_torch(B, T, 5, dev)
for it in range(2):
    cmds, pars = eng.forward(b["frames"][:, :-1], O.normalize_actions(b["actions"][:, :-1]), b["cad_image"]); sync(f"forward {it}")
    eng.loss(cmds, pars, b["actions"][:, 1:], [0.2] * 5); sync("loss")
    for st in range(len(eng.buckets)):
        eng.backward(stage=st); sync(f"stage {st}")
    eng.optimizer_step(); sync("adam")
    eng.backward(); sync("whole backward")
print("done", float(pars.float().abs().mean()))
