"""ViT attention forward / backward through the C ABI at the C2 frame-ViT shape (2048 frames x 16 heads x 50 tokens x 64).  Usage: python tools/attn_bench.py [variant]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocad_amd import lib as L

lib = L.load_ab()
dev = "cuda:0"
B, H, T, D = 2048, 16, 50, 64
if len(sys.argv) > 1:
    lib.vcad_debug_attn_variant(int(sys.argv[1]))
qkv = (torch.randn(B, T, 3, H, D, device=dev) * 0.5).to(torch.bfloat16)
o = torch.empty(B, T, H, D, dtype=torch.bfloat16, device=dev)
do = torch.randn(B, T, H, D, device=dev).to(torch.bfloat16)
dqkv = torch.zeros_like(qkv)
lse = torch.empty(B, H, T, device=dev); delta = torch.empty(B, H, T, device=dev)
es = 2; ld = 3 * H * D; base = qkv.data_ptr(); db = dqkv.data_ptr()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
scale = D ** -0.5


def fwd():
    assert lib.vcad_op_attention_fwd(1, D, C.c_void_p(base), C.c_void_p(base + H * D * es), C.c_void_p(base + 2 * H * D * es), p(o), ld, ld, ld, H * D, p(lse),
                                     B, H, T, T, T, 0, scale, st) == 0


def bwd():
    assert lib.vcad_op_attention_bwd_o(1, D, C.c_void_p(base), C.c_void_p(base + H * D * es), C.c_void_p(base + 2 * H * D * es), p(o), H * D, p(do), ld, ld, ld, H * D,
                                       p(lse), p(delta), C.c_void_p(db), C.c_void_p(db + H * D * es), C.c_void_p(db + 2 * H * D * es), ld, ld, ld,
                                       B, H, T, T, T, 0, scale, st) == 0


for fn, name, nbytes in ((fwd, "fwd", 4 * B * T * H * D * 2), (bwd, "bwd", 7 * B * T * H * D * 2)):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"vit attention {name}: {ms * 1e3:7.1f} us   {nbytes / ms / 1e9:6.2f} TB/s of algorithmic bytes", flush=True)
