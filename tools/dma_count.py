import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocad_amd import lib as L, bench_impl as BI
from videocad_amd.engine import NativeEngine, make_config
dev = torch.device("cuda:0")
eng = NativeEngine(make_config(dtype=L.VCAD_BF16, **BI.CANONICAL), dev); BI.init_weights(eng, dev)
fr, ac, cad = BI.synthetic_batch(32, 64, 1, dev)
st = BI.Stepper(eng, 1, 0, dropout=0.1)
st.step(fr, ac, cad); torch.cuda.synchronize()
n0 = eng.lib.vcad_debug_gemm_dma_launches()
st.step(fr, ac, cad); torch.cuda.synchronize()
print("dma launches per step:", eng.lib.vcad_debug_gemm_dma_launches() - n0)
