import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from videocad_amd import lib as L
import oputil as U
lib = L.load(); dev = "cuda:0"
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def gemm(A, W, M, N, K):
    qa, sa = U.mx8_quant_ref(A); qw, sw = U.mx8_quant_ref(W)
    qa, sa, qw, sw = qa.to(dev), sa.to(dev), qw.to(dev), sw.to(dev)
    Cm = torch.zeros(M, N, device=dev)
    assert lib.vcad_op_gemm_mx8(0, p(qa), p(sa), p(qw), p(sw), p(Cm), M, N, K, N, None, 0, None, N, st) == 0, lib.vcad_last_error()
    ref = U.mx8_dequant(qa.cpu(), sa.cpu()) @ U.mx8_dequant(qw.cpu(), sw.cpu()).t()
    return Cm.cpu().double(), ref
for name, M, N, K, fa, fw in [
    ("ones", 128, 128, 128, lambda: torch.ones(128, 128), lambda: torch.ones(128, 128)),
    ("ones K=256", 128, 128, 256, lambda: torch.ones(128, 256), lambda: torch.ones(128, 256)),
    ("A rows scaled", 128, 128, 128, lambda: torch.ones(128, 128) * torch.arange(1, 129).float()[:, None], lambda: torch.ones(128, 128)),
    ("W rows scaled", 128, 128, 128, lambda: torch.ones(128, 128), lambda: torch.ones(128, 128) * torch.arange(1, 129).float()[:, None]),
    ("A k-ramp", 128, 128, 128, lambda: torch.ones(128, 128) * (1 + (torch.arange(128) // 32).float())[None, :], lambda: torch.ones(128, 128)),
    ("A k one-hot 5", 128, 128, 128, lambda: torch.nn.functional.one_hot(torch.full((128,), 5), 128).float(), lambda: torch.arange(128).float()[None, :].repeat(128, 1) + 1),
    ("A k one-hot 70", 128, 128, 128, lambda: torch.nn.functional.one_hot(torch.full((128,), 70), 128).float(), lambda: torch.arange(128).float()[None, :].repeat(128, 1) + 1),
    ("random", 128, 128, 128, lambda: torch.randn(128, 128), lambda: torch.randn(128, 128)),
    ("random signs only", 128, 128, 128, lambda: torch.sign(torch.randn(128, 128)), lambda: torch.sign(torch.randn(128, 128))),
    ("random M=300", 300, 256, 384, lambda: torch.randn(300, 384), lambda: torch.randn(256, 384)),
]:
    torch.manual_seed(0)
    got, ref = gemm(fa(), fw(), M, N, K)
    err = float((got - ref).norm() / ref.norm())
    print(f"{name:20s} rel err {err:.3e}   got[0,:4] {got[0,:4].tolist()}  ref[0,:4] {ref[0,:4].tolist()}  got[37,66] {float(got[37 % M, 66 % N]):.4g} ref {float(ref[37 % M, 66 % N]):.4g}")
