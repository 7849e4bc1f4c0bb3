"""bench.py on the A/B build (tools/_bin/libvcad_ab.so, `make -C videocad_amd/csrc ab`): the process-global selectors of csrc/ab.h — the
slower kernel variants kept for the measurements under profiles/ — are set here, every other flag goes to bench.py unchanged.

    python tools/bench_ab.py --attn-variant 1 --split-gelu 0 -- --steps 10 --no-modes --no-cpu-baseline
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videocad_amd import lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gemm-policy", type=int, default=0, help="r02 dispatcher rules: 1 = activation epilogues stay off the persistent kernel, 2 = small-tile-count wgrads too")
ap.add_argument("--attn-variant", type=int, default=0, help="1 = r01 attention kernels")
ap.add_argument("--split-gelu", type=int, default=1, help="0 = GELU / GELU' fused into the MLP GEMM epilogues (r01)")
ap.add_argument("--gemm-waves", type=int, default=8, help="4 = four-wave form of the 256-wide tile")
ap.add_argument("--gemm-epilogue", type=int, default=-1, help="0 row-per-lane, 1 column-per-lane")
ap.add_argument("--gemm-variant", type=int, default=0, help="1 = ping-pong persistent kernel")
ap.add_argument("rest", nargs="*")
a = ap.parse_args()
lib = L.load_ab()
lib.vcad_debug_gemm_policy(a.gemm_policy); lib.vcad_debug_attn_variant(a.attn_variant); lib.vcad_debug_split_gelu(a.split_gelu)
lib.vcad_debug_gemm_waves(a.gemm_waves); lib.vcad_debug_gemm_epilogue(a.gemm_epilogue); lib.vcad_debug_gemm_variant(a.gemm_variant)
import bench  # noqa: E402

bench.main(a.rest)
