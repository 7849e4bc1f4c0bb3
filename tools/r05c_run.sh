#!/bin/bash
# r05 GPU call c: small-shape f16 smoke, full GPU suite on the tree (deferred ViT column sums + residual add and branch dropout in the LayerNorm pass + f16 default), interleaved A/B of the builds
mkdir -p gpurun_out/r05c
timeout 120 python tools/debug_small_step.py - 2 6 f16 2>&1 | grep -v amdgpu.ids | tail -3
python -m pytest tests -m gpu -x -q > gpurun_out/r05c/gputests.txt 2>&1
timeout 900 python tools/bench_lib_ab.py tools/_bin/libvcad_hip_oldhash.so tools/_bin/libvcad_hip_B.so tools/_bin/libvcad_hip_C.so tools/_bin/libvcad_hip_D.so 3 20 > gpurun_out/r05c/abcd.txt 2>&1
tail -12 gpurun_out/r05c/gputests.txt | cut -c1-300; cat gpurun_out/r05c/abcd.txt
