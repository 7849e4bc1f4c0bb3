#!/bin/bash
# r05 GPU call c: full GPU suite on the tree (deferred ViT column sums + residual add in the LayerNorm pass + f16 default), then the interleaved A/B of the three builds
mkdir -p gpurun_out/r05c
python -m pytest tests -m gpu -x -q > gpurun_out/r05c/gputests.txt 2>&1
timeout 900 python tools/bench_lib_ab.py tools/_bin/libvcad_hip_oldhash.so tools/_bin/libvcad_hip_B.so tools/_bin/libvcad_hip_C.so 3 20 > gpurun_out/r05c/abc.txt 2>&1
tail -12 gpurun_out/r05c/gputests.txt; cat gpurun_out/r05c/abc.txt
