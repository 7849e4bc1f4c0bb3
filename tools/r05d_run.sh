#!/bin/bash
# r05 GPU call d: decoder residual adds in the post-norm LayerNorm pass (E) against D and A at C2 and at configs[3]'s per-GPU shape; 16-bit golden tests on the tree
mkdir -p gpurun_out/r05d
python -m pytest tests/test_engine_gpu.py tests/test_f16_engine_gpu.py tests/test_boundary_gpu.py -x -q -k "bf16 or f16 or dropout" > gpurun_out/r05d/tests.txt 2>&1
timeout 600 python tools/bench_lib_ab.py tools/_bin/libvcad_hip_oldhash.so tools/_bin/libvcad_hip_D.so tools/_bin/libvcad_hip_E.so 3 20 > gpurun_out/r05d/ade_c2.txt 2>&1
timeout 600 python tools/bench_lib_ab.py tools/_bin/libvcad_hip_oldhash.so tools/_bin/libvcad_hip_D.so tools/_bin/libvcad_hip_E.so 2 10 --batch 16 --seq 186 > gpurun_out/r05d/ade_t186.txt 2>&1
tail -4 gpurun_out/r05d/tests.txt | cut -c1-300; cat gpurun_out/r05d/ade_c2.txt gpurun_out/r05d/ade_t186.txt
