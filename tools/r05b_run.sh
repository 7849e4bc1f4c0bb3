#!/bin/bash
# r05 GPU call b: fp16 overflow drill + dropout parity on the device, interleaved A/B of the dropout-hash change, f16 trajectories (2 000 / 1 000 steps)
mkdir -p gpurun_out/r05b
python -m pytest tests/test_f16_engine_gpu.py -q -x -s -k "drill or overflow" > gpurun_out/r05b/drill.txt 2>&1
python -m pytest tests/test_engine_gpu.py -q -x -k "dropout or train_mode" > gpurun_out/r05b/dropout_tests.txt 2>&1
timeout 600 python tools/bench_lib_ab.py tools/_bin/libvcad_hip_oldhash.so - 3 20 > gpurun_out/r05b/hash_ab.txt 2>&1
timeout 900 python tools/mode_trajectory.py 2000 1e-5 bf16x3 32 64 1 > gpurun_out/r05b/traj_lr1e-5.txt 2>&1
timeout 600 python tools/mode_trajectory.py 1000 1e-4 bf16x3 32 64 1 > gpurun_out/r05b/traj_lr1e-4.txt 2>&1
tail -4 gpurun_out/r05b/drill.txt; tail -3 gpurun_out/r05b/dropout_tests.txt; cat gpurun_out/r05b/hash_ab.txt; tail -7 gpurun_out/r05b/traj_lr1e-5.txt; tail -7 gpurun_out/r05b/traj_lr1e-4.txt
