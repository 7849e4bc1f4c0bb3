#!/bin/bash
# r05 GPU call f: the weight-gradient kernel on a four-stage ring of 32-deep stages (G) against the tree before it (D): op-level parity on the device, A/B at C2 and T=186, per-kernel durations
mkdir -p gpurun_out/r05f
R=$PWD
python -m pytest tests/test_ops_gpu.py tests/test_f16_ops_gpu.py -x -q -k "gemm" > gpurun_out/r05f/ops_tests.txt 2>&1
python -m pytest tests/test_engine_gpu.py tests/test_f16_engine_gpu.py -x -q -k "c1_full or bf16_step or nhead8_large" > gpurun_out/r05f/engine_tests.txt 2>&1
timeout 600 python tools/bench_lib_ab.py tools/_bin/libvcad_hip_D.so tools/_bin/libvcad_hip_G.so 3 20 > gpurun_out/r05f/dg_c2.txt 2>&1
timeout 600 python tools/bench_lib_ab.py tools/_bin/libvcad_hip_D.so tools/_bin/libvcad_hip_G.so 2 10 --batch 16 --seq 186 > gpurun_out/r05f/dg_t186.txt 2>&1
(cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05f/trace -o bench -- python $R/bench.py --profile-only > $R/gpurun_out/r05f/bench_prof.json 2> $R/gpurun_out/r05f/bench_prof.err)
python tools/kernel_stats_table.py gpurun_out/r05f/trace/bench_kernel_stats.csv "C2 bf16, tree at r05 call f" > gpurun_out/r05f/kernel_stats.txt 2>&1
tail -3 gpurun_out/r05f/ops_tests.txt | cut -c1-200; tail -3 gpurun_out/r05f/engine_tests.txt | cut -c1-200; cat gpurun_out/r05f/dg_c2.txt gpurun_out/r05f/dg_t186.txt; grep "gemm_dma" gpurun_out/r05f/kernel_stats.txt
