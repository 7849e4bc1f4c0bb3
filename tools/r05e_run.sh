#!/bin/bash
# r05 GPU call e: Adam with a four-deep load queue (F) against the tree before it (D); fresh per-kernel table of the tree (F)
mkdir -p gpurun_out/r05e
R=$PWD
timeout 600 python tools/bench_lib_ab.py tools/_bin/libvcad_hip_D.so tools/_bin/libvcad_hip_F.so 3 20 > gpurun_out/r05e/df.txt 2>&1
(cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05e/trace -o bench -- python $R/bench.py --profile-only > $R/gpurun_out/r05e/bench_prof.json 2> $R/gpurun_out/r05e/bench_prof.err)
python tools/kernel_stats_table.py gpurun_out/r05e/trace/bench_kernel_stats.csv "C2 bf16, tree at r05 call e" > gpurun_out/r05e/kernel_stats.txt 2>&1
cat gpurun_out/r05e/df.txt; head -45 gpurun_out/r05e/kernel_stats.txt | cut -c1-160
