#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box.  Usage: VCAD_COMMIT=<sha> tools/collect_profiles.sh <tag>   (writes gpurun_out/<tag>/)
# 1) kernel trace + stats of the DEFAULT bench command; 2) separate PMC passes (HBM read / write bytes; matrix-core busy cycles IN THE
# MODEL) of a short bench run; 3) kernel-trace stats of the two other quoted shapes / modes (seq_len 186 at 16 clips; bf16x3 at C2).
# Every pass runs under its own `timeout`: a counter set the tool rejects aborts that pass, not the call.  PMC passes never carry trace
# domains other than --kernel-trace.
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
# --profile-only: ONLY the C2 train steps run (no seq-186 leg, no PCIe leg, no parity forward, no CPU baseline), so every traced kernel
# belongs to a C2 step; the summariser counts the steps that actually ran under the tracer (adam_kernel launches once per step) instead of assuming them
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --profile-only > $OUT/bench_default.json 2> $OUT/bench_default.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 1 --profile-only > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
# matrix-core occupancy per kernel inside the train step (r02 had it for isolated launches only): SQ_VALU_MFMA_BUSY_CYCLES against the
# kernel's own GRBM_GUI_ACTIVE (summed over the 8 XCDs) x 256 CUs x 4 SIMDs / 8
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_MFMA -o pmc -- python $R/bench.py --steps 1 --warmup 1 --profile-only > $OUT/pmc_MFMA.json 2> $OUT/pmc_MFMA.err
# the other quoted shapes / modes: per-kernel stats only
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_t186 -o t186 -- python $R/bench.py --profile-only --batch 16 --seq 186 --steps 6 --warmup 2 > $OUT/bench_t186.json 2> $OUT/bench_t186.err
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_x3 -o x3 -- python $R/bench.py --profile-only --dtype bf16x3 --steps 3 --warmup 1 > $OUT/bench_x3.json 2> $OUT/bench_x3.err
# the fp16-storage build (libvcad_hip_f16.so): per-kernel stats + matrix-core occupancy of the same step
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_f16 -o f16 -- python $R/bench.py --profile-only --dtype f16 --steps 6 --warmup 2 > $OUT/bench_f16.json 2> $OUT/bench_f16.err
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_MFMA_f16 -o pmc -- python $R/bench.py --dtype f16 --steps 1 --warmup 1 --profile-only > $OUT/pmc_MFMA_f16.json 2> $OUT/pmc_MFMA_f16.err
cd $R
python tools/summarize_profiles.py $TAG > $OUT/summary.md 2> $OUT/summary.err
python tools/kernel_stats_table.py $OUT/trace_t186/t186_kernel_stats.csv "seq_len 186, 16 clips (BASELINE configs[3] per-GPU shape), bf16" > $OUT/t186_kernel_stats.txt 2>> $OUT/summary.err
python tools/kernel_stats_table.py $OUT/trace_x3/x3_kernel_stats.csv "C2 (32 clips x 64 steps), bf16x3" > $OUT/x3_kernel_stats.txt 2>> $OUT/summary.err
python tools/kernel_stats_table.py $OUT/trace_f16/f16_kernel_stats.csv "C2 (32 clips x 64 steps), f16 (libvcad_hip_f16.so)" > $OUT/f16_kernel_stats.txt 2>> $OUT/summary.err
tail -1 $OUT/bench_default.json | cut -c1-400
