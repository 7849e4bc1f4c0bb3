#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box.  Usage: tools/collect_profiles.sh <tag>   (writes gpurun_out/<tag>/)
# 1) kernel trace + stats of the DEFAULT bench command; 2) separate PMC passes (HBM read / write bytes) of a short bench run.
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
# --profile-only: ONLY the C2 train steps run (no seq-186 leg, no PCIe leg, no parity forward, no CPU baseline), so every traced kernel
# belongs to a C2 step; the summariser counts the steps that actually ran (adam_kernel launches once per step) instead of assuming them
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --profile-only > $OUT/bench_default.json 2> $OUT/bench_default.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 1 --profile-only > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
cd $R
python tools/summarize_profiles.py $TAG > $OUT/summary.md 2> $OUT/summary.err
tail -1 $OUT/bench_default.json | cut -c1-400
