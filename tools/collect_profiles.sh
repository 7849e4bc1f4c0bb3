#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box.  Usage: tools/collect_profiles.sh <tag>   (writes gpurun_out/<tag>/)
# 1) kernel trace + stats of the DEFAULT bench command; 2) separate PMC passes (HBM read / write bytes) of a short bench run.
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
cd $R
python tools/summarize_profiles.py $TAG > $OUT/summary.md 2> $OUT/summary.err
tail -1 $OUT/bench_default.json | cut -c1-400
