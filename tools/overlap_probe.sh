#!/bin/bash
# r06 probe: is there aggregate throughput to be had from overlapping two half-batch steps (MFMA-bound GEMMs of one beside HBM-bound passes of the other)?
# Runs one bench process at B=32, one at B=16, then TWO B=16 processes concurrently on the same GPU, and prints frames/s of each.
R=$PWD; OUT=$R/gpurun_out/${1:-r06_overlap}; mkdir -p $OUT
one() { python $R/bench.py --profile-only --steps ${3:-40} --warmup 5 --batch $1 --seq 64 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$2', j['value'], j['ms_per_step'])"; }
one 32 solo_b32 | tee $OUT/solo32.txt
one 16 solo_b16 | tee $OUT/solo16.txt
( one 16 pairA 80 > $OUT/pairA.txt ) &
( one 16 pairB 80 > $OUT/pairB.txt ) &
wait
cat $OUT/pairA.txt $OUT/pairB.txt
