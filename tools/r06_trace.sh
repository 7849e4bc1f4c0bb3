#!/bin/bash
# kernel trace of the headline steps + per-launch timeline of one step.  Usage: tools/r06_trace.sh <tag> [extra bench flags]
TAG=${1:-r06a}; shift
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --profile-only --steps 6 --warmup 3 "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $R
python tools/step_timeline.py $(ls $OUT/trace/*/*kernel_trace.csv $OUT/trace/*kernel_trace.csv 2>/dev/null | head -1) --all > $OUT/timeline.txt 2> $OUT/timeline.err
tail -1 $OUT/bench.json | cut -c1-300
