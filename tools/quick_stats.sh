#!/bin/bash
# per-kernel stats of a short profile-only bench run: tools/quick_stats.sh <tag> [bench args]
TAG=$1; shift
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o q -- python $R/bench.py --profile-only --steps 6 --warmup 2 "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $R
python - <<EOF
import csv
rows=list(csv.DictReader(open('$OUT/trace/q_kernel_stats.csv')))
steps=[int(r['Calls']) for r in rows if r['Name'].startswith('adam_kernel')][0]
for r in rows[:45]:
    print(f"{float(r['TotalDurationNs'])/1e6/steps:7.3f} ms {int(r['Calls'])/steps:6.1f} x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
EOF
