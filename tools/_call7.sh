mkdir -p gpurun_out/c7
timeout 60 python tools/x3_bench.py attn
timeout 300 python -m pytest tests -m gpu -q -x -k "attention or x3 or f32_step or long or evaluate or rgb8" > gpurun_out/c7/pytest_new.log 2>&1; echo "new rc=$?"; tail -3 gpurun_out/c7/pytest_new.log
timeout 200 python bench.py --dtype bf16x3 --steps 5 --warmup 2 --no-seq186 --no-pcie --no-cpu-baseline --no-modes > gpurun_out/c7/bench_x3.json 2> gpurun_out/c7/bench_x3.err; python - <<PY
import json
d=json.loads(open('gpurun_out/c7/bench_x3.json').read().strip().splitlines()[-1])
print('x3', d['value'], d['ms_per_step'], d['parity'])
print({k:v['ms'] for k,v in d['kernel_breakdown'].items()})
PY
