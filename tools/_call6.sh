mkdir -p gpurun_out/c6
timeout 400 python -m pytest tests -m gpu -q -s > gpurun_out/c6/pytest_all.log 2>&1; echo "all rc=$?"; grep -h "measured" gpurun_out/c6/pytest_all.log | head -30; tail -4 gpurun_out/c6/pytest_all.log
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/c6/bench_default.json 2> gpurun_out/c6/bench_default.err; tail -c 4500 gpurun_out/c6/bench_default.json
timeout 120 python tools/cpu_thread_sweep.py > gpurun_out/c6/cpu_thread_sweep.txt 2>&1; cat gpurun_out/c6/cpu_thread_sweep.txt
bash tools/collect_profiles.sh r03 > gpurun_out/c6/collect.log 2>&1; tail -3 gpurun_out/c6/collect.log; tail -30 gpurun_out/r03/summary.md
