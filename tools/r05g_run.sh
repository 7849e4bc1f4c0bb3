#!/bin/bash
# r05 GPU call g: the driver's commands on the final tree — full GPU suite, smoke(), bench line
mkdir -p gpurun_out/r05g
python -m pytest tests -m gpu -x -q > gpurun_out/r05g/gputests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05g/smoke.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r05g/bench.json 2> gpurun_out/r05g/bench.err
tail -4 gpurun_out/r05g/gputests.txt | cut -c1-200; tail -2 gpurun_out/r05g/smoke.txt; tail -c 3000 gpurun_out/r05g/bench.json
