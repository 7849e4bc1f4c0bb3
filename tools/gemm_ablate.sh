#!/bin/bash
# Builds ablated copies of the library (GD_ABLATE bits, see gemm_dma.h) into tools/_bin/ — run here, then `gpurun -- python tools/gemm_bench.py ablate`
set -e
cd "$(dirname "$0")/../videocad_amd/csrc"
make -s -j8 all
mkdir -p ../../tools/_bin
for B in 1 2 4 3 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DGD_ABLATE=$B -c ops_gemm_dma.hip -o build/abl$B.o &
done
wait
for B in 1 2 4 3 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "emu_\|abl\|ops_gemm_dma.o") build/abl$B.o -o ../../tools/_bin/libvcad_abl$B.so
done
ls -la ../../tools/_bin/
