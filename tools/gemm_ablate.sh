#!/bin/bash
# Builds ablated copies of the library (GD_ABLATE bits, see gemm_dma.h) into tools/_bin/ — A/B objects + an ablated ops_gemm_dma; run here, then on the GPU box `VCAD_AB_LIB=tools/_bin/libvcad_abl<bits>.so python tools/gemm_bench.py mainloop`
set -e
cd "$(dirname "$0")/../videocad_amd/csrc"
make -s -j8 ab
mkdir -p ../../tools/_bin
for B in 1 2 4 3 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVCAD_AB -DGD_ABLATE=$B -c ops_gemm_dma.hip -o build/abl$B.o &
done
wait
for B in 1 2 4 3 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/ab_*.o | grep -v "ab_ops_gemm_dma.o") build/abl$B.o -o ../../tools/_bin/libvcad_abl$B.so
done
ls -la ../../tools/_bin/
