#!/bin/bash
# PMC passes over tools/attn_bench.py (a few dozen dispatches — a whole train step under many SQ counters takes > 15 min)
R=$PWD; OUT=$R/gpurun_out/attn_pmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/attn_bench.py"
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o p -- $CMD > $OUT/sq.log 2>&1
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR --kernel-trace --output-format csv -d $OUT/in -o p -- $CMD > $OUT/in.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/mem -o p -- $CMD > $OUT/mem.log 2>&1
cd $R
for sub in sq in mem; do for f in $(find gpurun_out/attn_pmc/$sub -name "*counter_collection.csv"); do python tools/summarize_pmc.py $f | grep -i "kernel |\|attn"; done; done
