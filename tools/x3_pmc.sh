#!/bin/bash
# PMC passes over tools/x3_pmc.py (bf16x3 GEMMs + fp32 attention); output: gpurun_out/x3_pmc/... + a per-dispatch summary on stdout
R=$PWD; OUT=$R/gpurun_out/x3_pmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o p -- python $R/tools/x3_pmc.py > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq2 -o p -- python $R/tools/x3_pmc.py > $OUT/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/rd -o p -- python $R/tools/x3_pmc.py > $OUT/rd.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/wr -o p -- python $R/tools/x3_pmc.py > $OUT/wr.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $OUT/tcp -o p -- python $R/tools/x3_pmc.py > $OUT/tcp.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for sub in ("sq", "sq2", "rd", "wr", "tcp"):
    for f in glob.glob(f"gpurun_out/x3_pmc/{sub}/**/*counter_collection.csv", recursive=True):
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not ("gemm_kernel" in k or "attn_f32" in k): continue
            key = (k[:70], r["Dispatch_Id"])
            agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        print("==", sub)
        seen = set()
        for (k, d), v in agg.items():
            if k in seen: continue            # first launch of each kernel only
            seen.add(k)
            extra = ""
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
                extra = f"  MFMA_UTIL={v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 256 * 4):.3f}"
            print(d, k, " ".join(f"{a}={b:.4g}" for a, b in v.items()) + extra)
PY
tail -3 $OUT/tcp.log
