#!/bin/bash
# PMC passes over tools/gemm_pmc.py; output: gpurun_out/gemm_pmc/<pass>/...csv + a per-kernel summary
R=$PWD; OUT=$R/gpurun_out/gemm_pmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o p -- python $R/tools/gemm_pmc.py > $OUT/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/rd -o p -- python $R/tools/gemm_pmc.py > $OUT/rd.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/wr -o p -- python $R/tools/gemm_pmc.py > $OUT/wr.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for sub in ("sq", "rd", "wr"):
    for f in glob.glob(f"gpurun_out/gemm_pmc/{sub}/**/*counter_collection.csv", recursive=True):
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm" not in k or "reduce" in k: continue
            key = (k[:60], r["Dispatch_Id"])
            agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        print("==", sub)
        for (k, d), v in agg.items():
            extra = ""
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:      # GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs
                extra = f"  MFMA_UTIL={v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 256 * 4):.3f}"
            print(d, k, " ".join(f"{a}={b:.4g}" for a, b in v.items()) + extra)
PY
