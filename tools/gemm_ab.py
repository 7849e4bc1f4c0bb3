"""In-model A/B of the two bf16 GEMM kernels: pairs the GEMM launches of one train step from two rocprofv3 kernel traces of
`bench.py --gemm-dma 0` (register-staged only) and `--gemm-dma 1|-1` and prints a markdown table.
Usage: python tools/gemm_ab.py gpurun_out/ab0 gpurun_out/ab-1"""
import csv
import sys


def load(d):
    rows = list(csv.DictReader(open(f"{d}/b_kernel_trace.csv")))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
    a, b = idx[-3], idx[-2]
    return [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[a + 1:b]
            if "gemm" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]]


o, n = load(sys.argv[1]), load(sys.argv[2])
assert len(o) == len(n)
print("| # | register-staged kernel | us | persistent DMA kernel | us | ratio |\n|---|---|---|---|---|---|")
to = tn = 0.0
for i, ((a, da), (b, db)) in enumerate(zip(o, n)):
    if "gemm_dma" in b:
        to += da; tn += db
        print(f"| {i} | `{a[5:70]}` | {da:.1f} | `{b[5:45]}` | {db:.1f} | {db / da:.2f} |")
print(f"\nsum over the launches that moved: {to:.0f} us -> {tn:.0f} us")
