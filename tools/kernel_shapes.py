"""Per-(kernel, grid) durations from a rocprofv3 kernel_trace.csv: which SHAPE of a kernel family costs what.  Usage: kernel_shapes.py <trace.csv> <title> [filter substring ...]"""
import collections
import csv
import sys

path, title, filt = sys.argv[1], sys.argv[2], sys.argv[3:]
rows = list(csv.DictReader(open(path)))
steps = sum(1 for r in rows if r["Kernel_Name"].startswith("adam_kernel"))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if filt and not any(f in n for f in filt):
        continue
    wg = int(r["Workgroup_Size_X"])
    key = (n, int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"# {title}: {steps} train steps under the tracer; grid = workgroups (x, y, z)\n")
print("| ms / step | calls / step | avg us | min us | kernel | grid |\n|---|---|---|---|---|---|")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f"| {sum(v) / steps / 1e3:.3f} | {len(v) / steps:.1f} | {sum(v) / len(v):.1f} | {min(v):.1f} | `{k[0][:110]}` | {k[1]} x {k[2]} x {k[3]} |")
