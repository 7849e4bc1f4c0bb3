"""rocprofv3 --kernel-trace --stats CSV -> per-step table of the library's kernels (steps counted from the once-per-step adam_kernel)."""
import csv
import sys

path, title = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
rows = list(csv.DictReader(open(path)))
steps = [int(r["Calls"]) for r in rows if r["Name"].startswith("adam_kernel")][0]
ours = [r for r in rows if "at::native" not in r["Name"] and "rocclr" not in r["Name"] and "elementwise_kernel" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in ours) / steps / 1e6
print(f"# {title}: {steps} train steps under the tracer, {tot:.2f} ms of library kernel time per step (streams overlap: more than the step's wall time)\n")
print("| ms / step | calls / step | avg us | kernel |\n|---|---|---|---|")
for r in sorted(ours, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print(f"| {float(r['TotalDurationNs']) / 1e6 / steps:.3f} | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | `{r['Name'].replace('void ', '').split('(')[0][:120]}` |")
