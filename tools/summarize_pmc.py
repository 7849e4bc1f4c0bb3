"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name.  Usage: summarize_pmc.py <csv> [<csv> ...]"""
import csv
import collections
import sys

for path in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name") or row.get("Kernel Name")
            name = k.split("(")[0][:90]
            agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (row.get("Dispatch_Id"), name)
            if key not in seen:
                seen.add(key); cnt[name] += 1
    print(f"## {path}")
    counters = sorted({c for v in agg.values() for c in v})
    print("| kernel | dispatches | " + " | ".join(counters) + " |")
    print("|---|---|" + "---|" * len(counters))
    rows = sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))
    for name, v in rows[:40]:
        print(f"| `{name}` | {cnt[name]} | " + " | ".join(f"{v.get(c, 0):.4g}" for c in counters) + " |")
    print()
