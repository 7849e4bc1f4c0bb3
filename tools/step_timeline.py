"""Ordered per-launch timeline of ONE train step from a rocprofv3 kernel_trace.csv (the last complete step between two adam_kernel launches).

    python tools/step_timeline.py <kernel_trace.csv> [--all]

Prints the step's phases (frame-ViT forward, decoder forward, loss, decoder backward, stem, ViT backward, optimiser — cut at marker kernels on the
main queue), the busy time per queue and, per phase, kernel-time by kernel name; with --all every launch as `start_us dur_us queue grid name`."""
import collections
import csv
import re
import sys


def short(n):
    n = n.split("(")[0].replace("void ", "")
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n[:100]


def main():
    path = sys.argv[1]
    show_all = "--all" in sys.argv[2:]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
    if len(adam) < 2:
        print("need two optimiser launches in the trace")
        return
    # (bench.py's last step is the HIP-event-profiled one: single stream, events around every launch — take an earlier, undisturbed step)
    k = -3 if len(adam) >= 4 else -1
    for a in sys.argv[2:]:
        if a.startswith("--step="):
            k = int(a.split("=")[1])
    lo, hi = adam[k - 1] + 1, adam[k] + 1
    step = rows[lo:hi]
    t0 = int(step[0]["Start_Timestamp"])
    qs = collections.Counter(r["Queue_Id"] for r in step)
    mainq = qs.most_common(1)[0][0]
    ev = []
    for r in step:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        wg = max(1, int(r["Workgroup_Size_X"]))
        ev.append(dict(s=s / 1e3, d=(e - s) / 1e3, q=r["Queue_Id"], n=short(r["Kernel_Name"]), g=int(r["Grid_Size_X"]) // wg * max(1, int(r["Grid_Size_Y"])) * max(1, int(r["Grid_Size_Z"]))))
    total = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
    print(f"# step: {len(ev)} launches, {total / 1e3:.3f} ms from the first launch to the end of adam_kernel; queues: {dict(qs)} (main = {mainq})")
    # phases on the main queue, cut at marker kernels
    main = [e for e in ev if e["q"] == mainq]
    def first(pred, start=0):
        for i in range(start, len(main)):
            if pred(main[i]["n"]):
                return i
        return None
    i_dec = first(lambda n: n.startswith("embed_action") or n.startswith("bcast_tanh"))
    i_loss = first(lambda n: n.startswith("loss_"))
    i_bwd = first(lambda n: not n.startswith("loss_") and not n.startswith("scale_kernel"), i_loss or 0) if i_loss is not None else None
    i_stem = first(lambda n: n.startswith("dtanh"), i_bwd or 0)
    i_vitb = None
    if i_stem is not None:
        # the ViT backward starts with the memset + cls-only final norm backward: first ln_bwd after the stem's last dtanh
        last_dtanh = max(i for i, e in enumerate(main) if e["n"].startswith("dtanh"))
        i_vitb = first(lambda n: n.startswith("ln_bwd"), last_dtanh)
    i_opt = first(lambda n: n.startswith("sumsq") or n.startswith("grad_norm"), i_vitb or 0)
    cuts = [("vit_forward(+stem)", 0), ("decoder_forward", i_dec), ("loss", i_loss), ("decoder_backward", i_bwd), ("stem_backward", i_stem), ("vit_backward", i_vitb), ("optimizer", i_opt)]
    cuts = [(n, i) for n, i in cuts if i is not None]
    print("\n| phase (main queue) | launches | wall ms | kernel ms |")
    print("|---|---|---|---|")
    per_phase = []
    for k, (name, i) in enumerate(cuts):
        j = cuts[k + 1][1] if k + 1 < len(cuts) else len(main)
        seg = main[i:j]
        if not seg:
            continue
        end = main[j]["s"] if j < len(main) else total
        print(f"| {name} | {len(seg)} | {(end - seg[0]['s']) / 1e3:.3f} | {sum(e['d'] for e in seg) / 1e3:.3f} |")
        per_phase.append((name, seg))
    for q in qs:
        if q != mainq:
            seg = [e for e in ev if e["q"] == q]
            print(f"| queue {q} | {len(seg)} | {(seg[-1]['s'] + seg[-1]['d'] - seg[0]['s']) / 1e3:.3f} (from {seg[0]['s'] / 1e3:.3f}) | {sum(e['d'] for e in seg) / 1e3:.3f} |")
    for name, seg in per_phase:
        agg = collections.defaultdict(list)
        for e in seg:
            agg[(e["n"], e["g"])].append(e["d"])
        print(f"\n## {name}: kernel time by (kernel, workgroups)\n")
        print("| ms | calls | avg us | kernel | workgroups |\n|---|---|---|---|---|")
        for (n, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:24]:
            print(f"| {sum(v) / 1e3:.3f} | {len(v)} | {sum(v) / len(v):.1f} | `{n}` | {g} |")
    if show_all:
        print("\n## every launch: start_us dur_us gap_us queue workgroups kernel")
        prev_end = {}
        for e in ev:
            gap = e["s"] - prev_end.get(e["q"], e["s"])
            prev_end[e["q"]] = e["s"] + e["d"]
            print(f"{e['s']:10.1f} {e['d']:8.1f} {gap:6.1f} {e['q']:>3} {e['g']:6d} {e['n']}")


if __name__ == "__main__":
    main()
