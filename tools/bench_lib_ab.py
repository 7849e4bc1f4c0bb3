"""Interleaved A/B of two BUILDS of the product library on one box (box-to-box spread is +-2-3 %, a 1 % change only shows inside one call).

    python tools/bench_lib_ab.py <libA.so> <libB.so> [<libC.so> ...] [rounds] [steps] [extra bench.py flags ...]

Runs `bench.py --profile-only` (the headline train steps only) in a fresh process per leg, A B A B ..., with `videocad_amd.lib.LIB_PATH` pointed at the
given build (tools only: the product has no library override), and prints ms per step of every leg plus the per-build medians.  A library path of
"-" means the in-tree build.  Typical use: an older build kept under tools/_bin/ (git worktree of the previous commit, `make libvcad_hip.so`) against HEAD."""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def leg(lib, steps, extra):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from videocad_amd import lib as L\n"
            "p = %r\n"
            "if p != '-':\n"
            "    L.LIB_PATH = p\n"
            "    import torch, ctypes; raw = ctypes.CDLL(p)           # an older build lacks entry points added since: declare what it has\n"
            "    L.PROTOTYPES = {k: v for k, v in L.PROTOTYPES.items() if hasattr(raw, k)}\n"
            "import bench\n"
            "bench.main(['--profile-only', '--steps', %r, '--warmup', '5'] + %r)\n") % (ROOT, lib, str(steps), list(extra))
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    for line in reversed(out.stdout.strip().splitlines()):
        if line.startswith("{"):
            j = json.loads(line)
            return j["ms_per_step"], j.get("ms_per_step_median")
    raise RuntimeError(f"no bench line from {lib}: {out.stderr[-800:]}")


def main():
    args = sys.argv[1:]
    libs = []
    while args and not args[0].isdigit() and not args[0].startswith("--"):
        libs.append(args.pop(0))
    rounds = int(args.pop(0)) if args and args[0].isdigit() else 3
    steps = int(args.pop(0)) if args and args[0].isdigit() else 20
    extra = args
    res = {lib: [] for lib in libs}
    for r in range(rounds):
        for i, lib in enumerate(libs):
            ms, med = leg(lib, steps, extra)
            res[lib].append(ms)
            print(f"round {r} {chr(65 + i)} {os.path.basename(lib) if lib != '-' else 'in-tree'}: {ms:.3f} ms/step (median of steps {med})", flush=True)
    meds = [statistics.median(res[lib]) for lib in libs]
    for i, lib in enumerate(libs):
        print(f"{chr(65 + i)} {os.path.basename(lib) if lib != '-' else 'in-tree'}: median {meds[i]:.3f} ms" + (f"  ({meds[i] - meds[0]:+.3f} ms, {(meds[i] / meds[0] - 1) * 100:+.2f} % vs A)" if i else ""))


if __name__ == "__main__":
    main()
