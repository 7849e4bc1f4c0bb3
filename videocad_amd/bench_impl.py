"""Implementation of bench.py (kept in the package so the repo-root script stays a thin CLI).

One step = `BaseTrainer.train_step` of the SHIPPED trainer (videocad_amd/trainer.py — the same object `_process_batch` calls):
forward -> fused loss -> backward (+ bucketed RCCL all-reduce under it when world > 1) -> clip(1.0) -> Adam, on a synthetic
loader-shaped batch already resident in HBM.  `python bench.py --gpus N` starts its N ranks itself (`torch.multiprocessing.spawn`,
what reference main.py:198 does) unless a launcher (torch.distributed.run) already set WORLD_SIZE.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import socket
import sys
import time

import numpy as np
import torch

from . import lib as L
from . import synth
from .engine import NativeEngine, make_config
from .model_factory import ModelFactory
from .trainer import create_trainer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANONICAL = dict(hidden_size=1024, nhead=4, num_decoder_layers=8, dim_feedforward=1024, window_size=10, act_dim=7,
                 num_classes=5, num_params=6, num_params_values=1000, max_ep_len=1000)
# transformer_experiments.json -> cad_past_10_actions_and_states_timestep_embedding: the JSON entry as the factory gets it
CANONICAL_MODEL_CONFIG = {"model_name": "autoregressive", "state_dim": 1644, "act_dim": 7, "hidden_size": 1024, "max_length": None,
                          "num_classes": 5, "encoder": "vit", "enable_past_actions": True, "nhead": 4, "num_decoder_layers": 8,
                          "dim_feedforward": 1024, "normalize": True, "num_views": 0, "window_size": 10,
                          "enable_timestep_embedding": True, "enable_past_states": True}
CLASS_WEIGHTS = os.path.join(ROOT, "tests", "golden", "class_weights.json")     # verbatim copy of the reference's data file

TRAIN_GF_PER_FRAME = {8: 6.25, 64: 5.70, 128: 5.68, 186: 5.68}     # fwd+bwd algorithmic GFLOP per frame
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3, "bf16x3": 2500.0 / 3}       # MI355X_MICROARCH.md dense MFMA peaks (bf16x3: three bf16 MFMAs per product)
CATS = ["gemm_fwd", "gemm_dgrad", "gemm_wgrad", "attention", "layernorm", "loss", "optimizer", "other"]


def train_gf_per_frame(T: int) -> float:
    if T in TRAIN_GF_PER_FRAME:
        return TRAIN_GF_PER_FRAME[T]
    # decoder attention grows with T; everything else is per frame.  Interpolate the survey's figures.
    ks = sorted(TRAIN_GF_PER_FRAME)
    lo = max([k for k in ks if k <= T], default=ks[0]); hi = min([k for k in ks if k >= T], default=ks[-1])
    if lo == hi:
        return TRAIN_GF_PER_FRAME[lo]
    w = (T - lo) / (hi - lo)
    return TRAIN_GF_PER_FRAME[lo] * (1 - w) + TRAIN_GF_PER_FRAME[hi] * w


def init_weights(eng: NativeEngine, device):
    """Random-init weights of the canonical architecture (deterministic integer hash; no checkpoints offline)."""
    for name, (off, numel, shape) in eng.table.items():
        eng.view(name).copy_(synth.make_param_torch(name, shape, device))
    eng.sync_shadow()


def synthetic_batch(B, T, seed, device, uint8=False):
    """Loader-shaped batch dict on `device`: frames ~ U[-1,1) fp32 — or uint8 pixels for the uint8 input path."""
    g = torch.Generator(device=device); g.manual_seed(seed)
    S = T + 1
    if uint8:
        frames = torch.randint(0, 256, (B, S, 1, 224, 224), device=device, generator=g, dtype=torch.uint8)
        cad = torch.randint(0, 256, (B, 1, 224, 224), device=device, generator=g, dtype=torch.uint8)
    else:
        frames = torch.rand(B, S, 1, 224, 224, device=device, generator=g) * 2 - 1        # Normalize(0.5, 0.5) range
        cad = torch.rand(B, 1, 224, 224, device=device, generator=g) * 2 - 1
    actions = torch.from_numpy(synth.make_actions(B, S, seed)).to(device)
    return {"frames": frames, "actions": actions, "cad_image": cad, "timesteps": torch.arange(S, device=device).repeat(B, 1)}


def build_trainer(dtype: str, dropout: float, device, rank: int, extra_tcfg=None):
    """ModelFactory.create_model -> create_trainer, exactly the objects experiment.py builds (reference experiment.py:69-119)."""
    cfg = dict(CANONICAL_MODEL_CONFIG, compute_dtype=dtype, dropout=dropout)
    model, mtype = ModelFactory().create_model(cfg["model_name"], cfg, device)
    init_weights(model._engine, device)
    model.mark_shadow_fresh()
    pk = {"loader": [], "sampler": None}
    tcfg = {"lr": 1e-5, "use_mse": True, "experiment_name": "bench", "class_weights_path": CLASS_WEIGHTS,
            "checkpoint_dir": os.path.join(os.environ.get("TMPDIR", "/tmp"), "vcad_bench_ckpt")}
    tcfg.update(extra_tcfg or {})
    cwd = os.getcwd()
    os.chdir(os.environ.get("TMPDIR", "/tmp"))           # the trainer creates logs/ under the CWD like the reference; keep the repo clean
    try:
        tr = create_trainer(pk, pk, pk, model, tcfg, device, mtype, rank=rank)
    finally:
        os.chdir(cwd)
    model.train()                                         # reference train loop: dropout 0.1 active
    return model, tr


def _cpu_baseline_worker(q, threads, seconds_budget):
    sys.path.insert(0, ROOT)
    from oracle import restatement as O
    torch.set_num_threads(threads)
    shapes = O.param_shapes()
    weights = {k: synth.make_param_torch(k, s, "cpu").numpy() for k, s in shapes.items()}
    ot = O.OracleTrainer(weights)
    res = {}
    for (B, T), budget, cap in (((2, 8), seconds_budget * 0.4, 12), ((2, 64), seconds_budget * 0.6, 4)):
        batch = synth.make_batch(B, T, seed=1)
        ot.step(batch)                                   # warm-up
        n, t0 = 0, time.time()
        while True:
            ot.step(batch); n += 1
            if time.time() - t0 > budget or n >= cap:
                break
        res[(B, T)] = (B * T * n / (time.time() - t0), n)
    v8, n8 = res[(2, 8)]; v64, n64 = res[(2, 64)]
    q.put({"value": round(v64, 2), "unit": "frames/s", "cores": threads, "kind": "port",
           "value_B2_T8": round(v8, 2), "host_cores": os.cpu_count(),
           "sample": f"{n64} full train steps of the oracle restatement at B=2,T=64 (value) and {n8} at B=2,T=8 (value_B2_T8); fp32, "
                     f"torch {torch.__version__} CPU, torch.set_num_threads({threads}) on a {os.cpu_count()}-core host "
                     f"(thread count from the measured sweep in profiles/r03_cpu_thread_sweep.txt: more threads make one step of this model slower)"})


def cpu_baseline(seconds_budget=24.0, hard_limit=240.0):
    """The oracle restatement (fp32 PyTorch-CPU, validated against the imported reference) timed on this host's cores on a
    bounded sample of the same workload, in a child process with a hard wall-clock limit.  Reported baseline only."""
    import multiprocessing as mp
    threads = min(os.cpu_count() or 1, 16)        # the thread count the sweep picks on the GPU box's 256-core host (profiles/r03_cpu_thread_sweep.txt: 8 / 16 / 32 / 64 threads -> 108 / 157 / 103 / 54 frames/s)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_cpu_baseline_worker, args=(q, threads, seconds_budget))
    pr.start()
    try:
        out = q.get(timeout=hard_limit)
    except Exception:
        out = {"value": None, "unit": "frames/s", "cores": threads, "kind": "port", "sample": f"did not finish within {hard_limit:.0f} s"}
    pr.join(timeout=5)
    if pr.is_alive():
        pr.terminate()
    return out


def logit_parity(model, device):
    """action-logit MAE / norm-wise error / arg-max agreement of this build's forward against the committed fp32 goldens
    (tests/golden/c1_full.npz: logits of the imported reference on the B=2, T=8 hash-generated batch with the same hash-init weights)."""
    try:
        gold = np.load(os.path.join(ROOT, "tests", "golden", "c1_full.npz"))
        b = synth.make_batch_torch(2, 8, 1, device)
        was = model.training
        model.eval()
        with torch.no_grad():
            an = b["actions"][:, :-1].clone(); an[:, :, 0] /= 4.0; an[:, :, 1:] /= 1000.0
            cmds, pars = model({"frames": b["frames"][:, :-1], "actions": an, "cad_image": b["cad_image"]})
        model.train(was)
        gp = torch.from_numpy(gold["params"]).to(device); gc = torch.from_numpy(gold["cmds"]).to(device)
        return {"vs": "fp32 goldens of the imported reference (tests/golden/c1_full.npz, B=2 T=8)",
                "logit_mae": float(((pars - gp).abs().sum() + (cmds - gc).abs().sum()) / (gp.numel() + gc.numel())),
                "rel_err": float((pars - gp).double().norm() / gp.double().norm()),
                "argmax_agreement": float((pars.argmax(-1).cpu().numpy() == gold["params_argmax"]).mean()),
                "cmd_argmax_agreement": float((cmds.argmax(-1).cpu().numpy() == gold["cmds_argmax"]).mean())}
    except Exception as ex:                                   # never let the reporting leg break the measurement
        return {"error": repr(ex)}


def _timed(tr, bd, steps, world, device, per_step=None):
    """K steps between barrier + synchronize on both sides (wall clock = the reported time); `per_step` (a list) additionally receives
    the K step durations in ms from events recorded on the compute stream inside the same region (no extra synchronisation)."""
    import torch.distributed as dist
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if per_step is not None else None
    t0 = time.perf_counter()
    for i in range(steps):
        if evs:
            evs[i].record()
        loss, met = tr.train_step(bd)
    if evs:
        evs[steps].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if evs:
        per_step.extend(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    return elapsed, loss


def short_leg(dtype, fp8, B, T, steps, warmup, device, rank, world, seed, dropout, want_parity=True):
    """One short measurement of another compute mode / shape with its own trainer (same model, same hash-init weights): reported BESIDE the
    headline line, never as `value`."""
    model, tr = build_trainer(dtype, dropout, device, rank)
    if fp8:
        model._engine.set_fp8(True)
    parity = logit_parity(model, device) if (want_parity and rank == 0) else None
    bd = synthetic_batch(B, T, seed + rank, device)
    for _ in range(warmup):
        tr.train_step(bd)
    per = []
    e, loss = _timed(tr, bd, steps, world, device, per)
    out = {"workload": f"seq_len={T} batch={B} per GPU", "dtype": dtype + ("+fp8 forward GEMMs (ViT)" if fp8 else ""),
           "value": round(world * B * T / (e / steps), 1), "unit": "frames/s", "ms_per_step": round(e / steps * 1e3, 3),
           "ms_per_step_median": round(float(np.median(per)), 3), "steps": steps, "warmup": warmup, "loss": float(loss.item())}
    if parity is not None:
        out["parity"] = parity
    if dtype == "f16" and rank == 0 and world == 1:          # the in-tolerance leg gets its own roofline block (same fields as the headline's, no PMC traffic)
        try:
            out["roofline"] = profile_step(tr, bd, dtype, B, T, B * T / (e / steps))[1]
        except Exception as ex:
            out["roofline"] = {"error": repr(ex)}
    del tr, model, bd
    torch.cuda.empty_cache()
    return out


def pmc_traffic(dom, d, dtype, B, T):
    """HBM bytes per launch of the dominant kernel.  PMC counters cannot be read inside this process; they come from the newest committed
    rocprofv3 --pmc passes of this same command (tools/collect_profiles.sh -> profiles/<round>_pmc.json; bf16 C2 workload only).  The file is
    quoted only while it still describes THIS library: same dominant family, same launches per step and the same algorithmic bytes per launch as
    the live profiled step (a changed dispatcher / tile policy makes it stale) — otherwise `traffic` is null and the source line says why."""
    try:
        cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc.json"))
        if not cands or dtype != "bf16" or (B, T) != (32, 64):
            return None, None
        name = cands[-1]
        pj = json.load(open(os.path.join(ROOT, "profiles", name)))
        src = f"profiles/{name} (collected at commit {pj.get('commit', 'unknown')} on {json.dumps(pj.get('box', 'unknown box'))}, {pj.get('collected_ms_per_step', '?')} ms/step under the tracer)"
        if pj.get("dominant_kernel") != dom or "dominant_hbm_bytes_per_launch" not in pj:
            return None, f"stale: {src} — dominant kernel there is {pj.get('dominant_kernel')}, here {dom}"
        if int(round(pj.get("dominant_launches_per_step", -1))) != int(d["launches"]):
            return None, f"stale: {src} — {pj.get('dominant_launches_per_step')} launches per step there, {d['launches']} in this library"
        alg = pj.get("dominant_alg_bytes_per_launch")
        live = d["bytes"] / max(d["launches"], 1)
        if alg and abs(alg - live) > 0.01 * live:
            return None, f"stale: {src} — algorithmic bytes per launch {alg} there, {round(live)} in this library"
        return round(pj["dominant_hbm_bytes_per_launch"]), src
    except Exception as ex:
        return None, f"unreadable: {ex!r}"


def profile_step(tr, bd, dtype, B, T, fps_per_gpu, headline=False):
    """ONE extra train step under the library's HIP-event profiler (events on the launch stream around every kernel family, single stream so the
    times do not overlap) -> (per-category breakdown, roofline block of the dominant kernel family)."""
    eng = tr.engine
    lib = eng.lib
    lib.vcad_profile_begin()
    tr.train_step(bd)
    torch.cuda.synchronize()
    pms, pfl, pby, pln = (C.c_double * 8)(), (C.c_double * 8)(), (C.c_double * 8)(), (C.c_int * 8)()
    lib.vcad_profile_end(C.byref(pms), C.byref(pfl), C.byref(pby), C.byref(pln))
    breakdown = {CATS[i]: {"ms": round(pms[i], 3), "launches": pln[i],
                           "tflops": round(pfl[i] / (pms[i] * 1e-3) / 1e12, 1) if pms[i] > 0 and pfl[i] > 0 else None,
                           "GBps": round(pby[i] / (pms[i] * 1e-3) / 1e9, 1) if pms[i] > 0 and pby[i] > 0 else None}
                 for i in range(8)}
    gemm_ms = pms[0] + pms[1] + pms[2]; gemm_fl = pfl[0] + pfl[1] + pfl[2]; gemm_n = pln[0] + pln[1] + pln[2]
    peak = PEAK_TFLOPS[dtype]
    fam = {}
    for tag, name in ((1, "gemm_dma_kernel"), (2, "gemm_kernel"), (3, "gemm_mid_kernel"), (4, "gemm_grouped_kernel")):
        o4 = (C.c_double * 4)(); lib.vcad_profile_kernel(tag, C.byref(o4))
        fam[name] = {"ms": o4[0], "flops": o4[1], "bytes": o4[2], "launches": int(o4[3])}
    # the dominant kernel: the persistent DMA-fed GEMM (every large ViT Linear, forward / dgrad / wgrad) — the family with the most time
    dom = max(fam, key=lambda k: fam[k]["ms"]) if any(v["ms"] > 0 for v in fam.values()) else "gemm_dma_kernel"
    d = fam[dom]
    ach = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
    ach_all = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    traffic, traffic_src = pmc_traffic(dom, d, dtype, B, T) if headline else (None, None)
    roof = {"bound": "mfma", "kernel": f"{dom} (dominant kernel family: {d['launches']} launches, {d['ms']:.2f} ms of the step; split-K launches include their slab reduction)",
            "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
            "alg_bytes_per_launch": round(d["bytes"] / max(d["launches"], 1)),
            "launches_per_step": d["launches"], "avg_launch_us": round(d["ms"] * 1e3 / max(d["launches"], 1), 1),
            "alg_tflop_per_step": round(d["flops"] / 1e12, 3),
            "all_linear_launches": {"achieved": round(ach_all, 1), "frac": round(ach_all / peak, 4), "launches_per_step": gemm_n, "ms": round(gemm_ms, 3),
                                    "alg_tflop_per_step": round(gemm_fl / 1e12, 3),
                                    "by_kernel": {k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                                                      "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 else None} for k, v in fam.items()}},
            "step_level_frac": round(fps_per_gpu * train_gf_per_frame(T) * 1e9 / (peak * 1e12), 4)}
    if not headline:      # legs: the compact form (dominant kernel + fraction); the headline block carries the details
        roof = {k: roof[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launches_per_step", "avg_launch_us", "alg_tflop_per_step", "step_level_frac")}
        roof["all_linear_frac"] = round(ach_all / peak, 4)
        roof["kernel_breakdown_ms"] = {k: v["ms"] for k, v in breakdown.items()}
    return breakdown, roof


def run(args):
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if torch.cuda.device_count() < (local + 1):
        raise SystemExit(f"bench.py: rank {rank} needs cuda:{local} but only {torch.cuda.device_count()} device(s) are visible")
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)            # "nccl" IS RCCL on ROCm (reference main.py:31-35)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    B, T = args.batch, args.seq
    # data-parallel exchange knobs (GradSync; defaults = the reference's fp32 all-reduce): --grad-wire half / --grad-exchange rs_ag|auto
    model, tr = build_trainer(args.dtype, args.dropout, device, rank,
                              {"grad_wire": getattr(args, "grad_wire", "fp32"), "grad_exchange": getattr(args, "grad_exchange", "all_reduce")})
    eng = model._engine
    # A/B switches expressible as per-engine kernel-selection flags (the product default is 0 = automatic); the process-global
    # selectors of the A/B build are set by tools/bench_ab.py before it calls launch()
    sel = lambda v, never, always: never if v == 0 else (always if v == 1 else 0)
    eng.set_gemm_flags(sel(getattr(args, "gemm_dma", -1), L.GEMM_DMA_NEVER, L.GEMM_DMA_ALWAYS) | sel(getattr(args, "gemm_wide", -1), L.GEMM_WIDE_NEVER, L.GEMM_WIDE_ALWAYS)
                       | sel(getattr(args, "gemm_mid", -1), L.GEMM_MID_NEVER, L.GEMM_MID_ALWAYS))
    if getattr(args, "no_side", 0):
        eng.set_side_stream(False)
    if getattr(args, "fp8", False):
        eng.set_fp8(True)            # VCAD_FP8 forward mode: ViT Linears on the block-scaled fp8 matrix cores (BASELINE configs[4] variant)
    bd = synthetic_batch(B, T, 1000 * 2 + rank, device, uint8=args.uint8_frames)
    # logit parity of this build against the committed fp32 goldens — BEFORE any optimiser step (the goldens are for the hash-init weights)
    parity = logit_parity(model, device) if (rank == 0 and not getattr(args, "no_parity", False)) else None

    for _ in range(args.warmup):
        tr.train_step(bd)
    per_step_ms = []
    elapsed, loss = _timed(tr, bd, args.steps, world, device, per_step_ms)
    ms = elapsed / args.steps * 1e3
    fps = world * B * T / (elapsed / args.steps)

    comm = None
    if world > 1:        # exposed communication: the same steps with the all-reduces skipped (diagnostic, outside the timed region)
        tr.gradsync.skip_comm = True
        e_nc, _ = _timed(tr, bd, max(3, args.steps // 2), world, device)
        tr.gradsync.skip_comm = False
        ms_nc = e_nc / max(3, args.steps // 2) * 1e3
        comm = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "buckets": len(eng.buckets),
                "bucket_MB": [round((hi - lo) * 4 / 1e6, 1) for lo, hi in eng.buckets],
                "grad_wire": tr.gradsync.wire, "grad_exchange": tr.gradsync.exchange,
                "ms_per_step_without_allreduce": round(ms_nc, 3), "exposed_comm_ms": round(ms - ms_nc, 3)}
        # one more step with issue / completion events around every collective (ms since the backward started): where each exchange sat and the
        # GB/s it achieved (wire bytes / duration; busGBps = x 2 (W - 1) / W, the per-link figure to hold against xGMI)
        tr.gradsync.timing = True
        tr.train_step(bd)
        comm["last_step"] = tr.gradsync.comm_report()
        tr.gradsync.timing = False

    # ---- one extra, profiled step (outside the timed region): HIP events on the launch stream around every kernel family
    breakdown, roof = profile_step(tr, bd, args.dtype, B, T, fps / world, headline=True)

    # ---- BASELINE configs[3]'s per-GPU shape (seq_len 186 — the maximum horizon — at 16 clips per GPU) and configs[2] (seq_len 128, batch 64 on
    # one GPU), a few steps each, reported beside the headline configuration (north_star asks for both horizons; BASELINE.md §3 lists
    # configs[2]); same trainer, same weights, workspace re-planned
    extra, extra_c3 = None, None
    if (B, T) == (32, 64) and not getattr(args, "no_seq186", False):
        for (B2, T2, K2, W2, name) in ((16, 186, 10, 3, "c4"), (64, 128, 4, 2, "c3")):
            if name == "c3" and world > 1:
                continue
            bd2 = synthetic_batch(B2, T2, 3000 + T2 + rank, device, uint8=args.uint8_frames)
            for _ in range(W2):
                tr.train_step(bd2)
            per2 = []
            e2, _ = _timed(tr, bd2, K2, world, device, per2)
            leg = {"workload": f"seq_len={T2} batch={B2} per GPU (BASELINE configs[{3 if name == 'c4' else 2}]" + (" per-GPU shape)" if name == "c4" else ")"),
                   "value": round(world * B2 * T2 / (e2 / K2), 1), "unit": "frames/s", "ms_per_step": round(e2 / K2 * 1e3, 3),
                   "ms_per_step_median": round(float(np.median(per2)), 3), "steps": K2, "warmup": W2}
            if rank == 0 and world == 1:
                leg["roofline"] = profile_step(tr, bd2, args.dtype, B2, T2, B2 * T2 / (e2 / K2))[1]
            if name == "c4":
                extra = leg
            else:
                extra_c3 = leg
            del bd2
        tr.train_step(bd)                       # back to the headline plan (keeps later legs on the C2 workspace)

    # ---- the other compute modes at the headline shape, short legs with their own parity blocks: bf16x3 = the IN-TOLERANCE mode (logits within
    # 1e-3 of the fp32 reference, arg-max exact), f32 = exact-fp32 MFMA.  (r05: the MXFP8 forward probe is no longer a bench leg — it lost to bf16 by
    # 12-13 % at configs[4]'s per-GPU shape in every r04 run, DESIGN.md §4.3 says why and what a winning fp8 path needs; `--fp8` still runs it on request)
    modes = None
    if rank == 0 and world == 1 and (B, T) == (32, 64) and args.dtype == "bf16" and not getattr(args, "no_modes", False):
        modes = {}
        # f16 = the fp16-storage build of the library (libvcad_hip_f16.so, VCAD_F16): the bf16 mode's kernels with ten mantissa bits — in tolerance at its speed
        for key, dt, f8, K3, B3, T3 in (("f16", "f16", False, 10, B, T), ("f16_seq_len_186", "f16", False, 6, 16, 186),
                                        ("bf16x3", "bf16x3", False, 4, B, T), ("bf16x3_seq_len_186", "bf16x3", False, 3, 16, 186), ("f32", "f32", False, 2, B, T)):
            try:
                modes[key] = short_leg(dt, f8, B3, T3, K3, 3 if dt == "f16" else 1, device, rank, world, 2000, args.dropout, want_parity=not key.endswith("_seq_len_186"))
            except Exception as ex:               # a reporting leg never breaks the headline measurement
                modes[key] = {"error": repr(ex)}

    # ---- input path: the same step fed from pinned host memory through the double-buffered stager (PCIe-inclusive; never `value`)
    pcie = None
    if rank == 0 and world == 1 and not getattr(args, "no_pcie", False):
        try:
            pcie = pcie_inclusive(tr, B, T, device)
        except Exception as ex:
            pcie = {"error": repr(ex)}

    if rank == 0:
        out = {"metric": "training frames/sec (224x224 grayscale frames, canonical AutoRegressiveTransformer)",
               "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 3), "ms_per_step_median": round(float(np.median(per_step_ms)), 3) if per_step_ms else None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.dtype + ("+fp8 forward GEMMs (ViT)" if getattr(args, "fp8", False) else ""), "data": "synthetic (" + ("uint8 pixels" if args.uint8_frames else "U[-1,1) fp32 frames") + " in HBM, hash-init weights)",
               "config": {"workload": f"autoregressive_transformer {args.dtype} seq_len={T} batch={B} per GPU, {world}xMI355X (BASELINE configs[1])"
                          if (T, B) == (64, 32) else f"canonical model seq_len={T} batch={B} per GPU",
                          "dropout": args.dropout, "clips_per_gpu": B, "seq_len": T, "global_batch": B * world, "parallelism": f"dp{world}",
                          "step": "videocad_amd.trainer.BaseTrainer.train_step", "loss": float(loss.item())},
               "parity": parity, "roofline": roof, "kernel_breakdown": breakdown}
        if comm:
            out["comm"] = comm
        if extra:
            out["seq_len_186"] = extra
        if extra_c3:
            out["seq_len_128_batch_64"] = extra_c3
        if modes:
            out["modes"] = modes
            # the number to quote next to "matches the reference" (north_star: logits within 1e-3 relative, arg-max exact): the fastest measured mode whose
            # own parity block (fp32 goldens of the imported reference) says so
            ok = lambda m: "value" in m and (m.get("parity") or {}).get("rel_err", 1.0) < 1e-3 and (m.get("parity") or {}).get("argmax_agreement") == 1.0 \
                and (m.get("parity") or {}).get("cmd_argmax_agreement") == 1.0
            cands = {k: modes[k] for k in ("f16", "bf16x3", "f32") if ok(modes.get(k) or {})}
            if cands:
                best = max(cands, key=lambda k: cands[k]["value"]); m = cands[best]
                out["in_tolerance"] = {"dtype": best, "workload": m["workload"], "value": m["value"], "unit": "frames/s", "ms_per_step": m["ms_per_step"],
                                       "rel_err": m["parity"]["rel_err"], "argmax_agreement": m["parity"]["argmax_agreement"],
                                       "cmd_argmax_agreement": m["parity"]["cmd_argmax_agreement"],
                                       "seq_len_186": {k: (modes.get(best + "_seq_len_186") or {}).get(k) for k in ("workload", "value", "ms_per_step")},
                                       "candidates": {k: {"value": v["value"], "rel_err": v["parity"]["rel_err"]} for k, v in cands.items()}}
        if pcie:
            out["pcie_inclusive"] = pcie
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as ex:          # the oracle is test infrastructure; never let it break the GPU number
                out["cpu_baseline"] = {"value": None, "error": repr(ex)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pcie_inclusive(tr, B, T, device, steps=6):
    """frames/s when every batch starts in PINNED HOST memory (the loader's hand-over point) and goes through data.DeviceStager:
    fp32 frames (the reference's contract, 4 B/pixel) vs uint8 pixels normalised in the patchify kernel (1 B/pixel)."""
    from .data import DeviceStager
    res = {}
    for name, u8 in (("fp32_frames", False), ("uint8_frames", True)):
        host = synthetic_batch(B, T, 77, "cpu", uint8=u8)
        host = {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in host.items()}
        loader = [host] * (steps + 2)
        it = iter(DeviceStager(loader, device))
        for _ in range(2):
            tr.train_step(tr.prepare_batch(next(it)))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 0
        for bd in it:
            tr.train_step(tr.prepare_batch(bd)); n += 1
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / max(n, 1)
        res[name] = {"frames_per_s": round(B * T / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                     "h2d_MB_per_step": round(sum(v.numel() * v.element_size() for v in host.values() if torch.is_tensor(v)) / 1e6, 1)}
    res["note"] = "host->HBM copy of batch i+1 overlapped with step i on a private stream (videocad_amd/data.py); never the headline value"
    return res


# ------------------------------------------------------------------------------------------------ self-launch
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn_entry(local_rank, args, port):
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(args.gpus),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    run(args)


def launch(args):
    """`--gpus N` without a launcher: one process per GPU on this node (reference main.py:198 `mp.spawn`)."""
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        n = torch.cuda.device_count()
        if n < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {n} device(s) are visible")
        import torch.multiprocessing as mp
        mp.spawn(_spawn_entry, args=(args, _free_port()), nprocs=args.gpus, join=True)
    else:
        run(args)
