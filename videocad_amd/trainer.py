"""Trainer with the reference's surface (reference trainer.py:193-1385): `create_trainer(...)`, `.train(epochs)`,
`._process_batch(batch, noise=False) -> (loss, metrics)`, `.compute_loss(action_preds, actions)`, `.evaluate(model, mode)`,
`.sample(...)`, `.find_first_mistake(...)`, `.save_checkpoint(...)`.

The hot loop (`_process_batch`, reference :480-496) is ONE pass through the C ABI:
    vcad_forward -> vcad_loss (loss + ~45 metric counters + dlogits, no host sync) -> vcad_backward_stage x5
    (gradient all-reduce of each finished bucket over RCCL on a side stream, world_size > 1) -> vcad_optimizer_step
    (global-norm clip 1.0 + Adam, reference :493-494).
The epoch loop keeps the reference's contract — per-epoch validation (`val_frequency`), checkpoints every `save_frequency`
epochs in the reference's on-disk format, early stopping with the all-ranks-agree flag, metric JSON dumps under logs/ — on top
of that step; evaluation accumulates the fused counters on the device and (unlike the reference, quirk 6) reduces them over ranks.
"""
from __future__ import annotations

import csv
import datetime
import json
import os
import random
import time
from typing import Optional

import torch

from . import lib as L
from .data import DeviceStager
from .model_factory import ModelType

TOLERANCE = 3                                                   # reference trainer.py:20
PARAM_NAMES = ["Label", "x", "y", "Key Pressed", "Times Key Pressed", "Scroll Amount", "Typed Value"]   # :834


def metrics_from_counters(m) -> dict:
    """int32[32] counter block (csrc/loss.h) -> the metric dict of reference trainer.py:1039-1061."""
    m = [int(x) for x in m]
    cc, cn = m[L.MET_CMD_CORRECT:L.MET_CMD_CORRECT + 5], m[L.MET_CMD_COUNT:L.MET_CMD_COUNT + 5]
    pc, pn = m[L.MET_PAR_CORRECT:L.MET_PAR_CORRECT + 6], m[L.MET_PAR_COUNT:L.MET_PAR_COUNT + 6]
    d = {"correct_predictions": m[L.MET_CORRECT], "total_predictions": m[L.MET_TOTAL], "cmd_corrects": cc, "cmd_counts": cn,
         "param_corrects": pc, "param_counts": pn, "cmd_correct_topk": m[L.MET_CMD_CORRECT_TOPK], "cmd_counts_topk": m[L.MET_CMD_COUNT_TOPK],
         "param_correct_topk": m[L.MET_PAR_CORRECT_TOPK], "param_counts_topk": m[L.MET_PAR_COUNT_TOPK],
         "perfect_sequences": 0, "perfect_commands": 0, "total_sequences": 0, "perfect_sequence_accuracy": 0}
    for i in range(6):
        d[f"param_corrects_{i}"] = pc[i]; d[f"param_counts_{i}"] = pn[i]
    for i in range(5):
        d[f"cmd_corrects_{i}"] = cc[i]; d[f"cmd_counts_{i}"] = cn[i]
    return d


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def unwrap(model):
    """The native module behind the wrappers reference experiment.py:92-109 may put around it: `torch.compile` (OptimizedModule,
    `._orig_mod`) and `DistributedDataParallel` (`.module`), in either nesting.  The trainer drives the engine of the native module
    directly — the wrappers' own forward (dynamo tracing, DDP's reducer hooks) is never entered: the arithmetic is one C-ABI call and
    the gradient exchange is `GradSync`."""
    for _ in range(4):          # (look in _modules: OptimizedModule forwards unknown attributes to the module it wraps, so hasattr() cannot tell them apart)
        mods = getattr(model, "_modules", {})
        if "_orig_mod" in mods:
            model = mods["_orig_mod"]
        elif "module" in mods and type(model).__name__ in ("DistributedDataParallel", "DataParallel"):
            model = mods["module"]
        else:
            break
    return model


# ------------------------------------------------------------------------------------------------ logs / checkpoints
class MetricsHandler:
    """logs/<experiment>/<ext>.json on rank 0 (reference trainer.py:86-131)."""

    def __init__(self, experiment_name, rank=0):
        self.experiment_name, self.is_master = experiment_name, rank == 0
        self.dir = os.path.join("logs", experiment_name)
        if self.is_master:
            os.makedirs(self.dir, exist_ok=True)

    def save_metrics(self, metrics, ext=""):
        if not self.is_master:
            return
        os.makedirs(self.dir, exist_ok=True)
        name = ext if ext else datetime.datetime.now().strftime("%Y%m%d_%H%M%S")
        with open(os.path.join(self.dir, f"{name}.json"), "w") as f:
            json.dump(metrics, f, indent=4)

    def print_metrics(self, metrics, mode=""):
        if not self.is_master:
            return
        tot = metrics.get("total_predictions", 0)
        acc = 100.0 * metrics.get("correct_predictions", 0) / tot if tot else 0.0
        print(f"{mode}: CMD accuracy: {metrics.get('cmd_accuracy', 0):.2f}%, Params accuracy: {metrics.get('params_accuracy', 0):.2f}%, "
              f"Overall: {acc:.2f}%, Top-30 CMD accuracy: {metrics.get('cmd_accuracy_topk', 0):.2f}%, "
              f"Top-30 Params accuracy: {metrics.get('param_accuracy_topk', 0):.2f}%")
        for i in range(6):
            print(f"  Parameter {i}: {metrics.get(f'param_accuracy_{i}', 0):.2f}%")


class NativeAdam:
    """What `trainer.optimizer` is here: the handle on the fused clip + Adam kernel's state (flat m / v buffers + step count).
    `state_dict()` / `load_state_dict()` speak torch.optim.Adam's format (state indexed by position in `model.parameters()`,
    one param_group per learning rate), so `checkpoint['optimizer_state_dict']` (reference trainer.py:146-151) round-trips and
    also loads into a `torch.optim.Adam(model.parameters())` built on THIS model.  Indices are over this model's 309 live
    tensors; the reference's Adam indexes its own, longer parameter list (dead GPT-2 trunk first) — to exchange optimiser state
    with a checkpoint the reference wrote use `state_dict_for(names)` / `load_state_dict_from(sd, names)` (name-keyed)."""

    def __init__(self, model, lr, groups=None, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.engine = model, model._engine
        self.betas, self.eps = betas, eps
        self.names = [n for n, _ in model.named_parameters()]      # torch.optim.Adam(model.parameters()) indexes state in this order
        self.bucket_lr = None                  # one lr per gradient bucket when the reference's `frozen` groups are used
        if groups is None:
            self.param_groups = [self._group(lr, list(range(len(self.names))))]
        else:                                   # reference :237-251: cad ViT, state ViT, everything else
            idx = {"cad": [], "state": [], "rest": []}
            for i, n in enumerate(self.names):
                idx["cad" if n.startswith("cad_embedding_model.") else "state" if n.startswith("state_embedding_model.") else "rest"].append(i)
            self.param_groups = [self._group(groups["lr_cad"], idx["cad"]), self._group(groups["lr_state"], idx["state"]), self._group(lr, idx["rest"])]
            self.bucket_lr = [lr, lr, groups["lr_cad"], groups["lr_state"], groups["lr_state"]]   # buckets: heads + decoder, stem, CAD ViT, frame ViT x2

    def _group(self, lr, params):
        return {"lr": lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None,
                "capturable": False, "differentiable": False, "fused": None, "params": params}

    @property
    def lr(self):
        if self.bucket_lr is not None:
            g = self.param_groups
            return [g[2]["lr"], g[2]["lr"], g[0]["lr"], g[1]["lr"], g[1]["lr"]]
        return self.param_groups[0]["lr"]

    def zero_grad(self, set_to_none=True):
        pass                                    # the engine WRITES its gradient buffer on every backward

    def state_dict(self):
        eng, state = self.engine, {}
        if eng.step_count > 0:
            for i, n in enumerate(self.names):
                state[i] = {"step": torch.tensor(float(eng.step_count)), "exp_avg": eng.view(n, eng.m), "exp_avg_sq": eng.view(n, eng.v)}
        return {"state": state, "param_groups": [dict(g) for g in self.param_groups]}

    # ---- name-keyed forms: torch.optim.Adam indexes its state by position in the parameter list it was built on.  The reference builds
    # it on `model.parameters()` of ITS module tree (trainer.py:251-253: GPT-2 trunk first, 77.6 M parameters that never get state), so
    # the positional dict above does not line up with a checkpoint the reference wrote — these convert through parameter NAMES.
    def named_state(self):
        """{parameter name: {'step', 'exp_avg', 'exp_avg_sq'}} for every live tensor (empty before the first step)"""
        sd = self.state_dict()["state"]
        return {self.names[i]: st for i, st in sd.items()}

    @staticmethod
    def optimizer_order(named_parameters, frozen=False):
        """`param_names` for state_dict_for / load_state_dict_from: the names of a model's parameters IN THE ORDER torch.optim.Adam numbers them.
        torch numbers parameters group by group.  The reference builds ONE group over `model.parameters()` (trainer.py:251-253) — then the order is
        that of `named_parameters()` — except in its `frozen` mode (trainer.py:236-248), where the groups are [cad_embedding_model.*,
        state_embedding_model.*, everything else]: index 0 is then the CAD ViT's first tensor, not the model's.  `named_parameters`: an iterable of
        names or of (name, parameter) pairs, e.g. `reference_model.named_parameters()` or tests/golden/reference_param_order.json's list."""
        names = [n if isinstance(n, str) else n[0] for n in named_parameters]
        if not frozen:
            return names
        cad = [n for n in names if n.startswith("cad_embedding_model.")]
        state = [n for n in names if n.startswith("state_embedding_model.")]
        rest = [n for n in names if not n.startswith(("cad_embedding_model.", "state_embedding_model."))]
        return cad + state + rest

    def state_dict_for(self, param_names):
        """optimizer_state_dict as a `torch.optim.Adam` built over parameters named `param_names` would hold it.  `param_names` must be in the
        OPTIMIZER's index order — `NativeAdam.optimizer_order(reference_model.named_parameters(), frozen=...)`: plain `named_parameters()` order for the
        reference's default single group, group order for its `frozen` mode.  State only for names this model trains (the reference's dead parameters
        never receive gradients, so torch keeps no state for them either); one param_group per native group (one, or the three `frozen` ones), each
        listing its indices in ascending order."""
        named, pos = self.named_state(), {n: i for i, n in enumerate(param_names)}
        missing = [n for n in named if n not in pos]
        if missing:
            raise KeyError(f"state_dict_for: {len(missing)} trained parameters are not in param_names (e.g. {missing[:3]})")
        # one param_group per native group (the reference's `frozen` mode, trainer.py:236-248: CAD ViT, state ViT, everything else — each with
        # its own lr); names this model does not hold (the reference's dead GPT-2 trunk etc.) belong to the reference's last group ("neither ViT")
        mine = {self.names[i]: gi for gi, g in enumerate(self.param_groups) for i in g["params"]}
        groups = [self._group(g["lr"], []) for g in self.param_groups]
        for i, n in enumerate(param_names):
            groups[mine.get(n, len(groups) - 1)]["params"].append(i)
        return {"state": {pos[n]: st for n, st in named.items()}, "param_groups": groups}

    def load_state_dict_from(self, sd, param_names):
        """inverse of state_dict_for: accepts the optimizer_state_dict of an Adam built over `param_names` (a reference checkpoint; `param_names` in the
        optimizer's index order, see optimizer_order).  Every moment tensor's shape is checked against the native parameter it is mapped to — and a
        moment must never land on a parameter this model does not train: most wrong orders fail here instead of loading moments into the wrong
        tensors.  NOT detectable: the reference's two ViT towers are isomorphic, so plain named_parameters() order on a `frozen` checkpoint (state
        ViT and CAD ViT swapped) passes every shape check — build the list with optimizer_order(..., frozen=True).  Learning rates are mapped through parameter NAMES: each native group takes the lr of the checkpoint group that holds its
        parameters (so the reference's three `frozen` groups, in whatever order and with its extra dead parameters, land on the right buckets)."""
        idx = {n: i for i, n in enumerate(self.names)}
        conv = {}
        for i, st in sd.get("state", {}).items():
            if not 0 <= int(i) < len(param_names):
                raise IndexError(f"load_state_dict_from: optimizer state index {i} outside param_names ({len(param_names)} names)")
            n = param_names[int(i)]
            if n not in idx:
                # torch keeps optimiser state only for parameters that received gradients; the reference's parameters this model does not hold
                # (GPT-2 trunk, ...) never do — state on such a name means the indices were read through the wrong name list
                raise ValueError(f"load_state_dict_from: state {i} -> '{n}', which is not a trained parameter of this model — "
                                 "param_names is not in the optimizer's index order (NativeAdam.optimizer_order)")
            want = tuple(self.engine.table[n][2])
            for key in ("exp_avg", "exp_avg_sq"):
                if tuple(st[key].shape) != want:
                    raise ValueError(f"load_state_dict_from: state {i} -> '{n}': {key} has shape {tuple(st[key].shape)}, the parameter {want} — "
                                     "param_names is not in the optimizer's index order (NativeAdam.optimizer_order)")
            conv[idx[n]] = st
        self.load_state_dict({"state": conv, "param_groups": []})
        src_groups = sd.get("param_groups") or []
        if not src_groups:
            return
        lr_of = {}
        for g in src_groups:
            for i in g["params"]:
                if 0 <= int(i) < len(param_names):
                    lr_of[param_names[int(i)]] = g["lr"]
        for g in self.param_groups:
            lrs = {lr_of[self.names[i]] for i in g["params"] if self.names[i] in lr_of}
            if len(lrs) > 1:
                raise ValueError(f"load_state_dict_from: the checkpoint trains one native parameter group at {len(lrs)} learning rates {sorted(lrs)}; "
                                 "the fused Adam kernel keeps one lr per gradient bucket (CAD ViT / state ViT / rest)")
            if lrs:
                g["lr"] = lrs.pop()

    def load_state_dict(self, sd):
        eng = self.engine
        steps = set()
        with torch.no_grad():
            for i, st in sd.get("state", {}).items():
                n = self.names[int(i)]
                eng.view(n, eng.m).copy_(st["exp_avg"].to(eng.device, torch.float32))
                eng.view(n, eng.v).copy_(st["exp_avg_sq"].to(eng.device, torch.float32))
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("NativeAdam.load_state_dict: per-parameter step counts differ (the fused kernel keeps one)")
        eng.step_count = steps.pop() if steps else 0
        for g, src in zip(self.param_groups, sd.get("param_groups", [])):
            g["lr"] = src["lr"]


class CheckpointHandler:
    """checkpoints/<experiment>/epoch_N.pt | best_model.pt = {'epoch','model_state_dict','optimizer_state_dict','loss'}, rank 0 only
    (reference trainer.py:133-180; loaded by experiment.py:61-71 / test.py through ModelFactory.create_model(state_dict=...))."""

    def __init__(self, experiment_name, rank=0, dir_name="checkpoints"):
        self.is_master = rank == 0
        self.checkpoint_dir = os.path.join(dir_name, experiment_name)
        if self.is_master:
            os.makedirs(self.checkpoint_dir, exist_ok=True)

    def build_checkpoint(self, epoch, loss, model, optimizer):
        return {"epoch": epoch + 1, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(), "loss": loss}

    def save_checkpoint(self, epoch, loss, model, optimizer, is_best=False):
        if not self.is_master:
            return None
        ckpt = self.build_checkpoint(epoch, loss, model, optimizer)
        path = os.path.join(self.checkpoint_dir, "best_model.pt" if is_best else f"epoch_{epoch + 1}.pt")
        torch.save(ckpt, path)
        print(f"Saved {'best ' if is_best else ''}model checkpoint for epoch {epoch + 1}")
        return ckpt


class GradSync:
    """Data-parallel gradient exchange (replaces the DDP wrap at reference experiment.py:104-109).

    Construction does what the DDP constructor did for the reference: rank 0's parameters (and optimiser state) are broadcast,
    so every replica starts from the same weights whatever each process's RNG drew.
    Each backward stage finalises one contiguous bucket of the flat gradient buffer: heads + decoder first (360 MB, 71 % of the bytes —
    its exchange starts before the stem's backward is even launched), then the stem (16 MB), the CAD ViT (65 MB, computed on the engine's
    side stream), and the two halves of the frame ViT.  Every exchange is issued on a communication stream right away so it runs over
    xGMI underneath the next stage's kernels; the stem's bucket travels with the CAD ViT's (adjacent ranges: one collective).  The 1/world
    mean is folded into the Adam kernel.  Only live parameters travel (508 MB fp32 instead of the reference's 818 MB incl. dead GPT-2 zeros).

    r05 — two knobs for the first multi-GPU runs (training_config `grad_wire`, `grad_exchange`, `grad_rs_min_mb`; DESIGN.md §6 says which to try when):
      wire = "fp32" (default: the reference's numerics) | "half": the bucket travels in the engine's 16-bit storage format (bf16; fp16 with a
             power-of-two scale from the all-reduced max |g| for VCAD_F16 engines) — half the xGMI bytes, packed / unpacked by the library
             (include/vcad.h: vcad_wire_*).  Limitation of the fp16 wire: ONE max |g| — one power-of-two scale — covers a whole exchanged range
             (up to ~90 M elements for the heads + decoder bucket), so with world = 8 elements below ~max|g| * 2^-27 flush to zero or go subnormal before the
             sum; Adam normalises per element, so a tensor whose gradients are that small next to a large-gradient neighbour can lose its update signal.
             The bf16 wire (bf16 engines) has fp32's range and no such floor; engines with a wide gradient range should take bf16 or stay on "fp32";
      exchange = "all_reduce" (default) | "rs_ag": reduce_scatter + all_gather of the same bucket — the two halves of a ring all-reduce as separate
             collectives, each a one-hop pattern on the fully connected xGMI mesh | "auto": rs_ag for buckets of at least `rs_min_mb` MB on the wire.
    `timing = True` records issue / completion events around every exchange; `comm_report()` turns the last step's into milliseconds
    relative to the start of the backward plus achieved GB/s (the first N-GPU run is then diagnosable from one line)."""

    def __init__(self, engine, group=None, on_params_changed=None, force_staged=False, wire="fp32", exchange="all_reduce", rs_min_mb=32.0):
        import torch.distributed as dist
        if wire not in ("fp32", "half") or exchange not in ("all_reduce", "rs_ag", "auto"):
            raise ValueError(f"GradSync: grad_wire must be 'fp32' or 'half' (got {wire!r}), grad_exchange 'all_reduce', 'rs_ag' or 'auto' (got {exchange!r})")
        self.eng, self.dist, self.group = engine, dist, group
        self.wire, self.exchange, self.rs_min_mb = wire, exchange, float(rs_min_mb)
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.rank = dist.get_rank(group) if inited else 0
        # force_staged: take the bucketed multi-stream path even in a 1-rank group (each all-reduce is then RCCL's one-rank copy): how the
        # stream / event choreography is exercised on a box with a single GPU (tests; training_config["force_bucketed_exchange"])
        self.staged = self.world > 1 or (bool(force_staged) and inited)
        self.stream = torch.cuda.Stream(device=engine.device) if (self.staged and engine.device.type == "cuda") else None
        self.skip_comm = False                                    # diagnostic only (bench: exposed communication time)
        self.timing, self._ev = False, []                         # diagnostic only: per-collective (label, MB, start, issue, done) events
        self.collectives = 0                                      # collectives issued so far (tests count them)
        self._wire_buf = self._shard_buf = self._amax = None      # scratch of the wire format / reduce_scatter shard (allocated on first use, largest bucket)
        if self.world > 1:
            engine.set_dynamic_items(True)        # RCCL's kernels hold CUs under the backward: the persistent GEMM draws its items with tickets then
            src = dist.get_global_rank(group, 0) if group is not None else 0
            for buf in (engine.params, engine.m, engine.v):
                dist.broadcast(buf, src=src, group=group)
            sc = torch.tensor([engine.step_count], dtype=torch.int64, device=engine.device)
            dist.broadcast(sc, src=src, group=group)
            engine.step_count = int(sc.item())
            if on_params_changed is not None:
                on_params_changed()

    # ---- one bucket (or a run of adjacent buckets) of the flat gradient buffer, summed over the ranks in place
    def _scratch(self, n):
        eng = self.eng
        if self.wire == "half" and (self._wire_buf is None or self._wire_buf.numel() < n):
            self._wire_buf = torch.empty(n, dtype=eng.wire_dtype, device=eng.device)
            if eng.wire_dtype == torch.float16 and self._amax is None:
                self._amax = torch.zeros(1025, dtype=torch.float32, device=eng.device)
        per = -(-n // max(self.world, 1))
        if self.exchange != "all_reduce" and (self._shard_buf is None or self._shard_buf.numel() < per or self._shard_buf.dtype != (eng.wire_dtype if self.wire == "half" else torch.float32)):
            self._shard_buf = torch.empty(per, dtype=eng.wire_dtype if self.wire == "half" else torch.float32, device=eng.device)

    def plan(self, lo, hi):
        """(wire bytes, 'all_reduce' | 'rs_ag') for the range — what exchange_range will do"""
        n = hi - lo
        nbytes = n * (2 if self.wire == "half" else 4)
        rs = self.exchange == "rs_ag" or (self.exchange == "auto" and nbytes >= self.rs_min_mb * 1e6)
        if n % max(self.world, 1):                                 # (never the case for the engine's buckets: their bounds are multiples of 128 floats)
            rs = False
        return nbytes, ("rs_ag" if rs else "all_reduce")

    def exchange_range(self, lo, hi):
        """grads[lo:hi] <- sum over ranks, on the CURRENT stream (the caller put the communication stream there and ordered it behind the producers)"""
        eng, dist, W = self.eng, self.dist, self.world
        n = hi - lo
        _, how = self.plan(lo, hi)
        self._scratch(n)
        amax = None
        if self.wire == "half":
            buf = self._wire_buf[:n]
            if eng.wire_dtype == torch.float16:                    # fp16: power-of-two scale from the global max |g| (one 4-byte all-reduce)
                amax = self._amax
                eng.wire_amax(lo, hi, amax)
                dist.all_reduce(amax[:1], op=dist.ReduceOp.MAX, group=self.group); self.collectives += 1
            eng.wire_pack(lo, hi, buf, amax, W)
        else:
            buf = eng.grads[lo:hi]
        if how == "rs_ag":
            shard = self._shard_buf[: n // W]
            dist.reduce_scatter_tensor(shard, buf, group=self.group)
            dist.all_gather_into_tensor(buf, shard, group=self.group)
            self.collectives += 2
        else:
            dist.all_reduce(buf, group=self.group); self.collectives += 1
        if self.wire == "half":
            eng.wire_unpack(lo, hi, buf, amax, W)

    def backward(self, dcmds=None, dpars=None):
        eng = self.eng
        if not self.staged:
            eng.backward(dcmds, dpars)
            return
        if self.stream is None:                                   # CPU / gloo (tests): sequential
            for st, (lo, hi) in enumerate(eng.buckets):
                eng.backward(dcmds, dpars, stage=st)
                self.exchange_range(lo, hi)
            return
        cur = torch.cuda.current_stream(eng.device)
        side = eng.side_stage
        t0 = None
        if self.timing:
            t0 = torch.cuda.Event(enable_timing=True); t0.record(cur); self._ev = []

        def reduce_range(b_lo, b_hi, wait_side=False):
            """exchange buckets b_lo..b_hi (adjacent ranges of the flat buffer) on the communication stream, behind everything enqueued on `cur`"""
            lo, hi = eng.buckets[b_lo][0], eng.buckets[b_hi][1]
            ev = torch.cuda.Event(); ev.record(cur)
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                if wait_side:
                    eng.join_side()                                 # comm stream <- completion event of the engine's side stream (no-op when it did not fork)
                if self.skip_comm:
                    return
                if self.timing:
                    a = torch.cuda.Event(enable_timing=True); a.record(self.stream)
                self.exchange_range(lo, hi)
                if self.timing:
                    b = torch.cuda.Event(enable_timing=True); b.record(self.stream)
                    nbytes, how = self.plan(lo, hi)
                    self._ev.append((f"{b_lo}-{b_hi}" if b_hi != b_lo else str(b_lo), nbytes / 1e6, t0, a, b, how))

        # The CAD ViT's stage (~230 small kernels) runs on the engine's side stream beside the frame ViT's stages.  Its bucket (+ the stem's) is
        # exchanged as soon as THAT stream is done: the communication stream — not the compute stream — waits for the side stream's completion
        # event (vcad_join_side on the communication stream), so the exchange runs under the frame ViT's stages instead of after them.  If the
        # engine did not fork (enable_past_states off, side stream disabled, profiler on) the stage ran on `cur` and join_side() is a no-op: the
        # event recorded on `cur` AFTER the stage's launch (reduce_range does that) is what orders the collective behind it; when it did fork
        # that event sits right behind the stem's kernels, so the wait costs nothing.
        eng.backward(dcmds, dpars, stage=0); reduce_range(0, 0)
        for st in range(1, side):
            eng.backward(dcmds, dpars, stage=st)
        eng.backward(dcmds, dpars, stage=side, side=True)
        reduce_range(1, side, wait_side=True)
        for st in range(side + 1, len(eng.buckets)):
            eng.backward(dcmds, dpars, stage=st); reduce_range(st, st)
        cur.wait_stream(self.stream)                                # (covers the side stream too: the comm stream waited for it)
        if self.timing:
            self._t_end = torch.cuda.Event(enable_timing=True); self._t_end.record(cur)

    def comm_report(self):
        """[{bucket, MB (on the wire), how, issue_ms, done_ms, GBps, busGBps}] of the last timed step (ms since the backward started) + when the backward's
        compute rejoined.  GBps = wire bytes / (done - issue) — the exchange as this rank saw it, pack / unpack passes of the half wire format
        included; busGBps = GBps x 2 (W - 1) / W, the per-link figure rccl-tests quotes (what to hold against xGMI's ~150 GB/s per link x 7)."""
        if not self._ev:
            return None
        torch.cuda.synchronize(self.eng.device)
        W = max(self.world, 1)
        rep = []
        for lab, mb, t0, a, b, how in self._ev:
            dur = a.elapsed_time(b)
            gbps = mb / 1e3 / (dur / 1e3) if dur > 0 else None
            rep.append({"bucket": lab, "MB": round(mb, 1), "how": how, "issue_ms": round(t0.elapsed_time(a), 3), "done_ms": round(t0.elapsed_time(b), 3),
                        "GBps": round(gbps, 1) if gbps else None, "busGBps": round(gbps * 2 * (W - 1) / W, 1) if gbps else None})
        return {"wire": self.wire if self.wire == "fp32" else str(self.eng.wire_dtype).replace("torch.", ""), "exchange": self.exchange,
                "collectives": rep, "backward_joined_ms": round(self._ev[0][2].elapsed_time(self._t_end), 3)}


# ------------------------------------------------------------------------------------------------ trainer
class BaseTrainer:
    def __init__(self, train_packet, val_packet, test_packet, model, training_config, device, rank=0):
        self.device, self.rank, self.is_master = device, rank, rank == 0
        self.training_config = training_config
        cfg = training_config.get
        self.checkpoint = cfg("checkpoint", False)
        self.early_stopping_enabled = cfg("early_stopping_enabled", False)          # reference :212-217
        self.early_stopping_patience = cfg("early_stopping_patience", 100)
        self.early_stopping_min_delta = cfg("early_stopping_min_delta", 0.0)
        self.early_stopping_metric = cfg("early_stopping_metric", "accuracy")
        self.early_stopping_mode = cfg("early_stopping_mode", "max")
        self.frozen = cfg("frozen", False)
        self.experiment_name = cfg("experiment_name", "default_" + datetime.datetime.now().strftime("%Y%m%d_%H%M%S"))
        self.metrics_handler = MetricsHandler(self.experiment_name, rank)
        self.checkpoint_handler = CheckpointHandler(self.experiment_name, rank, cfg("checkpoint_dir", "checkpoints"))
        self.train_loader, self.val_loader, self.test_loader = train_packet["loader"], val_packet["loader"], test_packet["loader"]
        self.train_sampler, self.val_sampler, self.test_sampler = train_packet["sampler"], val_packet["sampler"], test_packet["sampler"]
        self.model = model
        self.native = unwrap(model)                               # tolerate the reference harness's torch.compile / DDP wrappers
        self.engine = self.native._engine
        self.lr = cfg("lr", 1e-3)                                 # reference :235
        groups = {"lr_cad": cfg("lr_cad", 1e-3), "lr_state": cfg("lr_state", 1e-3)} if self.frozen else None
        self.optimizer = NativeAdam(self.native, self.lr, groups)
        self.use_mse = cfg("use_mse", False)
        self.reduce_eval_metrics = cfg("reduce_eval_metrics", True)
        # The reference "reloads" best_model_state at the end of train() (:370-380), but that state dict aliases the live parameters, so the
        # model it returns holds the LAST weights.  Default = that behaviour; restore_best_weights=True copies the best epoch's weights back.
        self.restore_best_weights = cfg("restore_best_weights", False)
        self.defer_unscale = cfg("defer_unscale", True)          # fp16 engines, single rank: the optimiser divides the gradient scale out (train_step)
        self.stage_inputs = cfg("stage_inputs", True)             # double-buffered H2D staging of the train loader (data.DeviceStager)
        self.native._drop_rank = rank                             # every rank draws its own dropout masks
        self.gradsync = GradSync(self.engine, on_params_changed=self._params_changed, force_staged=cfg("force_bucketed_exchange", False),
                                 wire=cfg("grad_wire", "fp32"), exchange=cfg("grad_exchange", "all_reduce"), rs_min_mb=cfg("grad_rs_min_mb", 32.0))
        self.action_mask = self.native.action_mask.to(device) if hasattr(self.native.action_mask, "to") else self.native.action_mask
        self._best_params = None

    def _params_changed(self):
        self.native._shadow_fresh = False
        if hasattr(self, "engine"):
            self._reset_overflow_watch()              # (norms of the replaced weights must not move the new run's gradient scale)

    def log(self, message):
        if self.is_master:
            print(message)

    def apply_action_mask(self, cmd_pred, param_pred):
        return self.native.apply_action_mask(cmd_pred, param_pred)

    def save_checkpoint(self, epoch, loss, is_best=False):
        return self.checkpoint_handler.save_checkpoint(epoch, loss, self.native, self.optimizer, is_best)

    def load_checkpoint(self, path_or_dict):
        """Resume: model weights + fused-Adam state (+ returns the stored epoch)."""
        ck = torch.load(path_or_dict, map_location="cpu") if isinstance(path_or_dict, (str, os.PathLike)) else path_or_dict
        from .model_factory import strip_prefixes
        self.native.load_state_dict(strip_prefixes(ck["model_state_dict"]), strict=False)
        if ck.get("optimizer_state_dict"):
            self.optimizer.load_state_dict(ck["optimizer_state_dict"])
        self._params_changed()
        return ck.get("epoch", 0)

    # ---- reference trainer.py:291-324 (uint8 frame batches stay uint8: they are normalised inside the patchify kernel)
    def prepare_batch(self, batch):
        def to_dev(t):
            return t.to(self.device, non_blocking=True) if t.dtype == torch.uint8 else t.to(self.device, dtype=torch.float, non_blocking=True)
        out = {"frames": to_dev(batch["frames"]), "actions": batch["actions"].to(self.device, dtype=torch.float, non_blocking=True),
               "cad_image": to_dev(batch["cad_image"])}
        if "timesteps" not in batch or batch["timesteps"] is None:
            out["timesteps"] = torch.zeros((out["frames"].size(0), 1), dtype=torch.long, device=self.device)
        else:
            out["timesteps"] = batch["timesteps"].to(self.device, dtype=torch.long)
        if batch.get("multiview_images", None) is not None:
            out["multiview_images"] = to_dev(batch["multiview_images"])
        return out

    # ---- reference trainer.py:800-804
    def normalize_actions(self, actions):
        """a copy with column 0 divided by 4 and the others by 1000 — the reference's clone + two slice assignments (five launches on a [B, T, 7] tensor at the head of
        every step) as ONE broadcast division by a cached divisor row: the same IEEE divisions, bit-identical values"""
        key = (actions.device, actions.dtype, actions.shape[-1])
        cache = self.__dict__.setdefault("_act_div", {})
        div = cache.get(key)
        if div is None:
            div = cache[key] = torch.tensor([4.0] + [1000.0] * (actions.shape[-1] - 1), device=actions.device, dtype=actions.dtype)
        return actions / div

    # ---- reference trainer.py:498-505
    def _add_noise_to_actions(self, actions):
        noise_actions = actions.clone()
        cmd_0 = (actions[:, :, 0] == 0).unsqueeze(-1)
        cmd_3 = (actions[:, :, 0] == 3).unsqueeze(-1)
        noise_actions[:, :, 1:3] += torch.randint_like(noise_actions[:, :, 1:3], -2, 3) * cmd_0
        noise_actions[:, :, -1:] += torch.randint_like(noise_actions[:, :, -1:], -2, 3) * cmd_3
        return noise_actions

    # ---- reference trainer.py:507-517
    def _prepare_model_inputs(self, batch_dict, noise):
        model_inputs = {"frames": batch_dict["frames"][:, :-1], "actions": self.normalize_actions(batch_dict["actions"][:, :-1]),
                        "timesteps": batch_dict["timesteps"], "cad_image": batch_dict["cad_image"]}
        if "multiview_images" in batch_dict:
            model_inputs["multiview_images"] = batch_dict["multiview_images"]
        return model_inputs

    # ---- reference trainer.py:480-496 — the hot path
    def _process_batch(self, batch, noise=False):
        bd = self.prepare_batch(batch)
        if noise:
            bd["actions"] = self._add_noise_to_actions(bd["actions"])
        loss, counters = self.train_step(bd)
        return loss, metrics_from_counters(counters.tolist())      # single D2H copy (the reference does ~40 .item() syncs)

    def train_step(self, bd):
        """device tensors in, device tensors out (no host sync): returns (loss 0-d tensor, int32[32] counters)."""
        eng = self.engine
        inputs = self._prepare_model_inputs(bd, False)
        if not self.native._shadow_fresh:
            eng.sync_shadow()
        self.native._arm_dropout()                               # model.train() -> dropout active, like the reference's train loop
        # fp16 engines, single rank: nothing reads gradients or dlogits between the loss and the optimiser here, so the gradient scale is divided out by the optimiser's
        # own passes instead of seven extra ones (include/vcad.h: vcad_set_defer_unscale; bit-identical updates).  Data-parallel runs keep true gradients in the buckets.
        defer = eng.cfg.dtype == L.VCAD_F16 and not self.gradsync.staged and self.defer_unscale
        if defer:
            eng.set_defer_unscale(True)
        try:
            cmds, pars = eng.forward(inputs["frames"], inputs["actions"], inputs["cad_image"], inputs.get("multiview_images") if self.native.num_views > 0 else None)
            out, met = eng.loss(cmds, pars, bd["actions"][:, 1:], self._label_w(), use_mse=self.use_mse, class_weights=self._class_w())
            self.gradsync.backward()
            norm = eng.optimizer_step(lr=self.optimizer.lr, betas=self.optimizer.betas, eps=self.optimizer.eps, max_norm=1.0,
                                      grad_scale=1.0 / self.gradsync.world)
        finally:
            if defer:
                eng.set_defer_unscale(False)                     # (also when the step raised: everything outside train_step sees true gradients and dlogits)
        self.native.mark_shadow_fresh()
        if eng.cfg.dtype == L.VCAD_F16:
            self._watch_overflow(norm)
        return out[0], met

    OVERFLOW_WINDOW = 32

    def _watch_overflow(self, norm):
        """fp16 engines: an overflowed backward leaves a non-finite gradient norm (the library skipped that update).  Non-finite norms are COUNTED ON
        THE DEVICE; the host reads one counter per OVERFLOW_WINDOW steps, and it reads the counter of the window BEFORE the one that just ended — so it
        never waits for a step it has just enqueued (r04 did one `.item()` per step once its ring was full).  Every rank sees the same
        all-reduced gradients, hence the same norms, counters and scale.  `_last_norm` keeps the newest norm tensor for diagnostics (tools/)."""
        eng = self.engine
        self._last_norm = norm
        if getattr(self, "_ovf_acc", None) is None:
            self._ovf_acc = torch.zeros((), dtype=torch.int32, device=norm.device); self._ovf_n = 0; self._ovf_prev = None; self._ovf_scale = eng.grad_scale
        self._ovf_acc += (~torch.isfinite(norm[0])).to(torch.int32)
        self._ovf_n += 1
        if self._ovf_n >= self.OVERFLOW_WINDOW:
            prev, self._ovf_prev = self._ovf_prev, (self._ovf_acc, self._ovf_n, self._ovf_scale)
            self._ovf_acc = torch.zeros((), dtype=torch.int32, device=norm.device); self._ovf_n = 0
            if prev is not None:
                bad = int(prev[0].item())                          # (a window that ended OVERFLOW_WINDOW steps ago)
                if eng.note_overflows(bad, prev[1], ran_at_scale=prev[2]):     # (a window that ran at a scale since lowered cannot lower it again)
                    self.log(f"fp16 gradient overflow: {bad} of {prev[1]} updates skipped, gradient scale now {eng.grad_scale:.0f}")
            self._ovf_scale = eng.grad_scale                       # the scale the window that starts now runs at

    def _reset_overflow_watch(self):
        self._ovf_acc = None; self._ovf_prev = None; self._ovf_n = 0
        self.engine.reset_overflow_history()

    def _class_w(self):
        return None

    def _label_w(self):
        raise NotImplementedError

    def compute_loss(self, action_preds, actions):
        raise NotImplementedError("Subclasses must implement compute_loss")

    def init_metrics(self):
        return {}

    def update_metrics(self, metrics, batch_metrics):
        pass

    def log_epoch_metrics(self, epoch, epochs, avg_loss, metrics):
        pass

    def print_metrics(self, metrics, mode=""):
        self.metrics_handler.print_metrics(metrics, mode)

    def save_metrics(self, metrics, ext=""):
        self.metrics_handler.save_metrics(metrics, ext)

    # ---- reference trainer.py:337-382
    def train(self, epochs, sequential=False, noise=False):
        self.model.train()
        best_value = float("inf") if self.early_stopping_mode == "min" else float("-inf")
        best_state, patience = None, 0
        t0 = time.time()
        for epoch in range(epochs):
            if self.train_sampler is not None and hasattr(self.train_sampler, "set_epoch"):
                self.train_sampler.set_epoch(epoch)
            avg_loss, metrics = self._train_epoch(epoch, noise)
            self.log_epoch_metrics(epoch, epochs, avg_loss, metrics)
            if (epoch + 1) % self.training_config.get("save_frequency", 10 ** 9) == 0:
                self.save_checkpoint(epoch, avg_loss)
            val_metrics = self._run_validation(epoch)
            d = _dist()
            if d is not None:
                d.barrier()
            best_value, patience, best_state, stop = self._handle_early_stopping(epoch, avg_loss, val_metrics, best_value, patience, best_state)
            if stop:
                self.log(f"Early stopping triggered after {epoch + 1} epochs")
                self._restore_best(best_state)
                break
            self.log(f"Epoch {epoch + 1} took {time.time() - t0:.2f} seconds")
            t0 = time.time()
        else:
            if self.early_stopping_enabled and best_state is not None and patience < self.early_stopping_patience:
                self._restore_best(best_state)
        return self.model

    def _restore_best(self, best_state):
        """The reference reloads `best_model_state['model_state_dict']` (:371), whose tensors alias the live parameters (a no-op there);
        here the best weights are a device-side copy of the flat buffer taken when the checkpoint was written."""
        if best_state is None or self._best_params is None or not self.restore_best_weights:
            if best_state is not None:
                self.log(f"Loaded best model from epoch {best_state['epoch']}")      # (the reference prints this; its load is an aliasing no-op)
            return
        with torch.no_grad():
            self.engine.params.copy_(self._best_params)
        self._params_changed()
        self.log(f"Loaded best model from epoch {best_state['epoch']}")

    def _make_profiler(self, epoch):
        """reference trainer.py:394-439: `enable_profiling` -> a torch.profiler over the first `profile_warmup_steps` + `profile_active_steps`
        batches of the epoch, traces under logs/<experiment>/profile_traces/epoch<E>/rank<R> (same keys, schedule and directory layout; on ROCm the
        CUDA activity is the HIP kernels of this library)."""
        if not self.training_config.get("enable_profiling", False):
            return None, 0, None
        warm, active = self.training_config.get("profile_warmup_steps", 5), self.training_config.get("profile_active_steps", 15)
        base = f"./logs/{self.experiment_name}/profile_traces"
        rank_dir = f"{base}/epoch{epoch}/rank{self.rank}"
        if self.is_master:
            os.makedirs(f"{base}/epoch{epoch}", exist_ok=True)
        d = _dist()
        if d is not None:
            d.barrier()
        try:
            os.makedirs(rank_dir, exist_ok=True)
        except OSError as e:
            print(f"Warning: Rank {self.rank} could not create profiler directory: {e}")
            rank_dir = f"./profile_traces_epoch{epoch}_rank{self.rank}"
            os.makedirs(rank_dir, exist_ok=True)
        acts = [torch.profiler.ProfilerActivity.CPU]
        if torch.device(self.device).type == "cuda":
            acts.append(torch.profiler.ProfilerActivity.CUDA)
        prof = torch.profiler.profile(activities=acts, schedule=torch.profiler.schedule(wait=0, warmup=warm, active=active, repeat=1),
                                      on_trace_ready=torch.profiler.tensorboard_trace_handler(rank_dir), record_shapes=True, profile_memory=True,
                                      with_stack=True)
        prof.__enter__()
        return prof, warm + active, rank_dir

    # ---- reference trainer.py:384-478
    def _train_epoch(self, epoch, noise=False):
        self.model.train()
        metrics = self.init_metrics()
        prof, prof_steps, prof_dir = self._make_profiler(epoch)
        loader = DeviceStager(self.train_loader, self.device) if (self.stage_inputs and torch.device(self.device).type == "cuda") else self.train_loader
        running = torch.zeros((), device=self.device)
        counters = torch.zeros(L.NMETRIC, dtype=torch.int64, device=self.device)
        n, log_every = 0, self.training_config.get("log_frequency", 2)      # reference :457-458: every 2nd batch (0 = never; each log is one host sync)
        for batch_idx, batch in enumerate(loader):
            bd = self.prepare_batch(batch)
            if noise:
                bd["actions"] = self._add_noise_to_actions(bd["actions"])
            loss, met = self.train_step(bd)
            running += loss.detach(); counters += met; n += 1      # stays on the device: no per-step host sync
            if log_every and (batch_idx + 1) % log_every == 0:
                m = self.init_metrics(); self.update_metrics(m, metrics_from_counters(counters.tolist()))
                self.log_metrics(epoch, self.training_config.get("epochs", 0), batch_idx, len(self.train_loader), float(loss), metrics=m)
            if prof is not None and batch_idx < prof_steps:
                prof.step()
        if prof is not None:
            prof.__exit__(None, None, None)
            self.log(f"Profiler trace for epoch {epoch} (rank {self.rank}) saved to: {prof_dir}")
        self.update_metrics(metrics, metrics_from_counters(counters.tolist()))
        return (float(running) / n if n else 0.0), metrics

    def _run_validation(self, epoch):
        val_metrics = None
        if (epoch + 1) % self.training_config.get("val_frequency", 10 ** 9) == 0:
            val_metrics = self.evaluate(self.model, mode="val", epoch=epoch)
            self.print_metrics(val_metrics, mode="Validation")
            self.model.train()
        return val_metrics

    def _handle_early_stopping(self, epoch, avg_loss, val_metrics, best_value, patience, best_state):
        if not self.early_stopping_enabled:
            return best_value, patience, best_state, False
        # ONE decision for all ranks (no rank may leave the loop alone: the next all-reduce would deadlock), computed ONCE from all-reduced raw
        # numerators / denominators — not from an average of per-rank values, which would mix a rank that fell back to its loss with ranks that
        # report accuracy, and let one rank's non-finite loss poison every rank's decision
        correct = float(val_metrics.get("correct_predictions", 0)) if val_metrics else 0.0
        total = float(val_metrics.get("total_predictions", 0)) if val_metrics else 0.0
        finite = 1.0 if (avg_loss == avg_loss and abs(avg_loss) != float("inf")) else 0.0
        v = [correct, total, float(avg_loss) if finite else 0.0, finite]
        d = _dist()
        if d is not None:
            t = torch.tensor(v, dtype=torch.float64, device=self.device)
            d.all_reduce(t); v = t.tolist()
        if self.early_stopping_metric == "accuracy" and v[1] > 0:
            cur = v[0] / v[1]
        elif v[3] > 0:
            cur = v[2] / v[3]                                        # mean training loss over the ranks that have a finite one
        else:
            cur = float("inf") if self.early_stopping_mode == "min" else float("-inf")     # nothing usable anywhere: counts as "no improvement"
        improved = cur < best_value - self.early_stopping_min_delta if self.early_stopping_mode == "min" else cur > best_value + self.early_stopping_min_delta
        if improved:
            self.log(f"Validation {self.early_stopping_metric} improved from {best_value:.4f} to {cur:.4f}")
            best_value, patience = cur, 0
            ck = self.save_checkpoint(epoch, avg_loss, is_best=True)
            best_state = {"epoch": epoch + 1} if ck is None else {"epoch": ck["epoch"]}
            if self.restore_best_weights:
                self._best_params = self.engine.params.clone()
        else:
            patience += 1
            self.log(f"Validation {self.early_stopping_metric} did not improve. Patience: {patience}/{self.early_stopping_patience}")
        stop = patience >= self.early_stopping_patience
        if d is not None:                                          # reference :559-563: stop only when every rank wants to
            flag = torch.tensor([int(stop)], device=self.device)
            d.all_reduce(flag, op=d.ReduceOp.MIN)
            stop = bool(flag.item())
        return best_value, patience, best_state, stop

    # ---- reference trainer.py:713-750
    @torch.no_grad()
    def evaluate(self, model, mode="test", ablation=False, epoch=-1):
        loader = {"train": self.train_loader, "val": self.val_loader}.get(mode, self.test_loader)
        model.eval()
        metrics = self.init_metrics()
        native = unwrap(model)
        counters = torch.zeros(L.NMETRIC, dtype=torch.int64, device=self.device)
        loss_sum = torch.zeros(2, dtype=torch.float64, device=self.device)           # [sum of batch losses, batches]
        fused = hasattr(native, "_engine")
        for batch in loader:
            bd = self.prepare_batch(batch)
            inputs = self._prepare_model_inputs(bd, False)
            if ablation:
                inputs["cad_image"] = torch.zeros_like(inputs["cad_image"])
            if fused:                                               # fused path: counters stay on the device
                cmds, pars = native(inputs)
                out, met = native._engine.loss(cmds, pars, bd["actions"][:, 1:], self._label_w(), use_mse=self.use_mse, class_weights=self._class_w())
                counters += met; loss_sum[0] += out[0].double(); loss_sum[1] += 1
            else:
                loss, bm = self.compute_loss(model(inputs), bd["actions"][:, 1:])
                counters += torch.tensor([bm["cmd_corrects"][i] for i in range(5)] + [bm["cmd_counts"][i] for i in range(5)] +
                                         [bm["param_corrects"][i] for i in range(6)] + [bm["param_counts"][i] for i in range(6)] +
                                         [bm["cmd_correct_topk"], bm["cmd_counts_topk"], bm["param_correct_topk"], bm["param_counts_topk"],
                                          bm["correct_predictions"], bm["total_predictions"], 0, 0, 0, 0], dtype=torch.int64, device=self.device)
                loss_sum[0] += float(loss); loss_sum[1] += 1
        d = _dist()
        if d is not None and self.reduce_eval_metrics:              # the reference reports per-rank numbers (quirk 6); here: the whole split
            d.all_reduce(counters); d.all_reduce(loss_sum)
        self.update_metrics(metrics, metrics_from_counters(counters.tolist()))
        metrics["loss"] = float(loss_sum[0] / loss_sum[1]) if float(loss_sum[1]) else 0.0
        if self.is_master:
            self.save_metrics(metrics, f"{mode}_epoch_{epoch + 1}" if epoch != -1 else mode)
        return metrics

    def _predict_actions(self, model, inputs):
        cmds, pars = model(inputs)
        pred_cmd = torch.argmax(cmds, dim=-1)
        pred_params = self.apply_action_mask(pred_cmd, torch.argmax(pars, dim=-1)).long()
        return pred_cmd.long(), pred_params

    # ---- reference trainer.py:1066-1128: argmax predictions of n random clips as CSV next to the ground truth
    @torch.no_grad()
    def sample(self, model, n=10, folder="outputs", mode="test", ablation=False):
        model.eval()
        loader = {"train": self.train_loader, "val": self.val_loader}.get(mode, self.test_loader)
        dataset = loader.dataset
        os.makedirs(folder, exist_ok=True)
        for idx in random.sample(range(len(dataset)), n):
            item = dataset[idx]
            files = getattr(dataset, "data_files", None)
            sample_id = os.path.basename(files[idx]).split("_")[0] if files is not None else str(idx)
            out_path = os.path.join(folder, f"pred_actions_{sample_id}.csv")
            if os.path.exists(out_path):
                continue
            bd = self.prepare_batch({k: v for k, v in item.items() if v is not None})
            if ablation:
                bd["cad_image"] = torch.zeros_like(bd["cad_image"])
            inputs = {"frames": bd["frames"].unsqueeze(0)[:, :-1], "actions": self.normalize_actions(bd["actions"].unsqueeze(0))[:, :-1],
                      "timesteps": bd["timesteps"].unsqueeze(0), "cad_image": bd["cad_image"].unsqueeze(0)}
            pred_cmd, pred_params = self._predict_actions(model, inputs)
            pred = torch.cat((pred_cmd.unsqueeze(-1), pred_params), dim=-1)[0].cpu().numpy()
            with open(out_path, "w", newline="") as f:
                csv.writer(f).writerows(row.tolist() for row in pred)
            with open(os.path.join(folder, f"actions_{sample_id}.csv"), "w", newline="") as f:
                csv.writer(f).writerows(row.tolist() for row in bd["actions"][1:].cpu().numpy())
            try:                                                     # the reference saves the CAD image with torchvision.utils.save_image
                from PIL import Image
                img = bd["cad_image"][0].float().cpu()
                if bd["cad_image"].dtype != torch.uint8:
                    img = (img.clamp(0, 1) * 255 + 0.5)
                Image.fromarray(img.clamp(0, 255).to(torch.uint8).numpy()).save(os.path.join(folder, f"images_{sample_id}.png"))
            except Exception:
                pass

    # ---- reference trainer.py:1131-1260
    @staticmethod
    def _param_error(diff, k, tolerance):
        if k in (0, 1, 5):
            return abs(diff) > tolerance
        return diff < 0 or diff >= {2: 50, 3: 200, 4: 500}[k]

    def _sequence_mistakes(self, a_cmd, a_par, p_cmd, p_par, tolerance):
        n = len(a_cmd)
        rec = {"First Mistakes": {**{f"cmd_{i}": [] for i in range(5)}, **{f"param_{i}": [] for i in range(6)}},
               "Memory": {"cmd": [], **{f"param_{i}": [] for i in range(6)}}, "Sequence Lengths": [], "Number of Mistakes": []}
        mistakes, first, noted = [0] * n, False, False
        for j in range(n):
            bad = False
            gt, pd = int(a_cmd[j]), int(p_cmd[j])
            rec["Memory"]["cmd"].append([gt, pd])
            if gt != pd:
                mistakes[j], bad = 1, True
                if not first:
                    rec["First Mistakes"][f"cmd_{gt}"].append(f"cmd_{pd}"); first = True
            for k in range(a_par.shape[-1]):
                g = int(a_par[j][k])
                if g == -1:
                    continue
                q = int(p_par[j][k])
                rec["Memory"][f"param_{k}"].append([g, q])
                err = self._param_error(q - g, k, tolerance)
                if err and not bad:
                    mistakes[j], bad = 1, True
                if err and not first:
                    rec["First Mistakes"][f"param_{k}"].append(f"param_{q}"); first = True
            if first and not noted:
                rec["Sequence Lengths"], noted = [j, n], True
        if not noted:
            rec["Sequence Lengths"] = [n, n]
        rec["Number of Mistakes"] = mistakes
        return rec

    @torch.no_grad()
    def find_first_mistake(self, model, mode="test", tol=3, ablation=False):
        model.eval()
        blank = lambda: {"First Mistakes": {**{f"cmd_{i}": [] for i in range(5)}, **{f"param_{i}": [] for i in range(6)}},
                         "Memory": {"cmd": [], **{f"param_{i}": [] for i in range(6)}}, "Sequence Lengths": [], "Number of Mistakes": []}
        data = [blank() for _ in range(tol)]
        loader = {"train": self.train_loader, "val": self.val_loader}.get(mode, self.test_loader)
        for batch in loader:
            bd = self.prepare_batch(batch)
            inputs = self._prepare_model_inputs(bd, False)
            if ablation:
                inputs["cad_image"] = torch.zeros_like(inputs["cad_image"])
            pred_cmd, pred_params = self._predict_actions(model, inputs)
            # one D2H of the arg-max predictions; the per-step bookkeeping is host work (it builds Python lists)
            a_cmd = bd["actions"][:, 1:, 0].long().cpu().numpy(); a_par = bd["actions"][:, 1:, 1:].long().cpu().numpy()
            p_cmd = pred_cmd.cpu().numpy(); p_par = pred_params.cpu().numpy()
            for t in range(tol):
                for i in range(len(a_cmd)):
                    rec = self._sequence_mistakes(a_cmd[i], a_par[i], p_cmd[i], p_par[i], t)
                    for key in rec["First Mistakes"]:
                        data[t]["First Mistakes"][key].extend(rec["First Mistakes"][key])
                    for key in rec["Memory"]:
                        data[t]["Memory"][key].extend(rec["Memory"][key])
                    data[t]["Sequence Lengths"].append(rec["Sequence Lengths"])
                    data[t]["Number of Mistakes"].append(rec["Number of Mistakes"])
        return data


class MultiClassesTrainer(BaseTrainer):
    def __init__(self, train_loader, val_loader, test_loader, model, training_config, device, rank):
        super().__init__(train_loader, val_loader, test_loader, model, training_config, device, rank=rank)
        # ./class_weights.json, CWD-relative and unconditional like the reference (:822-825): "Label" holds the 5 command-CE class
        # weights, which are also the six per-head loss multipliers through [0,0,1,1,2,3] (:962); the other six keys are the
        # per-class weights of the parameter heads (use_mse = False)
        with open(training_config.get("class_weights_path", "class_weights.json"), "r") as f:
            self.weights = json.load(f)
        self.cmd_weights = [float(x) for x in self.weights["Label"]]
        if len(self.cmd_weights) != 5:
            raise ValueError("class_weights.json: 'Label' must hold 5 weights")
        self.param_to_label = [0, 0, 1, 1, 2, 3]
        self.tolerances = [TOLERANCE - 1, TOLERANCE - 1, 50, 200, 500, TOLERANCE - 1]
        self.above = [False, False, True, True, True, False]
        self.param_names = PARAM_NAMES
        self._cw = None

    def _label_w(self):
        return self.cmd_weights

    def _class_w(self):
        if self.use_mse:
            return None
        if self._cw is None:
            self._cw = torch.tensor([self.weights[k] for k in self.param_names[1:]], dtype=torch.float32, device=self.engine.device).contiguous()
        return self._cw

    def compute_loss(self, action_preds, actions, mse=True):
        """reference trainer.py:935-1063 on the fused loss kernels: (loss, metrics dict).  The loss tensor is a plain device
        scalar; when the predictions carry autograd history the gradient is attached through the engine's own dlogits."""
        cmds, pars = action_preds
        out, met = self.engine.loss(cmds.detach(), pars.detach(), actions, self._label_w(), use_mse=self.use_mse, class_weights=self._class_w())
        loss = out[0]
        if cmds.requires_grad:
            B, T = cmds.shape[:2]
            loss = _LossWithGrad.apply(cmds, pars, loss.clone(), self.engine.dl_views(B, T))
        return loss, metrics_from_counters(met.tolist())

    # ---- reference trainer.py:1265-1340: running totals + derived percentages
    def init_metrics(self):
        m = {k: 0 for k in ("correct_predictions", "total_predictions", "cmd_accuracy", "params_accuracy", "cmd_corrects", "cmd_counts",
                            "param_corrects", "param_counts", "cmd_correct_topk", "param_correct_topk", "cmd_counts_topk", "param_counts_topk",
                            "cmd_accuracy_topk", "param_accuracy_topk", "perfect_sequences", "total_sequences", "perfect_commands",
                            "perfect_command_accuracy", "perfect_sequence_accuracy")}
        for i in range(6):
            m[f"param_accuracy_{i}"] = m[f"param_corrects_{i}"] = m[f"param_counts_{i}"] = 0
        for i in range(5):
            m[f"cmd_accuracy_{i}"] = m[f"cmd_corrects_{i}"] = m[f"cmd_counts_{i}"] = 0
        return m

    def update_metrics(self, metrics, bm):
        pct = lambda a, b: 100 * a / b
        for k in ("cmd_correct_topk", "param_correct_topk", "cmd_counts_topk", "param_counts_topk", "correct_predictions", "total_predictions",
                  "perfect_sequences", "perfect_commands", "total_sequences"):
            metrics[k] += bm[k]
        for i in range(6):
            metrics[f"param_corrects_{i}"] += bm[f"param_corrects_{i}"]; metrics[f"param_counts_{i}"] += bm[f"param_counts_{i}"]
            if metrics[f"param_counts_{i}"] > 0:
                metrics[f"param_accuracy_{i}"] = pct(metrics[f"param_corrects_{i}"], metrics[f"param_counts_{i}"])
        for i in range(5):
            metrics[f"cmd_corrects_{i}"] += bm[f"cmd_corrects_{i}"]; metrics[f"cmd_counts_{i}"] += bm[f"cmd_counts_{i}"]
            if metrics[f"cmd_counts_{i}"] > 0:
                metrics[f"cmd_accuracy_{i}"] = pct(metrics[f"cmd_corrects_{i}"], metrics[f"cmd_counts_{i}"])
        if metrics["cmd_counts_topk"] > 0:
            metrics["cmd_accuracy_topk"] = pct(metrics["cmd_correct_topk"], metrics["cmd_counts_topk"])
        if metrics["param_counts_topk"] > 0:
            metrics["param_accuracy_topk"] = pct(metrics["param_correct_topk"], metrics["param_counts_topk"])
        tc, tp = sum(metrics[f"cmd_counts_{i}"] for i in range(5)), sum(metrics[f"param_counts_{i}"] for i in range(6))
        if tc > 0:
            metrics["cmd_accuracy"] = pct(sum(metrics[f"cmd_corrects_{i}"] for i in range(5)), tc)
        if tp > 0:
            metrics["params_accuracy"] = pct(sum(metrics[f"param_corrects_{i}"] for i in range(6)), tp)
        if metrics["total_predictions"] > 0:
            metrics["overall_accuracy"] = pct(metrics["correct_predictions"], metrics["total_predictions"])
        if metrics["total_sequences"] > 0:
            metrics["perfect_sequence_accuracy"] = pct(metrics["perfect_sequences"], metrics["total_sequences"])
            metrics["perfect_command_accuracy"] = pct(metrics["perfect_commands"], metrics["total_sequences"])

    def log_metrics(self, epoch, epochs, batch_idx, loader_len, loss, **kwargs):
        m = kwargs.get("metrics", {})
        self.save_metrics(m, ext=f"epoch_{epoch + 1}")
        self.log(f"Epoch [{epoch + 1}/{epochs}], Batch [{batch_idx + 1}/{loader_len}], Loss: {loss:.4f}, CMD Accuracy: {m['cmd_accuracy']:.2f}%, "
                 f"Params Accuracy: {m['params_accuracy']:.2f}%")

    def log_epoch_metrics(self, epoch, epochs, avg_loss, metrics):
        tot = metrics["total_predictions"]                       # (the reference divides unguarded and raises on an empty epoch, quirk 11)
        acc = 100 * metrics["correct_predictions"] / tot if tot else 0.0
        self.log(f"Epoch [{epoch + 1}/{epochs}] Average Loss: {avg_loss:.4f}, Average Accuracy: {acc:.2f}%, CMD Accuracy: {metrics['cmd_accuracy']:.2f}%, "
                 f"Params Accuracy: {metrics['params_accuracy']:.2f}%, Top-30 CMD Accuracy: {metrics['cmd_accuracy_topk']:.2f}%, "
                 f"Top-30 Params Accuracy: {metrics['param_accuracy_topk']:.2f}%")


class _LossWithGrad(torch.autograd.Function):
    """Makes `loss.backward()` work for the reference-style sequence model(inputs) -> compute_loss -> backward."""

    @staticmethod
    def forward(ctx, cmds, pars, loss, dl):
        ctx.dl = dl
        return loss

    @staticmethod
    def backward(ctx, g):
        dc, dp = ctx.dl
        return dc * g, dp.view(dp.shape[0], dp.shape[1], 6, -1) * g, None, None


def create_trainer(train_packet, val_packet, test_packet, model, training_config, device, model_type, rank=0):
    """reference trainer.py:1384-1385"""
    assert model_type == ModelType.MULTI_CLASSES
    return MultiClassesTrainer(train_packet, val_packet, test_packet, model, training_config, device, rank)
