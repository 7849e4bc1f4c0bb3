"""Trainer with the reference's surface (reference trainer.py:193-1385): `create_trainer(...)`, `.train(epochs)`,
`._process_batch(batch, noise=False) -> (loss, metrics)`, `.compute_loss(action_preds, actions)`, `.evaluate(model, mode)`.

The hot loop (`_process_batch`, reference :480-496) is ONE pass through the C ABI:
    vcad_forward -> vcad_loss (loss + ~45 metric counters + dlogits, no host sync) -> vcad_backward_stage x4
    (gradient all-reduce of each finished bucket over RCCL on a side stream, world_size > 1) -> vcad_optimizer_step
    (global-norm clip 1.0 + Adam, reference :493-494).
Epoch loop / logging / checkpoints are plain Python plumbing kept deliberately small (out of scope).
"""
from __future__ import annotations

import json
import os
import time
from typing import Optional

import torch

from . import lib as L
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


class GradSync:
    """Data-parallel gradient exchange (replaces the DDP wrap at reference experiment.py:104-109).

    Each backward stage finalises one contiguous bucket of the flat gradient buffer (heads+decoder+stem first — 75 % of the
    bytes — then CAD ViT, then the two halves of the frame ViT); its all-reduce(SUM) is issued on a side stream right away
    so it runs over xGMI underneath the next stage's kernels.  The 1/world mean is folded into the Adam kernel.
    Only live parameters travel (508 MB fp32 instead of the reference's 818 MB incl. dead GPT-2 zeros)."""

    def __init__(self, engine, group=None):
        import torch.distributed as dist
        self.eng, self.dist, self.group = engine, dist, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.stream = torch.cuda.Stream(device=engine.device) if (self.world > 1 and engine.device.type == "cuda") else None

    def backward(self, dcmds=None, dpars=None):
        eng = self.eng
        if self.world == 1:
            eng.backward(dcmds, dpars)
            return
        if self.stream is None:                                   # CPU / gloo (tests): sequential
            for st, (lo, hi) in enumerate(eng.buckets):
                eng.backward(dcmds, dpars, stage=st)
                self.dist.all_reduce(eng.grads[lo:hi], group=self.group)
            return
        cur = torch.cuda.current_stream(eng.device)

        def reduce_bucket(st):
            lo, hi = eng.buckets[st]
            ev = torch.cuda.Event(); ev.record(cur)
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                self.dist.all_reduce(eng.grads[lo:hi], group=self.group)

        # stage 1 (CAD ViT: ~230 small kernels) runs on the engine's side stream beside the frame ViT's stages 2-3; its bucket is
        # reduced last, after join_side() has ordered it before `cur`
        eng.backward(dcmds, dpars, stage=0); reduce_bucket(0)
        eng.backward(dcmds, dpars, stage=1, side=True)
        for st in range(2, len(eng.buckets)):
            eng.backward(dcmds, dpars, stage=st); reduce_bucket(st)
        eng.join_side(); reduce_bucket(1)
        cur.wait_stream(self.stream)


class BaseTrainer:
    def __init__(self, train_packet, val_packet, test_packet, model, training_config, device, rank=0):
        self.device, self.rank, self.is_master = device, rank, rank == 0
        self.training_config = training_config
        self.train_loader, self.val_loader, self.test_loader = train_packet["loader"], val_packet["loader"], test_packet["loader"]
        self.train_sampler, self.val_sampler, self.test_sampler = train_packet["sampler"], val_packet["sampler"], test_packet["sampler"]
        self.model = model
        self.native = getattr(model, "module", model)            # tolerate a DDP-style wrapper
        self.engine = self.native._engine
        self.lr = training_config.get("lr", 1e-3)                 # reference :235
        if training_config.get("frozen", False):
            raise NotImplementedError("per-group learning rates ('frozen', reference :237-251) are not on the native path yet")
        self.use_mse = training_config.get("use_mse", False)
        self.experiment_name = training_config.get("experiment_name", "default_" + time.strftime("%Y%m%d_%H%M%S"))
        self.early_stopping_enabled = training_config.get("early_stopping_enabled", False)
        self.early_stopping_patience = training_config.get("early_stopping_patience", 100)
        self.gradsync = GradSync(self.engine)
        self.action_mask = self.native.action_mask

    def log(self, message):
        if self.is_master:
            print(message)

    # ---- reference trainer.py:291-324
    def prepare_batch(self, batch):
        out = {"frames": batch["frames"].to(self.device, dtype=torch.float, non_blocking=True),
               "actions": batch["actions"].to(self.device, dtype=torch.float, non_blocking=True),
               "cad_image": batch["cad_image"].to(self.device, dtype=torch.float, non_blocking=True)}
        if "timesteps" not in batch:
            out["timesteps"] = torch.zeros((out["frames"].size(0), 1), dtype=torch.long, device=self.device)
        else:
            out["timesteps"] = batch["timesteps"].to(self.device, dtype=torch.long)
        if batch.get("multiview_images", None) is not None:
            out["multiview_images"] = batch["multiview_images"].to(self.device, dtype=torch.float)
        return out

    # ---- reference trainer.py:800-804
    def normalize_actions(self, actions):
        actions = actions.clone()
        actions[:, :, 0] = actions[:, :, 0] / 4.0
        actions[:, :, 1:] = actions[:, :, 1:] / 1000.0
        return actions

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

    train_step_name = "train_step"

    def train_step(self, bd):
        """device tensors in, device tensors out (no host sync): returns (loss 0-d tensor, int32[32] counters)."""
        eng = self.engine
        inputs = self._prepare_model_inputs(bd, False)
        if not self.native._shadow_fresh:
            eng.sync_shadow()
        self.native._arm_dropout()                               # model.train() -> dropout active, like the reference's train loop
        cmds, pars = eng.forward(inputs["frames"], inputs["actions"], inputs["cad_image"])
        out, met = eng.loss(cmds, pars, bd["actions"][:, 1:], use_mse=self.use_mse, class_weights=self._class_w())
        self.gradsync.backward()
        eng.optimizer_step(lr=self.lr, max_norm=1.0, grad_scale=1.0 / self.gradsync.world)
        self.native.mark_shadow_fresh()
        return out[0], met

    def _class_w(self):
        return None

    def compute_loss(self, action_preds, actions):
        raise NotImplementedError("Subclasses must implement compute_loss")

    # ---- plumbing: reference trainer.py:337-478 reduced to its contract
    def train(self, epochs, sequential=False, noise=False):
        self.model.train()
        best = None
        for epoch in range(epochs):
            if self.train_sampler is not None and hasattr(self.train_sampler, "set_epoch"):
                self.train_sampler.set_epoch(epoch)
            t0, running, n, agg = time.time(), 0.0, 0, {}
            for batch in self.train_loader:
                loss, metrics = self._process_batch(batch, noise)
                running += float(loss.item()); n += 1
                for k, v in metrics.items():
                    if isinstance(v, int):
                        agg[k] = agg.get(k, 0) + v
            acc = 100.0 * agg.get("correct_predictions", 0) / max(agg.get("total_predictions", 0), 1)
            self.log(f"Epoch [{epoch + 1}/{epochs}] loss {running / max(n, 1):.4f} acc {acc:.2f}% ({time.time() - t0:.1f}s)")
            best = running / max(n, 1) if best is None else min(best, running / max(n, 1))
        return self.model

    @torch.no_grad()
    def evaluate(self, model, mode="test", ablation=False, epoch=-1):
        loader = {"train": self.train_loader, "val": self.val_loader, "test": self.test_loader}[mode]
        model.eval()
        agg, total_loss, n = {}, 0.0, 0
        for batch in loader:
            bd = self.prepare_batch(batch)
            preds = model(self._prepare_model_inputs(bd, False))
            loss, metrics = self.compute_loss(preds, bd["actions"][:, 1:])
            total_loss += float(loss.item()); n += 1
            for k, v in metrics.items():
                if isinstance(v, int):
                    agg[k] = agg.get(k, 0) + v
        agg["loss"] = total_loss / max(n, 1)
        agg["accuracy"] = 100.0 * agg.get("correct_predictions", 0) / max(agg.get("total_predictions", 0), 1)
        return agg


class MultiClassesTrainer(BaseTrainer):
    def __init__(self, train_loader, val_loader, test_loader, model, training_config, device, rank):
        super().__init__(train_loader, val_loader, test_loader, model, training_config, device, rank=rank)
        self.param_to_label = [0, 0, 1, 1, 2, 3]
        self.tolerances = [TOLERANCE - 1, TOLERANCE - 1, 50, 200, 500, TOLERANCE - 1]
        self.above = [False, False, True, True, True, False]
        self.param_names = PARAM_NAMES
        self._cw = None
        path = "class_weights.json"                               # CWD-relative like the reference (:822)
        if os.path.exists(path):
            with open(path) as f:
                self.weights = json.load(f)
            self.cmd_weights = self.weights["Label"]
        else:
            self.weights, self.cmd_weights = None, None
            if not self.use_mse:
                raise FileNotFoundError("class_weights.json (needed for use_mse=False) not found in the working directory")

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
        out, met = self.engine.loss(cmds.detach(), pars.detach(), actions, use_mse=self.use_mse, class_weights=self._class_w())
        loss = out[0]
        if cmds.requires_grad:
            B, T = cmds.shape[:2]
            loss = _LossWithGrad.apply(cmds, pars, loss.clone(), self.engine.dl_views(B, T))
        return loss, metrics_from_counters(met.tolist())


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
