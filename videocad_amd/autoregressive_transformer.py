"""Drop-in `AutoRegressiveTransformer` (reference model/autoregressive_transformer.py:6-275) whose arithmetic runs in
libvcad_hip.so.  Same constructor kwargs, same `forward(inputs) -> (cmds [B,T,5], params [B,T,6,1000])`, same
`state_dict()` keys for every LIVE parameter, same helpers (`apply_action_mask`,
`normalize_actions`, `sequential_inference`).

Differences that are deliberate and documented in DESIGN.md:
  * the 77.6 M dead parameters of the reference (GPT-2 trunk, embed_timestep, embed_ln, predict_action — constructed at
    model/base_transformer.py:38-60 but never used by forward) are not materialised; the reference loads checkpoints with
    strict=False (model/model_factory.py:35) so both directions interoperate
  * parameters are views into ONE flat fp32 buffer (and `.grad` views into one flat gradient buffer) so that clipping,
    Adam and the RCCL all-reduce run over contiguous memory
  * every wiring of forward (:149-213) is native: past actions and/or past states, with or without the timestep embedding;
    parameters a wiring never touches (e.g. image_projection without both flags) keep a zero gradient and are left unchanged
    by the optimiser, as in the reference where their .grad stays None; encoder != "vit" raises
  * dropout is a stateless counter-based mask (hash of step seed, site, element index) regenerated in the backward:
    statistically equivalent to nn.Dropout at the same sites, not bit-identical to torch's Philox stream
  * `compute_dtype` (extra kwarg / JSON key): "f16" (DEFAULT since r05 — 16-bit tensors with ten mantissa bits on libvcad_hip_f16.so: logits within 1e-3
    of the fp32 reference and arg-max exact, north_star's tolerance, at 98 % of the bf16 mode's speed; gradient scale handled by the engine and the
    trainer), "bf16" (the throughput mode of BASELINE's metric: 3.9e-3 on logits), "bf16x3" (fp32 tensors, 7e-6), "f32" (exact fp32 MFMA)
There is no CPU fallback: calling forward on a module whose buffers are not device memory raises (the library refuses).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import lib as L
from .engine import NativeEngine, make_config


class _Holder(nn.Module):
    """Parameter container mirroring the reference module tree (no compute).  Numeric children index like the
    reference's ModuleList / Sequential (`transformer_decoder.layers[3]`, `transformer.layers[i][0]`)."""

    def __getitem__(self, i):
        return self._modules[str(i)]

    def __len__(self):
        return sum(1 for k in self._modules if k.isdigit())

    def __iter__(self):
        return (self._modules[k] for k in sorted((k for k in self._modules if k.isdigit()), key=int))


def _attach(root: nn.Module, dotted: str, p: nn.Parameter):
    parts = dotted.split(".")
    m = root
    for part in parts[:-1]:
        if not hasattr(m, part):
            m.add_module(part, _Holder())
        m = getattr(m, part)
    m.register_parameter(parts[-1], p)


class _EngineFn(torch.autograd.Function):
    """autograd bridge: forward -> vcad_forward, backward -> vcad_backward (gradients land in the flat grad buffer and
    are handed to autograd as views, so `loss.backward()`, `clip_grad_norm_` and torch optimisers work unchanged)."""

    @staticmethod
    def forward(ctx, model, frames, actions, cad, mv, *params):
        cmds, pars = model._engine.forward(frames, actions, cad, mv)
        ctx.model = model
        ctx.fwd_id = model._engine.fwd_serial       # ANY later engine forward (no_grad / eval forwards, trainer.train_step, evaluate, cached
        return cmds, pars                           # inference) bumps it: a backward through this node then raises instead of using their activations

    @staticmethod
    def backward(ctx, dcmds, dpars):
        model = ctx.model
        eng = model._engine
        if ctx.fwd_id != eng.fwd_serial:
            raise RuntimeError("videocad_amd: backward through a forward that is no longer the engine's current one (the engine keeps the "
                               "activations of the most recent forward only)")
        eng.backward(dcmds.contiguous(), dpars.reshape(dpars.shape[0], dpars.shape[1], -1).contiguous())
        # the engine WRITES its flat gradient buffer on every backward; autograd may adopt what we return as p.grad and add the next
        # backward into it in place, so hand it a private copy (one flat clone, parameters are views of it)
        flat = eng.grads.clone()
        grads = tuple(eng.view(n, flat) for n in model._param_names)
        return (None, None, None, None, None) + grads


class AutoRegressiveTransformer(nn.Module):
    def __init__(self, state_dim, act_dim, hidden_size, max_length=None, max_ep_len=1000, action_tanh=True,
                 enable_past_actions=False, enable_past_states=False, enable_timestep_embedding=False, num_classes=5,
                 num_params=6, num_params_values=1000, num_decoder_layers=8, dim_feedforward=512,
                 use_pretrained_cad_model=False, nhead=4, dropout=0.1, normalize=False, device=None, encoder="vit",
                 num_views=0, window_size=1, compute_dtype: str = "f16", **kwargs):
        super().__init__()
        assert window_size > 0, "Window size must be greater than 0"          # reference :52
        if encoder != "vit" or use_pretrained_cad_model:
            raise NotImplementedError(f"encoder={encoder!r}/gencad is out of scope of the MI355X path (needs torchvision weights; "
                                      "reference model/trajectory_model.py:68-74)")
        if num_views and enable_past_states and not enable_past_actions:
            raise NotImplementedError("num_views > 0 with past states but no past actions: the reference's image_projection fan-in (:69-76) does not match "
                                      "the inputs it concatenates (:158-170) in that wiring")
        self.state_dim, self.act_dim, self.hidden_size = state_dim, act_dim, hidden_size
        self.max_length, self.max_ep_len = max_length, max_ep_len
        self.enable_past_actions, self.enable_past_states = enable_past_actions, enable_past_states
        self.enable_timestep_embedding = enable_timestep_embedding
        self.window_size, self.normalize, self.num_views = window_size, normalize, num_views
        self.use_pretrained_cad_model = use_pretrained_cad_model
        self.num_inputs = 1 + (1 if enable_past_states else 0) + (1 if num_views > 0 else 0)      # reference :67-76
        self.state_embedding_model_size = self.cad_embedding_model_size = 512
        self.dropout_p = dropout
        self.compute_dtype = compute_dtype
        if compute_dtype not in ("bf16", "f16", "fp16", "f32", "fp32", "bf16x3"):
            raise ValueError(f"compute_dtype={compute_dtype!r}: expected 'bf16', 'f16', 'bf16x3' or 'f32'")
        dt = {"bf16": L.VCAD_BF16, "bf16x3": L.VCAD_BF16X3, "f16": L.VCAD_F16, "fp16": L.VCAD_F16}.get(compute_dtype, L.VCAD_F32)
        cfg = make_config(hidden_size=hidden_size, nhead=nhead, num_decoder_layers=num_decoder_layers,
                          dim_feedforward=dim_feedforward, window_size=window_size, act_dim=act_dim, num_classes=num_classes,
                          num_params=num_params, num_params_values=num_params_values, max_ep_len=max_ep_len, dtype=dt,
                          enable_past_actions=enable_past_actions, enable_past_states=enable_past_states,
                          enable_timestep_embedding=enable_timestep_embedding, num_views=num_views,
                          **{k: kwargs[k] for k in ("vit_depth",) if k in kwargs})    # extension (tests): shallower ViT
        self._engine = NativeEngine(cfg, "cpu")
        self._param_names = list(self._engine.table.keys())
        for name in self._param_names:
            _attach(self, name, nn.Parameter(self._engine.view(name), requires_grad=True))
        self._plist = [dict(self.named_parameters())[n] for n in self._param_names]
        self._shadow_fresh = False
        self._drop_step = 0
        self._drop_rank = 0                                                  # data-parallel rank (set by the trainer): every rank draws its own masks
        self.reset_parameters()
        self.action_mask = torch.tensor([[1, 1, 0, 0, 0, 0], [0, 0, 1, 1, 0, 0], [0, 0, 0, 0, 1, 0],
                                         [0, 0, 0, 0, 0, 1], [0, 0, 0, 0, 0, 0]]).float()   # reference :83-89 (plain attribute)
        if device is not None:
            self.to(device)

    # ------------------------------------------------------------------ parameters
    @torch.no_grad()
    def reset_parameters(self):
        """PyTorch-default initialisation of the reference modules (SURVEY.md Appendix A 'Init')."""
        import math
        for name, p in self.named_parameters():
            leaf = name.rsplit(".", 1)[-1]
            if name.endswith("pos_embedding") or name.endswith("cls_token") or name.startswith("timestep_embedding"):
                p.normal_(0.0, 1.0)
            elif p.dim() == 1 and ("norm" in name or "to_patch_embedding.1." in name or "to_patch_embedding.3." in name or ".net.0." in name):
                p.fill_(1.0 if leaf == "weight" else 0.0)
            elif name.endswith("in_proj_weight"):
                nn.init.xavier_uniform_(p)
            elif name.endswith("in_proj_bias") or name.endswith("out_proj.bias"):
                p.zero_()
            elif p.dim() == 2:
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
            else:   # Linear bias: U(+-1/sqrt(fan_in)) with fan_in of the matching weight
                w = dict(self.named_parameters()).get(name[: -len("bias")] + "weight")
                bound = 1.0 / math.sqrt(w.shape[1]) if w is not None and w.dim() == 2 else 0.02
                p.uniform_(-bound, bound)
        self._shadow_fresh = False

    def _apply(self, fn, recurse=True):
        """`.to(device)` / `.cuda()`: move the flat buffer once and re-point every parameter at its view."""
        probe = fn(torch.empty(0, dtype=torch.float32, device=self._engine.params.device))
        if probe.dtype != torch.float32:
            raise TypeError("master weights stay fp32 (the compute dtype is chosen with compute_dtype=...)")
        if probe.device != self._engine.params.device:
            self._engine.allocate(probe.device, params=self._engine.params)
            for name in self._param_names:
                mod, leaf = self, name
                parts = name.split(".")
                for part in parts[:-1]:
                    mod = getattr(mod, part)
                p = mod._parameters[parts[-1]]
                p.data = self._engine.view(name)
                p.grad = None
            self._shadow_fresh = False
        self.action_mask = fn(self.action_mask)
        return self

    @staticmethod
    def _check_vit_layout(state_dict):
        """The ViT towers come from `vit-pytorch`, which the reference neither vendors nor pins (requirements.txt:8).  This engine implements the >= 1.2
        layout (LayerNorm inside `to_patch_embedding`, `Attention.norm`, `FeedForward.net.0` = LayerNorm, a final `transformer.norm`); the older PreNorm
        layout (`transformer.layers.N.K.fn.*` / `.norm.*`, `to_patch_embedding.1` = the Linear, no final norm) is DIFFERENT ARITHMETIC, not just other key
        names.  The reference's factory loads with strict=False (model/model_factory.py:35), which would silently leave both towers at their initial
        weights for such a checkpoint: refuse loudly instead."""
        for tower in ("state_embedding_model.", "cad_embedding_model."):
            keys = [k for k in state_dict if k.startswith(tower)]
            if not keys:
                continue
            import re
            legacy = [k for k in keys if ".fn." in k or re.match(re.escape(tower) + r"transformer\.layers\.\d+\.1\.norm\.", k)]     # PreNorm(fn=...) wrappers / PreNorm around the FeedForward
            v = state_dict.get(tower + "to_patch_embedding.1.weight")
            if legacy or (v is not None and getattr(v, "dim", lambda: 1)() == 2):
                raise RuntimeError(f"checkpoint keys such as '{(legacy or [tower + 'to_patch_embedding.1.weight'])[0]}' belong to the pre-1.2 vit-pytorch layout (PreNorm wrappers, no LayerNorm in "
                                   "the patch embedding, no final transformer.norm): different arithmetic from the layout this engine implements (DESIGN.md §2, ViT caveat) — "
                                   "the weights cannot be loaded; with strict=False they would have been silently ignored")

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        self._check_vit_layout(state_dict)
        own = set(self._param_names)
        filtered = {k: v for k, v in state_dict.items() if k in own}
        missing = [k for k in own if k not in state_dict]
        unexpected = [k for k in state_dict if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]}, unexpected {unexpected[:5]} "
                               "(the reference's dead GPT-2 parameters are not materialised; use strict=False)")
        with torch.no_grad():
            for k, v in filtered.items():
                self._engine.view(k).copy_(v.to(self._engine.params.device, torch.float32))
        self._shadow_fresh = False
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def _arm_dropout(self):
        """train(): p = dropout with a fresh seed per forward (the backward reuses it); eval(): off."""
        if self.training and self.dropout_p > 0:
            self._drop_step += 1
            self._engine.set_dropout(self.dropout_p, seed=(self._drop_seed_base << 44) | ((self._drop_rank & 0xFFF) << 32) | (self._drop_step & 0xFFFFFFFF))
        else:
            self._engine.set_dropout(0.0, 0)

    _drop_seed_base = 0x5EED

    def mark_shadow_fresh(self):
        """The native optimiser step rewrites the bf16 weight shadow itself; anything else (torch optimisers,
        load_state_dict, manual edits) leaves it stale, so forward re-casts it unless this was called."""
        self._shadow_fresh = True

    # ------------------------------------------------------------------ reference helpers
    def apply_action_mask(self, cmd_pred, param_pred):
        """reference model/autoregressive_transformer.py:91-108 (cmd_pred [B,T] int, param_pred [B,T,6])"""
        mask = self.action_mask.to(cmd_pred.device)[cmd_pred]
        masked = param_pred.clone()
        masked[mask == 0] = -1
        masked[:, :, 3] = torch.where((masked[:, :, 2] >= 200) & (masked[:, :, 2] < 250), masked[:, :, 3], -1)
        return masked

    def process_actions(self, actions):
        raise NotImplementedError("fused into the native forward (embed_action kernel)")

    def normalize_actions(self, actions):
        """in-place, like the reference (:115-118)"""
        actions[:, :, 0] = actions[:, :, 0] / 4.0
        actions[:, :, 1:] = actions[:, :, 1:] / 1000.0
        return actions

    # ------------------------------------------------------------------ forward
    def forward(self, inputs, attention_mask=None):
        """inputs: dict with 'frames' [B,T,1,224,224], 'actions' [B,T,7] (normalised), 'cad_image' [B,1,224,224]
        ('timesteps' is ignored exactly as in the reference, :144).  Returns (cmds [B,T,5], params [B,T,6,1000])."""
        frames, actions, cad = inputs["frames"], inputs["actions"], inputs["cad_image"]
        mv = inputs.get("multiview_images", None) if self.num_views > 0 else None      # reference :141,167: used only when both are set
        if self.num_views > 0 and mv is None:
            raise RuntimeError(f"model built with num_views = {self.num_views}: inputs['multiview_images'] [B, V, 1, S, S] is required "
                               "(the reference's image_projection would fail on the missing input)")
        # (a model that still lives in host memory: the library refuses — "no CPU fallback" — before any kernel runs, vcad_forward / vcad_infer_begin)
        if not self._shadow_fresh:
            self._engine.sync_shadow()
        self._shadow_fresh = False
        self._arm_dropout()
        if frames.dtype != torch.uint8:                          # uint8 pixel batches are normalised inside the patchify kernel
            frames = frames.float(); cad = cad.float()
            mv = mv.float() if mv is not None else None
        actions = actions.float()
        if torch.is_grad_enabled():
            cmds, pars = _EngineFn.apply(self, frames, actions, cad, mv, *self._plist)
        else:
            cmds, pars = self._engine.forward(frames, actions, cad, mv)
        return cmds, pars

    @torch.no_grad()
    def sequential_inference(self, ui_images, cad_image, action=False, cached=True):
        """reference :222-275: step-by-step prediction, (cmds [B,T,5], params [B,T,6,1000]).

        The reference re-runs `forward` on the prefix [0..t] for every t (t+1 ViT passes per step).  The model is causal, so with
        `cached=True` (default) each step encodes one new frame per clip and attends to cached keys / values (vcad_infer_begin /
        vcad_infer_step): same numbers, O(T) instead of O(T^2) frames.  `cached=False` keeps the reference's literal loop.
        action=True feeds back the masked arg-max of the previous step (the reference's version of that branch indexes a [B,6]
        tensor with [:, :, 3] and raises; here the time axis is kept so it runs)."""
        B, T = ui_images.shape[:2]
        device = ui_images.device
        was_training = self.training
        self.eval()
        try:
            if cached and T <= 1024:
                return self._sequential_cached(ui_images, cad_image, action)
            cmds_out, pars_out = [], []
            actions = torch.zeros(B, 1, self.act_dim, device=device) if action else None
            for t in range(T):
                inputs = {"frames": ui_images[:, : t + 1], "actions": actions if action else torch.zeros(B, t + 1, 7, device=device),
                          "timesteps": torch.arange(t + 1, device=device), "cad_image": cad_image}
                cmd, params = self.forward(inputs)
                if action:
                    actions = torch.cat([actions, self._next_action(cmd[:, -1:], params[:, -1:])], dim=1)
                cmds_out.append(cmd[:, -1].clone()); pars_out.append(params[:, -1].clone())
            return torch.stack(cmds_out, dim=1), torch.stack(pars_out, dim=1)
        finally:
            self.train(was_training)

    def _next_action(self, cmd, params):
        """[B,1,5], [B,1,6,1000] logits -> the normalised action [B,1,7] fed at the next position (reference :249-263)"""
        cmd_pred = torch.argmax(cmd, dim=-1)                                  # [B,1]
        nxt = self.apply_action_mask(cmd_pred, torch.argmax(params, dim=-1)).float()
        return self.normalize_actions(torch.cat([cmd_pred.unsqueeze(-1).float(), nxt], dim=2))

    def _sequential_cached(self, ui_images, cad_image, action):
        eng = self._engine
        if not self._shadow_fresh:
            eng.sync_shadow()
        B, T = ui_images.shape[:2]
        if ui_images.dtype != torch.uint8:
            ui_images = ui_images.float(); cad_image = cad_image.float()
        eng.infer_begin(cad_image, B, T)
        cmds = torch.empty(B, T, 5, device=ui_images.device); pars = torch.empty(B, T, 6, self._engine.cfg.num_params_values, device=ui_images.device)
        a = torch.zeros(B, self.act_dim, device=ui_images.device)
        for t in range(T):
            c, p = eng.infer_step(t, ui_images[:, t], a)
            cmds[:, t] = c; pars[:, t] = p
            if action:
                a = self._next_action(c.unsqueeze(1), p.unsqueeze(1))[:, 0]
        return cmds, pars
