"""Thin Python handle on a `vcad_engine` (include/vcad.h): owns the flat fp32 parameter / gradient / Adam buffers
and the activation workspace as torch tensors (PyTorch = device memory + streams only) and forwards every call
to the C ABI on torch's current HIP stream."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import lib as L

VIT_DEFAULTS = dict(vit_dim=512, vit_depth=6, vit_heads=16, vit_dim_head=64, vit_mlp=512, image_size=224, patch_size=32)
# ^ the ViT(...) call at reference model/trajectory_model.py:54-65


def make_config(hidden_size, nhead=4, num_decoder_layers=8, dim_feedforward=512, window_size=1, act_dim=7, num_classes=5,
                num_params=6, num_params_values=1000, max_ep_len=1000, dtype=L.VCAD_F32, enable_past_actions=True,
                enable_past_states=True, enable_timestep_embedding=True, num_views=0, **vit) -> L.Config:
    v = dict(VIT_DEFAULTS); v.update({k: vit[k] for k in vit if k in VIT_DEFAULTS})
    return L.Config(hidden_size=hidden_size, nhead=nhead, num_decoder_layers=num_decoder_layers, dim_feedforward=dim_feedforward,
                    window_size=window_size, act_dim=act_dim, num_classes=num_classes, num_params=num_params,
                    num_params_values=num_params_values, max_ep_len=max_ep_len, dtype=dtype,
                    enable_past_actions=int(bool(enable_past_actions)), enable_past_states=int(bool(enable_past_states)),
                    enable_timestep_embedding=int(bool(enable_timestep_embedding)), num_views=int(num_views), **v)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class NativeEngine:
    def __init__(self, cfg: L.Config, device):
        self.lib = L.load(L.storage_format(cfg.dtype))
        self.cfg = cfg
        self.device = torch.device(device)
        h = C.c_void_p()
        L.check(self.lib, self.lib.vcad_engine_create(C.byref(cfg), C.byref(h)), "engine_create")
        self.h = h
        self.total = int(self.lib.vcad_param_total(h))
        self.table: Dict[str, tuple] = {}
        name = C.create_string_buffer(256)
        for i in range(self.lib.vcad_param_count(h)):
            off, numel, nd = C.c_int64(), C.c_int64(), C.c_int()
            shape = (C.c_int64 * 4)()
            L.check(self.lib, self.lib.vcad_param_info(h, i, name, 256, C.byref(off), C.byref(numel), C.byref(shape), C.byref(nd)), "param_info")
            self.table[name.value.decode()] = (off.value, numel.value, tuple(shape[k] for k in range(nd.value)))
        self.buckets = []
        for b in range(self.lib.vcad_bucket_count(h)):
            lo, hi = C.c_int64(), C.c_int64()
            L.check(self.lib, self.lib.vcad_bucket_range(h, b, C.byref(lo), C.byref(hi)), "bucket_range")
            self.buckets.append((lo.value, hi.value))
        self.side_stage = int(self.lib.vcad_side_stage(h))      # the backward stage (CAD ViT) that may run on the library's side stream
        self.params = self.grads = self.m = self.v = self.shadow = None
        self.ws = None
        self.step_count = 0
        self.gemm_flags = 0           # kernel-selection flags of this engine's GEMMs (lib.GEMM_*; the data-parallel trainer sets GEMM_DYNAMIC)
        self.fwd_serial = 0           # bumped by every call that overwrites the engine's single activation set (forward, infer_begin)
        self.fp8 = False
        self.allocate(self.device)
        if os.environ.get("VCAD_FP8", "0") == "1" and self.cfg.dtype == L.VCAD_BF16:
            self.set_fp8(True)

    # ------------------------------------------------------------------ buffers
    def allocate(self, device, params: Optional[torch.Tensor] = None):
        self.device = torch.device(device)
        z = lambda dt=torch.float32: torch.zeros(self.total, dtype=dt, device=self.device)
        self.params = params.to(self.device) if params is not None else z()
        self.grads, self.m, self.v = z(), z(), z()
        # bf16: bf16 copy of the weights; bf16x3: the weights pre-split into hi | lo bf16 words (same 4 bytes per element); f32: none
        self.shadow = {L.VCAD_BF16: lambda: z(torch.bfloat16), L.VCAD_F16: lambda: z(torch.float16), L.VCAD_BF16X3: lambda: z(torch.int32)}.get(self.cfg.dtype, lambda: None)()
        self.ws = None
        self._bind()

    def _bind(self):
        L.check(self.lib, self.lib.vcad_bind(self.h, _ptr(self.params), _ptr(self.grads), _ptr(self.m), _ptr(self.v), _ptr(self.shadow)), "bind")
        if self.ws is not None:
            L.check(self.lib, self.lib.vcad_set_workspace(self.h, _ptr(self.ws), self.ws.numel()), "set_workspace")

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream) if self.device.type == "cuda" else None

    def view(self, name: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        off, numel, shape = self.table[name]
        return (self.params if buf is None else buf)[off: off + numel].view(shape)

    def sync_shadow(self):
        L.check(self.lib, self.lib.vcad_sync_shadow(self.h, self.stream()), "sync_shadow")

    def ensure_workspace(self, B: int, T: int):
        need = int(self.lib.vcad_workspace_bytes(self.h, B, T))
        if self.ws is None or self.ws.numel() < need:
            self.ws = None
            self.ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            L.check(self.lib, self.lib.vcad_set_workspace(self.h, _ptr(self.ws), need), "set_workspace")

    def set_fp8(self, on: bool = True):
        """VCAD_FP8 forward mode (include/vcad.h: vcad_set_fp8): the ViT's Linear layers on the block-scaled fp8 matrix cores; bf16 engines
        only, backward unchanged.  The workspace is re-planned on the next forward."""
        L.check(self.lib, self.lib.vcad_set_fp8(self.h, 1 if on else 0), "set_fp8")
        self.fp8 = bool(on)

    # ------------------------------------------------------------------ gradient scale (fp16 engines; include/vcad.h: vcad_set_grad_scale)
    @property
    def grad_scale(self) -> float:
        return float(self.lib.vcad_grad_scale(self.h))

    def set_grad_scale(self, scale: float):
        """a power of two, or 0 = automatic (fp16 engines' default: follows the planned batch, 4096 at 32 x 64); the backward runs on scaled gradients, the
        gradient buffer holds true ones.  Re-plans the workspace on the next forward."""
        L.check(self.lib, self.lib.vcad_set_grad_scale(self.h, float(scale)), "set_grad_scale")

    def set_defer_unscale(self, on: bool):
        """fp16 engines, the native train step only (include/vcad.h: vcad_set_defer_unscale): gradients stay multiplied by the gradient scale until the optimiser divides
        (exactly) in its norm pass and in Adam; the loss writes scaled dlogits straight into the backward's copy.  Nothing may read gradients or dlogits in between."""
        L.check(self.lib, self.lib.vcad_set_defer_unscale(self.h, 1 if on else 0), "set_defer_unscale")

    GROW_AFTER = 2000                 # finite steps in a row before a lowered scale is doubled again (torch.amp.GradScaler's growth_interval)

    def note_overflows(self, bad: int, steps: int, ran_at_scale: Optional[float] = None) -> bool:
        """fp16 engines, host logic of the gradient scale (no kernels).  `bad` of the last `steps` optimizer steps came back with a non-finite
        gradient norm — the scaled backward overflowed fp16 and the library skipped those updates.  Then: the Adam step counter is taken back
        by `bad` (a skipped update must not advance the bias correction), the scale is halved once (explicit mode from here on) and the value
        the automatic rule had chosen is remembered as the target.  Without an overflow the scale is only ever RAISED towards that target —
        an engine that never overflowed stays in automatic mode at what `2 x pow2ceil(B T), floor 1024` gives it (r04 grew every
        engine to 4096) — doubling after GROW_AFTER finite steps in a row; on reaching the target it returns to automatic mode, so that later
        batch shapes adapt again."""
        if self.cfg.dtype != L.VCAD_F16:
            return False
        if bad > 0:
            self._good_norms = 0
            self.step_count = max(0, self.step_count - int(bad))
            self.skipped_steps = getattr(self, "skipped_steps", 0) + int(bad)
            #  the host reads a window's counter one window late: with a persistent overflow the window AFTER the one that triggered a halving also
            # ran at the old scale and is non-finite too — its skipped updates are booked above, but it says nothing about the scale in force now: no second halving
            if ran_at_scale is not None and ran_at_scale > self.grad_scale:
                return False
            if getattr(self, "_scale_target", None) is None:
                self._scale_target = self.grad_scale                 # (what the automatic rule chose for this batch shape)
            if self.grad_scale > 1.0:
                self.set_grad_scale(self.grad_scale / 2)             # explicit from here on
            return True
        target = getattr(self, "_scale_target", None)
        if target is None:                                           # never lowered: nothing to grow back to
            return False
        self._good_norms = getattr(self, "_good_norms", 0) + int(steps)
        if self._good_norms >= self.GROW_AFTER:
            self._good_norms = 0
            if self.grad_scale * 2 >= target:
                self.set_grad_scale(0.0); self._scale_target = None  # back at the automatic rule's value: automatic mode again
            else:
                self.set_grad_scale(self.grad_scale * 2)
        return False

    def check_grad_overflow(self, norm: torch.Tensor) -> bool:
        """one norm (what optimizer_step returned some steps ago) through note_overflows.  Host sync on `norm`: the trainer does not call this per
        step — it counts non-finite norms on the device and reads one counter per 32 steps (trainer.BaseTrainer.train_step)."""
        if self.cfg.dtype != L.VCAD_F16 or norm is None:
            return False
        return self.note_overflows(0 if bool(torch.isfinite(norm[0]).item()) else 1, 1)

    def reset_overflow_history(self):
        """after the parameters were replaced (checkpoint load, restore_best_weights): norms of the previous weights say nothing about the new ones"""
        self._good_norms = 0

    def set_dropout(self, p: float, seed: int = 0):
        """p = 0 disables; call with a fresh seed before every training forward (masks = hash(seed, site, index))."""
        L.check(self.lib, self.lib.vcad_set_dropout(self.h, float(p), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF)), "set_dropout")

    def dropout_mask(self, module: int, layer: int, kind: int, n: int, first: int = 0) -> torch.Tensor:
        """keep-multipliers of elements first .. first + n - 1 of one dropout site (tests: the oracle applies them as explicit multipliers)"""
        out = torch.empty(n, dtype=torch.float32)
        if first:
            L.check(self.lib, self.lib.vcad_dropout_mask_range(self.h, module, layer, kind, first, n, C.c_void_p(out.data_ptr())), "dropout_mask_range")
        else:
            L.check(self.lib, self.lib.vcad_dropout_mask(self.h, module, layer, kind, n, C.c_void_p(out.data_ptr())), "dropout_mask")
        return out

    # kernel-selection flags (tests: small batches on the kernels the C2 shapes take; lib.GEMM_*) / launches per kernel family / side stream
    def set_gemm_flags(self, flags: int):
        self.gemm_flags = int(flags)
        L.check(self.lib, self.lib.vcad_set_gemm_flags(self.h, int(flags)), "set_gemm_flags")

    def set_dynamic_items(self, on: bool = True):
        """persistent GEMM draws its items with tickets (robust while communication kernels hold CUs): the data-parallel trainer turns it on"""
        self.set_gemm_flags((self.gemm_flags & ~L.GEMM_DYNAMIC) | (L.GEMM_DYNAMIC if on else 0))

    def kernel_launches(self, family: int) -> int:
        return int(self.lib.vcad_kernel_launches(self.h, int(family)))

    def set_side_stream(self, on: bool):
        L.check(self.lib, self.lib.vcad_set_side_stream(self.h, int(bool(on))), "set_side_stream")

    # ------------------------------------------------------------------ hot path
    def forward(self, frames: torch.Tensor, actions_norm: torch.Tensor, cad: torch.Tensor, multiview: Optional[torch.Tensor] = None):
        """frames [B,T,1,S,S] (any batch stride, frames contiguous within a clip), actions_norm [B,T,7], cad [B,1,S,S].
        frames / cad are either fp32 (already normalised, the reference loader's contract) or BOTH uint8 grayscale pixels
        (normalised inside the patchify kernel: vcad_forward_u8).  multiview (engines with num_views > 0): [B,V,1,S,S] in the CAD
        image's pixel format (reference model/autoregressive_transformer.py:131,167-170)."""
        B, T = int(actions_norm.shape[0]), int(actions_norm.shape[1])
        S = self.cfg.image_size
        u8 = frames.dtype == torch.uint8
        rgb = u8 and frames.dim() == 5 and tuple(frames.shape[2:]) == (S, S, 3)       # the dataset's stored pixels [B,T,H,W,3]: gray + normalise in-kernel
        assert frames.dtype in (torch.float32, torch.uint8) and frames.shape[1] == T and (rgb or tuple(frames.shape[2:]) == (1, S, S))
        if rgb:
            if frames.stride()[1:] != (S * S * 3, S * 3, 3, 1):
                frames = frames.contiguous()
            fb = (frames.stride(0) if B > 1 else T * S * S * 3) // 3                  # batch stride in pixels
            assert fb % 4 == 0
        else:
            if frames.stride()[1:] != (S * S, S * S, S, 1):
                frames = frames.contiguous()
            fb = frames.stride(0) if B > 1 else T * S * S
        actions_norm = actions_norm.contiguous().float()
        cad = cad.contiguous() if u8 else cad.contiguous().float()
        assert cad.dtype == frames.dtype, "frames and cad_image must both be fp32 or both be uint8"
        self.ensure_workspace(B, T)
        cmds = torch.empty(B, T, self.cfg.num_classes, device=self.device)
        pars = torch.empty(B, T, self.cfg.num_params, self.cfg.num_params_values, device=self.device)
        V = self.cfg.num_views
        if V > 0:
            if multiview is None:
                raise RuntimeError(f"this model was built with num_views = {V}: inputs['multiview_images'] is required")
            assert multiview.dtype == cad.dtype and multiview.numel() == B * V * S * S, (multiview.dtype, tuple(multiview.shape))
            multiview = multiview.contiguous()
            L.check(self.lib, self.lib.vcad_set_multiview(self.h, _ptr(multiview)), "set_multiview")
        self._keep = (frames, actions_norm, cad, multiview)          # backward re-reads the inputs (patch-LN grads, embed_action wgrad)
        self.fwd_serial += 1
        fn = self.lib.vcad_forward_rgb8 if rgb else (self.lib.vcad_forward_u8 if u8 else self.lib.vcad_forward)
        L.check(self.lib, fn(self.h, _ptr(frames), fb, _ptr(actions_norm), _ptr(cad), B, T, _ptr(cmds), _ptr(pars), self.stream()), "forward")
        return cmds, pars

    # ------------------------------------------------------------------ incremental inference (include/vcad.h: vcad_infer_*)
    def infer_begin(self, cad: torch.Tensor, B: int, Tmax: int):
        need = max(int(self.lib.vcad_infer_workspace_bytes(self.h, B, Tmax)), self.ws.numel() if self.ws is not None else 0)
        if self.ws is None or self.ws.numel() < need:
            self.ws = None
            self.ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            L.check(self.lib, self.lib.vcad_set_workspace(self.h, _ptr(self.ws), need), "set_workspace")
        u8 = cad.dtype == torch.uint8
        cad = cad.contiguous() if u8 else cad.contiguous().float()
        self._keep_i = cad
        self.fwd_serial += 1                                 # the (B, 1) incremental plan replaces the training activations
        fn = self.lib.vcad_infer_begin_u8 if u8 else self.lib.vcad_infer_begin
        L.check(self.lib, fn(self.h, _ptr(cad), B, Tmax, self.stream()), "infer_begin")
        self._infer_u8 = u8

    def infer_step(self, t: int, frame: Optional[torch.Tensor], action_norm: Optional[torch.Tensor]):
        """frame [B,1,S,S] (view of clip frames is fine: only the batch stride matters), action_norm [B,7] -> (cmds [B,5], params [B,6,1000])"""
        S = self.cfg.image_size
        B = int(frame.shape[0]) if frame is not None else int(action_norm.shape[0])
        fb = 0
        if frame is not None:
            assert (frame.dtype == torch.uint8) == self._infer_u8
            if not self._infer_u8:
                frame = frame.float()
            if frame.stride()[1:] != (S * S, S, 1):
                frame = frame.contiguous()
            fb = frame.stride(0) if B > 1 else S * S
        if action_norm is not None:
            action_norm = action_norm.contiguous().float()
        cmds = torch.empty(B, self.cfg.num_classes, device=self.device)
        pars = torch.empty(B, self.cfg.num_params, self.cfg.num_params_values, device=self.device)
        self._keep_s = (frame, action_norm)
        L.check(self.lib, self.lib.vcad_infer_step(self.h, t, _ptr(frame), fb, _ptr(action_norm), _ptr(cmds), _ptr(pars), self.stream()), "infer_step")
        return cmds, pars

    def loss(self, cmds, pars, targets, label_weights, use_mse=True, class_weights: Optional[torch.Tensor] = None):
        """label_weights: the 5 floats of class_weights.json["Label"] (host; the trainer reads the file like reference trainer.py:822)."""
        B, T = cmds.shape[0], cmds.shape[1]
        targets = targets.reshape(B * T, 7).contiguous().float()
        out = torch.empty(8, device=self.device); met = torch.empty(L.NMETRIC, dtype=torch.int32, device=self.device)
        lw = (C.c_float * 5)(*[float(x) for x in label_weights])
        L.check(self.lib, self.lib.vcad_loss(self.h, _ptr(cmds), _ptr(pars), _ptr(targets), B, T, int(use_mse), C.byref(lw), _ptr(class_weights),
                                             _ptr(out), _ptr(met), self.stream()), "loss")
        self._keep_t = targets
        return out, met

    def dl_views(self, B: int, T: int):
        """(dcmds [B,T,5], dparams [B,T,6000]) fp32 views of the workspace region vcad_loss filled."""
        oc, op = C.c_size_t(), C.c_size_t()
        L.check(self.lib, self.lib.vcad_dlogits_offsets(self.h, C.byref(oc), C.byref(op)), "dlogits_offsets")
        nc, npar = self.cfg.num_classes, self.cfg.num_params * self.cfg.num_params_values
        dc = self.ws[oc.value: oc.value + B * T * nc * 4].view(torch.float32).view(B, T, nc)
        dp = self.ws[op.value: op.value + B * T * npar * 4].view(torch.float32).view(B, T, npar)
        return dc, dp

    def backward(self, dcmds=None, dpars=None, stage: Optional[int] = None, side: bool = False):
        """stage=None: whole backward.  stage=k: one gradient bucket's worth (data-parallel callers).  side=True (stage `side_stage` only): launch on
        the library's side stream — call join_side() before touching that bucket's gradients."""
        if dcmds is not None:
            dcmds = dcmds.contiguous().float(); dpars = dpars.contiguous().float()
        if side:
            L.check(self.lib, self.lib.vcad_backward_stage_side(self.h, stage, _ptr(dcmds), _ptr(dpars), self.stream()), "backward_stage_side")
        elif stage is None:
            L.check(self.lib, self.lib.vcad_backward(self.h, _ptr(dcmds), _ptr(dpars), self.stream()), "backward")
        else:
            L.check(self.lib, self.lib.vcad_backward_stage(self.h, stage, _ptr(dcmds), _ptr(dpars), self.stream()), "backward_stage")

    def join_side(self):
        L.check(self.lib, self.lib.vcad_join_side(self.h, self.stream()), "join_side")

    # ------------------------------------------------------------------ half-precision wire format of the gradient exchange (include/vcad.h: vcad_wire_*)
    @property
    def wire_dtype(self) -> torch.dtype:
        """the 16-bit format this engine's library packs gradient buckets into: its storage format (bf16, or fp16 for VCAD_F16 engines)"""
        return torch.float16 if self.lib.vcad_storage_format() == b"f16" else torch.bfloat16

    def wire_amax(self, lo: int, hi: int, amax: torch.Tensor):
        """amax: fp32 [1 + 1024] on the engine's device; amax[0] <- max |g| over grads[lo:hi] (the caller all-reduces it with MAX)"""
        assert amax.dtype == torch.float32 and amax.numel() >= 1025 and amax.is_contiguous()
        L.check(self.lib, self.lib.vcad_wire_amax(self.h, lo, hi, _ptr(amax), self.stream()), "wire_amax")

    def wire_pack(self, lo: int, hi: int, wire: torch.Tensor, amax: Optional[torch.Tensor], world: int):
        assert wire.dtype == self.wire_dtype and wire.numel() >= hi - lo and wire.is_contiguous()
        L.check(self.lib, self.lib.vcad_wire_pack(self.h, lo, hi, _ptr(wire), _ptr(amax), int(world), self.stream()), "wire_pack")

    def wire_unpack(self, lo: int, hi: int, wire: torch.Tensor, amax: Optional[torch.Tensor], world: int):
        assert wire.dtype == self.wire_dtype and wire.numel() >= hi - lo and wire.is_contiguous()
        L.check(self.lib, self.lib.vcad_wire_unpack(self.h, lo, hi, _ptr(wire), _ptr(amax), int(world), self.stream()), "wire_unpack")

    def optimizer_step(self, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0, grad_scale=1.0):
        """lr: one float, or one float per gradient bucket (the reference's `frozen` parameter groups)."""
        self.step_count += 1
        if self.ws is None:
            self.ensure_workspace(1, 1)
        norm = torch.empty(2, device=self.device)
        if isinstance(lr, (list, tuple)):
            assert len(lr) == len(self.buckets)
            lrs = (C.c_float * len(lr))(*[float(x) for x in lr])
            L.check(self.lib, self.lib.vcad_optimizer_step_groups(self.h, lrs, betas[0], betas[1], eps, max_norm, self.step_count, grad_scale,
                                                                  _ptr(norm), self.stream()), "optimizer_step")
        else:
            L.check(self.lib, self.lib.vcad_optimizer_step(self.h, lr, betas[0], betas[1], eps, max_norm, self.step_count, grad_scale,
                                                           _ptr(norm), self.stream()), "optimizer_step")
        return norm

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.vcad_engine_destroy(self.h)
                self.h = None
        except Exception:
            pass
