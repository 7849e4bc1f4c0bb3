"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (`videocad_amd/`).

CPU fp32 restatement (PyTorch-CPU functional ops + autograd) of the reference's
behaviour-cloning training step.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module, and only as the checker /
reported CPU baseline — never as the thing measured or shipped.

What it restates (reference file:line, all relative to /root/reference):
  * AutoRegressiveTransformer.forward ........ model/autoregressive_transformer.py:121-220
  * ViT frame / CAD encoder (constructor) .... model/trajectory_model.py:52-67, 90-100
      arithmetic = third-party `vit-pytorch` (requirements.txt:8, UNPINNED, not vendored):
      restated here from its published vit.py (>=1.2 layout).
  * embed_state / embed_image ................ model/base_transformer.py:53-54
  * nn.TransformerDecoder (post-norm, ReLU) .. constructed model/autoregressive_transformer.py:54-62
  * _prepare_model_inputs / normalize_actions  trainer.py:507-517, 800-804
  * compute_loss / flexible_cross_entropy .... trainer.py:853-917, 935-1063
  * _process_batch (clip + Adam) ............. trainer.py:480-496

PINNING STATUS.  Validated in this container against the *imported* reference
(tests/golden/make_goldens.py, stubs for the four missing packages) and against the
committed golden vectors in tests/golden/*.npz.  The ViT arithmetic itself is pinned
only to the restated `vit_pytorch` (the reference neither vendors nor pins it and has no
tests): **ViT parity is unpinned upstream** — goldens pin the restatement.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

CANONICAL_CONFIG = dict(   # transformer_experiments.json -> cad_past_10_actions_and_states_timestep_embedding
    hidden_size=1024, nhead=4, num_decoder_layers=8, dim_feedforward=1024, window_size=10,
    act_dim=7, num_classes=5, num_params=6, num_params_values=1000, max_ep_len=1000,
    enable_past_actions=True, enable_past_states=True, enable_timestep_embedding=True,
    vit_dim=512, vit_depth=6, vit_heads=16, vit_dim_head=64, vit_mlp=512, image_size=224, patch_size=32,
)

TOLERANCE = 3                                   # trainer.py:20
TOLERANCES = [TOLERANCE - 1, TOLERANCE - 1, 50, 200, 500, TOLERANCE - 1]   # trainer.py:827
ABOVE = [False, False, True, True, True, False]                            # trainer.py:829
PARAM_TO_LABEL = [0, 0, 1, 1, 2, 3]                                        # trainer.py:825
LABEL_WEIGHTS = [0.04332685213392362, 0.02915898563179938, 0.267566828114559,
                 0.6005346809501417, 0.05941265316957628]                  # class_weights.json "Label"


# ----------------------------------------------------------------------------------------
# parameter inventory (live parameters only)
# ----------------------------------------------------------------------------------------

def param_shapes(cfg: dict = CANONICAL_CONFIG) -> Dict[str, Tuple[int, ...]]:
    H, D = cfg["hidden_size"], cfg["vit_dim"]
    inner = cfg["vit_heads"] * cfg["vit_dim_head"]
    pd = cfg["patch_size"] ** 2
    ntok = (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1
    s: Dict[str, Tuple[int, ...]] = {}
    for pre in ("state_embedding_model.", "cad_embedding_model."):
        s[pre + "pos_embedding"] = (1, ntok, D)
        s[pre + "cls_token"] = (1, 1, D)
        s[pre + "to_patch_embedding.1.weight"] = (pd,)
        s[pre + "to_patch_embedding.1.bias"] = (pd,)
        s[pre + "to_patch_embedding.2.weight"] = (D, pd)
        s[pre + "to_patch_embedding.2.bias"] = (D,)
        s[pre + "to_patch_embedding.3.weight"] = (D,)
        s[pre + "to_patch_embedding.3.bias"] = (D,)
        for L in range(cfg["vit_depth"]):
            a = f"{pre}transformer.layers.{L}.0."
            s[a + "norm.weight"] = (D,)
            s[a + "norm.bias"] = (D,)
            s[a + "to_qkv.weight"] = (3 * inner, D)
            s[a + "to_out.0.weight"] = (D, inner)
            s[a + "to_out.0.bias"] = (D,)
            f = f"{pre}transformer.layers.{L}.1.net."
            s[f + "0.weight"] = (D,)
            s[f + "0.bias"] = (D,)
            s[f + "1.weight"] = (cfg["vit_mlp"], D)
            s[f + "1.bias"] = (cfg["vit_mlp"],)
            s[f + "4.weight"] = (D, cfg["vit_mlp"])
            s[f + "4.bias"] = (D,)
        s[pre + "transformer.norm.weight"] = (D,)
        s[pre + "transformer.norm.bias"] = (D,)
    s["embed_state.weight"] = (H, D); s["embed_state.bias"] = (H,)
    s["embed_image.weight"] = (H, D); s["embed_image.bias"] = (H,)
    V = cfg.get("num_views", 0)
    num_inputs = 1 + (1 if cfg.get("enable_past_states", True) else 0) + (1 if V > 0 else 0)       # autoregressive_transformer.py:68-76
    s["image_projection.weight"] = (H, num_inputs * H); s["image_projection.bias"] = (H,)
    if V > 0:
        s["embed_multiview.weight"] = (H, D * V); s["embed_multiview.bias"] = (H,)               # :72-74
    s["embed_action.weight"] = (H, cfg["act_dim"]); s["embed_action.bias"] = (H,)
    if cfg.get("enable_timestep_embedding", True):
        s["timestep_embedding.weight"] = (cfg["max_ep_len"], H)
    ff = cfg["dim_feedforward"]
    for L in range(cfg["num_decoder_layers"]):
        p = f"transformer_decoder.layers.{L}."
        for att in ("self_attn.", "multihead_attn."):
            s[p + att + "in_proj_weight"] = (3 * H, H)
            s[p + att + "in_proj_bias"] = (3 * H,)
            s[p + att + "out_proj.weight"] = (H, H)
            s[p + att + "out_proj.bias"] = (H,)
        s[p + "linear1.weight"] = (ff, H); s[p + "linear1.bias"] = (ff,)
        s[p + "linear2.weight"] = (H, ff); s[p + "linear2.bias"] = (H,)
        for n in ("norm1.", "norm2.", "norm3."):
            s[p + n + "weight"] = (H,); s[p + n + "bias"] = (H,)
    s["predict_action_class_0_4.weight"] = (cfg["num_classes"], H)
    s["predict_action_class_0_4.bias"] = (cfg["num_classes"],)
    nv = cfg["num_params"] * cfg["num_params_values"]
    s["predict_action_class_0_999.weight"] = (nv, H)
    s["predict_action_class_0_999.bias"] = (nv,)
    return s


# ----------------------------------------------------------------------------------------
# ViT (restated vit-pytorch >=1.2; constructor args at model/trajectory_model.py:54-65)
# ----------------------------------------------------------------------------------------

def _ln(x: Tensor, P: dict, pre: str) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), P[pre + "weight"], P[pre + "bias"], 1e-5)


def _m(masks: Optional[dict], key: str, x: Tensor) -> Tensor:
    """explicit dropout: multiply by a supplied keep-multiplier tensor (0 or 1/(1-p)); no-op without masks (eval mode)."""
    return x * masks[key] if masks is not None and key in masks else x


def vit_forward(P: dict, pre: str, img: Tensor, cfg: dict = CANONICAL_CONFIG, taps: Optional[dict] = None,
                masks: Optional[dict] = None) -> Tensor:
    """img [N,1,S,S] -> cls embedding [N,D] (mlp_head = Identity, trajectory_model.py:66)."""
    N = img.shape[0]
    p = cfg["patch_size"]; g = cfg["image_size"] // p
    heads, dh = cfg["vit_heads"], cfg["vit_dim_head"]
    # Rearrange 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' with c = 1
    x = img.reshape(N, 1, g, p, g, p).permute(0, 2, 4, 3, 5, 1).reshape(N, g * g, p * p)
    x = _ln(x, P, pre + "to_patch_embedding.1.")
    x = F.linear(x, P[pre + "to_patch_embedding.2.weight"], P[pre + "to_patch_embedding.2.bias"])
    x = _ln(x, P, pre + "to_patch_embedding.3.")
    x = torch.cat([P[pre + "cls_token"].expand(N, 1, -1), x], dim=1)
    x = _m(masks, pre + "emb", x + P[pre + "pos_embedding"][:, : g * g + 1])          # emb_dropout
    if taps is not None:
        taps[pre + "embed"] = x
    for L in range(cfg["vit_depth"]):
        a = f"{pre}transformer.layers.{L}.0."
        h = _ln(x, P, a + "norm.")
        qkv = F.linear(h, P[a + "to_qkv.weight"])
        q, k, v = (t.reshape(N, -1, heads, dh).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        dots = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)
        attn = _m(masks, f"{pre}L{L}.attn", dots.softmax(dim=-1))
        out = torch.matmul(attn, v).transpose(1, 2).reshape(N, -1, heads * dh)
        x = _m(masks, f"{pre}L{L}.out", F.linear(out, P[a + "to_out.0.weight"], P[a + "to_out.0.bias"])) + x
        f = f"{pre}transformer.layers.{L}.1.net."
        h = _ln(x, P, f + "0.")
        h = _m(masks, f"{pre}L{L}.mlp_act", F.gelu(F.linear(h, P[f + "1.weight"], P[f + "1.bias"])))
        x = _m(masks, f"{pre}L{L}.mlp_out", F.linear(h, P[f + "4.weight"], P[f + "4.bias"])) + x
        if taps is not None:
            taps[f"{pre}layer{L}"] = x
    x = _ln(x, P, pre + "transformer.norm.")
    return x[:, 0]


# ----------------------------------------------------------------------------------------
# decoder (torch.nn.TransformerDecoderLayer defaults: post-norm, ReLU, eps 1e-5, batch_first=False)
# ----------------------------------------------------------------------------------------

def _mha(xq: Tensor, xkv: Tensor, P: dict, pre: str, nhead: int, mask: Tensor, masks: Optional[dict] = None, mkey: str = "") -> Tensor:
    """Batch-first [B,T,E] restatement of nn.MultiheadAttention with an additive [T,T] mask."""
    B, T, E = xq.shape
    d = E // nhead
    w, b = P[pre + "in_proj_weight"], P[pre + "in_proj_bias"]
    q = F.linear(xq, w[:E], b[:E]).reshape(B, T, nhead, d).transpose(1, 2)
    k = F.linear(xkv, w[E:2 * E], b[E:2 * E]).reshape(B, -1, nhead, d).transpose(1, 2)
    v = F.linear(xkv, w[2 * E:], b[2 * E:]).reshape(B, -1, nhead, d).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d) + mask
    o = torch.matmul(_m(masks, mkey, s.softmax(dim=-1)), v).transpose(1, 2).reshape(B, T, E)
    return F.linear(o, P[pre + "out_proj.weight"], P[pre + "out_proj.bias"])


def window_mask(T: int, window: int) -> Tensor:
    """0 where (j > i - window) & (j <= i), -inf elsewhere (autoregressive_transformer.py:182-188).
    window >= T gives the causal mask of generate_square_subsequent_mask (:180)."""
    i = torch.arange(T)[:, None]; j = torch.arange(T)[None, :]
    m = torch.full((T, T), float("-inf"))
    m[(j > i - window) & (j <= i)] = 0.0
    return m


def decoder_forward(P: dict, tgt: Tensor, mem: Tensor, cfg: dict, taps: Optional[dict] = None, masks: Optional[dict] = None,
                    tgt_band: bool = False) -> Tensor:
    B, T, E = tgt.shape
    band = window_mask(T, cfg["window_size"])
    causal = band if tgt_band else window_mask(T, T)
    x = tgt
    for L in range(cfg["num_decoder_layers"]):
        p = f"transformer_decoder.layers.{L}."
        x = _ln(x + _m(masks, f"dec{L}.sa_out", _mha(x, x, P, p + "self_attn.", cfg["nhead"], causal, masks, f"dec{L}.sa")), P, p + "norm1.")
        x = _ln(x + _m(masks, f"dec{L}.ca_out", _mha(x, mem, P, p + "multihead_attn.", cfg["nhead"], band, masks, f"dec{L}.ca")), P, p + "norm2.")
        ffn = F.linear(_m(masks, f"dec{L}.ff_act", F.relu(F.linear(x, P[p + "linear1.weight"], P[p + "linear1.bias"]))),
                       P[p + "linear2.weight"], P[p + "linear2.bias"])
        x = _ln(x + _m(masks, f"dec{L}.ff_out", ffn), P, p + "norm3.")
        if taps is not None:
            taps[f"dec{L}"] = x
    return x


# ----------------------------------------------------------------------------------------
# full forward (all wirings; the optional multiview branch included)
# ----------------------------------------------------------------------------------------

def model_forward(P: dict, frames: Tensor, actions_norm: Tensor, cad: Tensor,
                  cfg: dict = CANONICAL_CONFIG, taps: Optional[dict] = None, masks: Optional[dict] = None,
                  multiview: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """frames [B,T,1,S,S], actions_norm [B,T,7] (already normalised), cad [B,1,S,S]
    -> cmds [B,T,5], params [B,T,6,1000]   (autoregressive_transformer.py:121-220)."""
    B, T = actions_norm.shape[:2]
    H = cfg["hidden_size"]
    pa, ps = cfg.get("enable_past_actions", True), cfg.get("enable_past_states", True)
    ts = P["timestep_embedding.weight"][:T] if cfg.get("enable_timestep_embedding", True) else torch.zeros(T, H)   # :144-147
    images, ui, e = [], None, None
    if ps:                                                                                                  # :152-159
        e = vit_forward(P, "state_embedding_model.", frames.reshape(B * T, *frames.shape[2:]), cfg, taps, masks)
        ui = torch.tanh(F.linear(e, P["embed_state.weight"], P["embed_state.bias"]).reshape(B, T, H) + ts)
        if pa:
            images.append(ui)
    c = vit_forward(P, "cad_embedding_model.", cad, cfg, taps, masks)                                        # :162
    images.append(F.linear(c, P["embed_image.weight"], P["embed_image.bias"]).unsqueeze(1).repeat(1, T, 1))  # :163-164
    V = cfg.get("num_views", 0)
    if multiview is not None and V > 0:                                                                     # :167-170 (trajectory_model.py:77-87)
        mv = vit_forward(P, "cad_embedding_model.", multiview.reshape(B * V, *multiview.shape[2:]), cfg, None, masks)
        mv = mv.reshape(B, V, -1).unsqueeze(1).expand(-1, T, -1, -1).reshape(B, T, -1)
        images.append(F.linear(mv, P["embed_multiview.weight"], P["embed_multiview.bias"]))
    mem = torch.cat(images, dim=-1)
    if len(images) > 1:
        mem = F.linear(mem, P["image_projection.weight"], P["image_projection.bias"])                       # :172-174
    mem = torch.tanh(mem)                                                                                    # :175
    act = torch.tanh(F.linear(actions_norm, P["embed_action.weight"], P["embed_action.bias"]) + ts)          # :176-178
    if taps is not None:
        taps.update(cls_state=e, cls_cad=c, ui=ui, mem=mem, act=act)
    if pa:
        h = decoder_forward(P, act, mem, cfg, taps, masks)                                                   # :190-197
    elif ps:
        h = decoder_forward(P, ui, mem, cfg, taps, masks, tgt_band=True)                                     # :198-205
    else:
        h = decoder_forward(P, mem, mem, cfg, taps, masks, tgt_band=True)                                    # :206-213
    cmds = F.linear(h, P["predict_action_class_0_4.weight"], P["predict_action_class_0_4.bias"])      # :217
    params = F.linear(h, P["predict_action_class_0_999.weight"], P["predict_action_class_0_999.bias"]
                      ).reshape(B, T, cfg["num_params"], cfg["num_params_values"])                   # :218
    return cmds, params


def normalize_actions(actions: Tensor) -> Tensor:
    """trainer.py:800-804 (clone; cmd/4, params/1000; -1 padding -> -0.25 / -0.001)."""
    a = actions.clone()
    a[:, :, 0] = a[:, :, 0] / 4.0
    a[:, :, 1:] = a[:, :, 1:] / 1000.0
    return a


# ----------------------------------------------------------------------------------------
# loss + metrics (trainer.py:853-917, 935-1063)
# ----------------------------------------------------------------------------------------

def flexible_ce(logits: Tensor, targets: Tensor, tolerance: int) -> Tensor:
    """Closed form of flexible_cross_entropy as compute_loss calls it (`above=self.above`, a non-empty
    list -> always the one-sided window {clamp(t+o,0,999) : o in range(tolerance)}; ignore_valid=True):
    mean over rows with target != -1 and argmax outside the window of LSE(z) - mean_{j in W} z_j;
    constant 0.0 when no rows remain (trainer.py:872-873, 895-896)."""
    C = logits.shape[-1]
    logits = logits.reshape(-1, C); targets = targets.reshape(-1)
    keep = targets != -1
    logits, targets = logits[keep], targets[keep]
    if logits.shape[0] == 0:
        return torch.tensor(0.0)
    pred = logits.argmax(dim=1)
    inside = (pred >= targets) & (pred <= torch.clamp(targets + tolerance - 1, max=C - 1))
    logits, targets = logits[~inside], targets[~inside]
    if logits.shape[0] == 0:
        return torch.tensor(0.0)
    j = torch.arange(C)[None, :]
    hi = torch.clamp(targets + tolerance - 1, max=C - 1)[:, None]
    win = (j >= targets[:, None]) & (j <= hi)                     # unique clamped indices
    wsize = win.sum(dim=1).to(logits.dtype)
    lse = torch.logsumexp(logits, dim=1)
    return (lse - (logits * win).sum(dim=1) / wsize).mean()


def compute_loss(cmds: Tensor, params: Tensor, actions: Tensor, use_mse: bool = True,
                 class_weights: Optional[dict] = None, label_weights=None) -> Tuple[Tensor, dict]:
    """actions = raw (un-normalised) batch['actions'][:, 1:]  (trainer.py:490, 935-1063).
    label_weights: class_weights.json["Label"] as the trainer read it (trainer.py:822-825); default = the shipped file's values."""
    actions = actions.long()
    a_cmd, a_par = actions[..., 0], actions[..., 1:]
    LABEL_WEIGHTS = list(label_weights) if label_weights is not None else globals()["LABEL_WEIGHTS"]
    w = torch.tensor(LABEL_WEIGHTS, dtype=torch.float32)
    loss_cmd = F.cross_entropy(cmds.reshape(-1, 5), a_cmd.reshape(-1), weight=w, ignore_index=-1)
    loss_params = 0
    names = ["x", "y", "Key Pressed", "Times Key Pressed", "Scroll Amount", "Typed Value"]
    for i in range(6):
        pi = params[..., i, :].reshape(-1, 1000); ti = a_par[..., i].reshape(-1)
        if use_mse:
            lp = flexible_ce(pi, ti, TOLERANCES[i])
        else:
            cw = torch.tensor(class_weights[names[i]], dtype=torch.float32)
            lp = F.cross_entropy(pi, ti, weight=cw, ignore_index=-1)
        if not torch.isnan(lp):
            loss_params = loss_params + lp * LABEL_WEIGHTS[PARAM_TO_LABEL[i]]
    loss = 2 * loss_cmd + loss_params
    return loss, compute_metrics(cmds, params, actions, use_mse)


def _count_correct(pp: Tensor, ap: Tensor, pm: Tensor, i: int, use_mse: bool) -> int:
    d = pp[..., i][pm[..., i]] - ap[..., i][pm[..., i]]
    if use_mse and ABOVE[i]:
        return int(((d >= 0) & (d < TOLERANCES[i])).sum())
    return int((d.abs() < TOLERANCE).sum())


def compute_metrics(cmds: Tensor, params: Tensor, actions: Tensor, use_mse: bool = True) -> dict:
    """The metric dict of trainer.py:969-1061 (perfect_* are hard-wired to 0 upstream)."""
    a_cmd, a_par = actions[..., 0], actions[..., 1:]
    cp = cmds.argmax(dim=-1); pp = params.argmax(dim=-1)
    cmask = a_cmd != -1
    cmd_corrects = [int((cp[a_cmd == i] == i).sum()) for i in range(5)]
    cmd_counts = [int((a_cmd == i).sum()) for i in range(5)]
    param_mask = cmask[..., None] & (a_par != -1)
    params_mask = param_mask & (cp == a_cmd)[..., None]
    param_corrects = [_count_correct(pp, a_par, params_mask, i, use_mse) for i in range(6)]
    param_counts = [int(param_mask[..., i].sum()) for i in range(6)]
    k = 30
    if use_mse:
        ptk = sum(_count_correct(pp[:, :k], a_par[:, :k], params_mask[:, :k], i, True) for i in range(6))
    else:
        ptk = int((pp[:, :k][params_mask[:, :k]] == a_par[:, :k][params_mask[:, :k]]).sum())
    m = {
        "correct_predictions": int((cp[cmask] == a_cmd[cmask]).sum()) + sum(param_corrects),
        "total_predictions": int(cmask.sum()) + int(param_mask.sum()),
        "cmd_corrects": cmd_corrects, "cmd_counts": cmd_counts,
        "param_corrects": param_corrects, "param_counts": param_counts,
        "cmd_correct_topk": int((cp[:, :k][cmask[:, :k]] == a_cmd[:, :k][cmask[:, :k]]).sum()),
        "cmd_counts_topk": int(cmask[:, :k].sum()),
        "param_correct_topk": ptk,
        "param_counts_topk": int(param_mask[:, :k].sum()),
        "perfect_sequences": 0, "perfect_commands": 0, "total_sequences": 0, "perfect_sequence_accuracy": 0,
    }
    for i in range(6):
        m[f"param_corrects_{i}"] = param_corrects[i]; m[f"param_counts_{i}"] = param_counts[i]
    for i in range(5):
        m[f"cmd_corrects_{i}"] = cmd_corrects[i]; m[f"cmd_counts_{i}"] = cmd_counts[i]
    return m


# ----------------------------------------------------------------------------------------
# one optimiser step (trainer.py:480-496): forward -> loss -> backward -> clip(1.0) -> Adam
# ----------------------------------------------------------------------------------------

class OracleTrainer:
    """Holds fp32 parameters + Adam state; `step(batch)` restates BaseTrainer._process_batch in eval mode
    (dropout off — the reference's _process_batch does not itself switch the module mode)."""

    def __init__(self, params_np: dict, cfg: dict = CANONICAL_CONFIG, lr: float = 1e-5,
                 betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 1.0, use_mse: bool = True):
        self.cfg = dict(cfg)
        self.P = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in params_np.items()}
        self.m = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.t = 0
        self.lr, self.betas, self.eps, self.max_norm, self.use_mse = lr, betas, eps, max_norm, use_mse

    masks: Optional[dict] = None            # explicit dropout keep-multipliers (train-mode checks); None = eval mode

    def forward(self, batch: dict, taps: Optional[dict] = None):
        frames = torch.as_tensor(batch["frames"], dtype=torch.float32)
        actions = torch.as_tensor(batch["actions"], dtype=torch.float32)
        cad = torch.as_tensor(batch["cad_image"], dtype=torch.float32)
        mv = batch.get("multiview_images", None)
        mv = torch.as_tensor(mv, dtype=torch.float32) if mv is not None else None
        cmds, params = model_forward(self.P, frames[:, :-1], normalize_actions(actions[:, :-1]), cad, self.cfg, taps, self.masks, mv)
        return cmds, params, actions[:, 1:]

    def loss_and_grads(self, batch: dict):
        for p in self.P.values():
            p.grad = None
        cmds, params, tgt = self.forward(batch)
        loss, metrics = compute_loss(cmds, params, tgt, self.use_mse)
        loss.backward()
        return loss.detach(), metrics, cmds.detach(), params.detach()

    def step(self, batch: dict):
        loss, metrics, cmds, params = self.loss_and_grads(batch)
        grads = {k: p.grad for k, p in self.P.items() if p.grad is not None}
        total = self.apply_grads(grads)
        return loss, metrics, total, cmds, params

    def apply_grads(self, grads: dict) -> float:
        """clip_grad_norm_(1.0) + Adam on externally supplied gradients (e.g. the mean over data-parallel ranks)."""
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
        # clip_grad_norm_: coef = max_norm / (total + 1e-6), clamped to 1.0
        coef = torch.clamp(self.max_norm / (total + 1e-6), max=1.0)
        self.t += 1
        b1, b2 = self.betas
        with torch.no_grad():
            for k, g in grads.items():
                g = g * coef
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                bc1 = 1 - b1 ** self.t; bc2 = 1 - b2 ** self.t
                denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
                self.P[k].addcdiv_(self.m[k], denom, value=-self.lr / bc1)
        return float(total)
