"""Parity at BASELINE.json's REAL sizes: C2 = 32 clips x 64 steps, C4's per-GPU shape = 16 x 186 (the
three-key-block decoder attention kernels), C3 = 64 x 128 — through the C ABI, against the oracle.

Size-independent construction: the full batch is K repetitions of the same two clips.  Every clip is processed independently by
forward, the loss is a mean over (identical, repeated) rows and the parameter gradient a mean over clips, so the full-size run
must reproduce — logits, loss, metric counters / K, every parameter gradient — what the fp32 oracle computes on the 2-clip batch
(seconds on the host), up to summation order.  Nothing is sampled away: all B*T frames go through every kernel at full grid size."""
import numpy as np
import pytest
import torch

import oputil as U
from oracle import restatement as O
from videocad_amd import lib as L
from videocad_amd import synth
from videocad_amd.engine import NativeEngine, make_config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CFG_KEYS = ("hidden_size", "nhead", "num_decoder_layers", "dim_feedforward", "window_size", "act_dim", "num_classes", "num_params",
            "num_params_values", "max_ep_len", "vit_dim", "vit_depth", "vit_heads", "vit_dim_head", "vit_mlp", "image_size", "patch_size")
GRAD_PROBES = ["predict_action_class_0_999.weight", "predict_action_class_0_4.bias", "transformer_decoder.layers.7.linear2.weight",
               "transformer_decoder.layers.7.multihead_attn.in_proj_weight", "transformer_decoder.layers.0.self_attn.in_proj_weight",
               "transformer_decoder.layers.0.norm1.weight", "embed_action.weight", "timestep_embedding.weight", "image_projection.weight",
               "embed_state.weight", "embed_image.bias", "state_embedding_model.transformer.layers.5.0.to_qkv.weight",
               "state_embedding_model.transformer.layers.3.1.net.1.weight", "state_embedding_model.transformer.layers.0.0.to_out.0.weight",
               "state_embedding_model.to_patch_embedding.2.weight", "state_embedding_model.pos_embedding",
               "cad_embedding_model.transformer.layers.2.1.net.4.weight", "cad_embedding_model.to_patch_embedding.1.weight"]

_ORACLE = {}


def record(name, **vals):
    """measured accuracy numbers of the full-size runs, appended as JSON lines under gpurun_out/ (copied to profiles/<round>_fullsize_measured.jsonl
    when they are quoted): the gates below are ~1.5x these"""
    import json, os
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "fullsize_measured.jsonl"), "a") as f:
            f.write(json.dumps(dict(case=name, **vals)) + "\n")
    except OSError:
        pass


def oracle_two_clips(T):
    """fp32 oracle on the 2-clip batch of horizon T (cached per T: C2/C3/C4 differ in T)."""
    if T not in _ORACLE:
        shapes = O.param_shapes()
        weights = {k: synth.make_param(k, s) for k, s in shapes.items()}
        batch = synth.make_batch(2, T, seed=40 + T)
        ot = O.OracleTrainer(weights)
        loss, metrics, cmds, pars = ot.loss_and_grads(batch)
        gn = {n: float(ot.P[n].grad.double().norm()) for n in GRAD_PROBES}
        total = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in ot.P.values() if p.grad is not None)))
        _ORACLE[T] = dict(batch=batch, loss=float(loss), metrics=metrics, cmds=cmds.detach(), pars=pars.detach(), gn=gn, total=total)
    return _ORACLE[T]


def build(dtype):
    cfg = O.CANONICAL_CONFIG
    eng = NativeEngine(make_config(dtype=dtype, **{k: cfg[k] for k in CFG_KEYS}), DEV)
    for k, s in O.param_shapes(cfg).items():
        eng.view(k).copy_(synth.make_param_torch(k, s, DEV))
    eng.sync_shadow()
    return eng


def tiled(batch, K):
    rep = lambda a: torch.from_numpy(a).to(DEV).repeat(K, *([1] * (a.ndim - 1)))
    return rep(batch["frames"]), rep(batch["actions"]), rep(batch["cad_image"])


@pytest.mark.parametrize("name,B,T", [("C2", 32, 64), ("C4_per_gpu", 16, 186), ("C3", 64, 128)])
def test_full_size_step_matches_oracle_f32(name, B, T):
    ref = oracle_two_clips(T)
    K = B // 2
    eng = build(L.VCAD_F32)
    frames, actions, cad = tiled(ref["batch"], K)
    cmds, pars = eng.forward(frames[:, :-1], O.normalize_actions(actions[:, :-1]), cad)
    # every repetition of the two clips gives the oracle's logits: 1e-3 relative gate (measured ~1e-6), arg-max bit-exact
    for rep in (0, K // 2, K - 1):
        c, p = cmds[2 * rep: 2 * rep + 2].cpu(), pars[2 * rep: 2 * rep + 2].cpu()
        assert U.relerr(c, ref["cmds"]) < 1e-4 and U.relerr(p, ref["pars"]) < 1e-4, (name, rep, U.relerr(p, ref["pars"]))
        assert float((p - ref["pars"]).abs().max()) < 1e-3 * float(ref["pars"].abs().max())
        assert torch.equal(p.argmax(-1), ref["pars"].argmax(-1)) and torch.equal(c.argmax(-1), ref["cmds"].argmax(-1))
    loss, met = eng.loss(cmds, pars, actions[:, 1:], U.LABEL_W)
    assert abs(float(loss[0]) - ref["loss"]) < 2e-5 * abs(ref["loss"]), (name, float(loss[0]), ref["loss"])
    m = met.tolist()
    assert m[L.MET_TOTAL] == K * ref["metrics"]["total_predictions"] and m[L.MET_CORRECT] == K * ref["metrics"]["correct_predictions"]
    assert m[L.MET_PAR_COUNT:L.MET_PAR_COUNT + 6] == [K * x for x in ref["metrics"]["param_counts"]]
    eng.backward()
    for n, want in ref["gn"].items():
        got = float(eng.view(n, eng.grads).double().norm())
        assert abs(got - want) <= 2e-3 * want + 1e-9, (name, n, got, want)
    norm = eng.optimizer_step(lr=1e-5)
    assert abs(float(norm[0]) - ref["total"]) < 1e-3 * ref["total"]


@pytest.mark.parametrize("name,B,T", [("C2", 32, 64), ("C4_per_gpu", 16, 186), ("C3", 64, 128)])
def test_full_size_step_bf16_close_and_train_mode_sane(name, B, T):
    """Throughput mode at full size: logits / loss / gradients close to the fp32 oracle (reported), then ONE train-mode step with
    dropout 0.1 on the same engine: finite, gradient norm in the eval run's ballpark, weights move."""
    ref = oracle_two_clips(T)
    K = B // 2
    eng = build(L.VCAD_BF16)
    frames, actions, cad = tiled(ref["batch"], K)
    an = O.normalize_actions(actions[:, :-1])
    cmds, pars = eng.forward(frames[:, :-1], an, cad)
    p = pars[:2].cpu()
    rel, mae = U.relerr(p, ref["pars"]), float((p - ref["pars"]).abs().mean())
    agree = float((p.argmax(-1) == ref["pars"].argmax(-1)).float().mean())
    print(f"\n[{name} bf16 vs fp32 oracle] logit MAE {mae:.3e}  rel {rel:.3e}  argmax agreement {agree:.4f}")
    # gates at ~1.5x the measured values (profiles/r04_fullsize_measured.jsonl: rel 3.42e-3 .. 3.51e-3, arg-max agreement 0.9935 .. 0.9955, loss 5.5e-4 .. 6.1e-4,
    # worst probed gradient norm 1.46e-3 .. 1.62e-3 on the three shapes): a 2x regression of the throughput mode at the benchmarked shape fails here
    assert rel < 5.3e-3 and agree > 0.990, (name, rel, agree)
    assert torch.equal(pars[:2], pars[2 * (K - 1):])                       # repetitions are bit-identical: no cross-clip leakage at full grid size
    loss, _ = eng.loss(cmds, pars, actions[:, 1:], U.LABEL_W)
    assert abs(float(loss[0]) - ref["loss"]) < 1e-3 * abs(ref["loss"])
    eng.backward()
    worst = max(abs(float(eng.view(n, eng.grads).double().norm()) - w) / w for n, w in ref["gn"].items())
    record(name + "_bf16", logit_mae=mae, rel_err=rel, argmax_agreement=agree, loss_rel=abs(float(loss[0]) - ref["loss"]) / abs(ref["loss"]), worst_grad_norm_rel=worst)
    assert worst < 2.5e-3, worst
    g_eval = float(eng.optimizer_step(lr=0.0)[0])                           # lr 0: weights untouched, norm reported
    w0 = eng.view("embed_state.weight").clone()
    eng.set_dropout(0.1, seed=7)
    cmds, pars = eng.forward(frames[:, :-1], an, cad)
    loss_t, _ = eng.loss(cmds, pars, actions[:, 1:], U.LABEL_W)
    eng.backward()
    g_train = float(eng.optimizer_step(lr=1e-5)[0])
    assert np.isfinite(float(loss_t[0])) and np.isfinite(g_train) and 0.3 * g_eval < g_train < 3.0 * g_eval, (g_eval, g_train)
    assert not torch.equal(pars[:2], pars[2 * (K - 1):])                    # dropout masks differ per clip
    assert not torch.equal(eng.view("embed_state.weight"), w0)


def _engine_masks(eng, B, T, clip0, nclips):
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("engine_emu_helpers", os.path.join(os.path.dirname(__file__), "test_engine_emu.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod.engine_masks(eng, O.CANONICAL_CONFIG, B, T, clip0, nclips)


def train_mode_pairs_check(name, B, T, dtype, tol_logit, tol_grad, builder=None):
    """The mode bench.py times — model.train(), dropout 0.1 at every site (reference trainer.py:339,386) — at the benchmark's shape, against the oracle with the
    SAME masks.  B distinct clips go through one full-size train-mode forward; for the first and the last clip pair the engine's masks of
    exactly those clips are exported (a clip range is a contiguous index range of every site) and the oracle runs the pair with them as explicit multipliers:
    logits must agree.  Backward: the oracle's d loss / d logits of the pair, zeros for every other clip, are handed to the full-size backward — every kernel
    runs at full grid size, all other clips contribute exact zeros — and the probed parameter gradients must be the oracle's."""
    eng = (builder or build)(dtype)
    eng.set_dropout(0.1, seed=123)
    batch = synth.make_batch(B, T, seed=700 + T)
    frames, actions, cad = (torch.from_numpy(batch[k]).to(DEV) for k in ("frames", "actions", "cad_image"))
    an = O.normalize_actions(actions[:, :-1])
    cmds, pars = eng.forward(frames[:, :-1], an, cad)
    weights = {k: eng.view(k).cpu().numpy() for k in eng.table}
    worst_logit, worst_grad = 0.0, 0.0
    for clip0 in (0, B - 2):
        sub = {k: (v[clip0:clip0 + 2] if v is not None else None) for k, v in batch.items()}
        ot = O.OracleTrainer(weights)
        ot.masks = _engine_masks(eng, B, T, clip0, 2)
        oc, op, tgt = ot.forward(sub)
        oc.retain_grad(); op.retain_grad()
        loss, _ = O.compute_loss(oc, op, tgt, True)
        loss.backward()
        rc, rp = U.relerr(cmds[clip0:clip0 + 2], oc.detach()), U.relerr(pars[clip0:clip0 + 2], op.detach())
        worst_logit = max(worst_logit, rc, rp)
        assert rp < tol_logit and rc < tol_logit * 2.5, (name, clip0, rc, rp)
        agree = float((pars[clip0:clip0 + 2].argmax(-1).cpu() == op.detach().argmax(-1)).float().mean())
        # (random-init logits: top-1 / top-2 gaps go down to 4e-4 of the logit scale, SURVEY §6 — the 16-bit modes flip a few near-ties: measured 765 of 768 in f16)
        assert agree == 1.0 if dtype == L.VCAD_F32 else agree > (0.99 if dtype == L.VCAD_F16 else 0.985), (name, clip0, agree)
        dc = torch.zeros_like(cmds); dp = torch.zeros_like(pars)
        dc[clip0:clip0 + 2] = oc.grad.to(DEV); dp[clip0:clip0 + 2] = op.grad.to(DEV)
        eng.backward(dc, dp)
        for n in GRAD_PROBES:
            want = float(ot.P[n].grad.double().norm())
            got = float(eng.view(n, eng.grads).double().norm())
            worst_grad = max(worst_grad, abs(got - want) / want)
            assert abs(got - want) <= tol_grad * want + 1e-12, (name, clip0, n, got, want)
        # one full tensor, element for element: the frame tower's last to_qkv.weight (class-token attention path, r06) and a decoder matrix
        for n, tol_e in (("state_embedding_model.transformer.layers.5.0.to_qkv.weight", 10 * tol_grad), ("transformer_decoder.layers.0.self_attn.in_proj_weight", 10 * tol_grad)):
            assert U.relerr(eng.view(n, eng.grads), ot.P[n].grad) < tol_e, (name, clip0, n, U.relerr(eng.view(n, eng.grads), ot.P[n].grad))
    record(name + "_train_mode_pairs_dtype%d" % dtype, worst_logit_rel=worst_logit, worst_probed_grad_norm_rel=worst_grad)


@pytest.mark.parametrize("name,B,T", [("C2", 32, 64), ("C4_per_gpu", 16, 186)])
def test_full_size_train_mode_matches_oracle_on_clip_pairs_f32(name, B, T):
    train_mode_pairs_check(name, B, T, L.VCAD_F32, 1e-4, 2e-3)


@pytest.mark.parametrize("name,B,T", [("C2", 32, 64), ("C4_per_gpu", 16, 186)])
def test_full_size_train_mode_matches_oracle_on_clip_pairs_bf16(name, B, T):
    """the headline dtype (BASELINE's): gated at ~2x what it measures (gpurun_out/fullsize_measured.jsonl)"""
    train_mode_pairs_check(name, B, T, L.VCAD_BF16, 6e-3, 3e-3)      # measured 3.6e-3 .. 3.8e-3 on logits, 1.4e-3 on the probed gradient norms


def test_full_size_fp8_forward_mode_c5_per_gpu_shape():
    """BASELINE configs[4]'s per-GPU shape (16 clips x 186 steps) in the VCAD_FP8 forward mode (ViT Linears on the MXFP8 matrix cores):
    logits against the fp32 oracle at fp8 accuracy (reported), repetitions bit-identical, the bf16 backward on top, one train-mode step."""
    B, T = 16, 186
    ref = oracle_two_clips(T)
    K = B // 2
    eng = build(L.VCAD_BF16)
    eng.set_fp8(True)
    frames, actions, cad = tiled(ref["batch"], K)
    an = O.normalize_actions(actions[:, :-1])
    cmds, pars = eng.forward(frames[:, :-1], an, cad)
    p = pars[:2].cpu()
    rel, mae = U.relerr(p, ref["pars"]), float((p - ref["pars"]).abs().mean())
    agree = float((p.argmax(-1) == ref["pars"].argmax(-1)).float().mean())
    print(f"\n[C5 per-GPU shape, fp8 forward vs fp32 oracle] logit MAE {mae:.3e}  rel {rel:.3e}  argmax agreement {agree:.4f}")
    assert rel < 1.25e-2 and agree > 0.977, (rel, agree)          # ~1.5x measured (8.1e-3, 0.985: profiles/r04_fullsize_measured.jsonl)
    assert torch.equal(pars[:2], pars[2 * (K - 1):])
    loss, _ = eng.loss(cmds, pars, actions[:, 1:], U.LABEL_W)
    assert abs(float(loss[0]) - ref["loss"]) < 5e-2 * abs(ref["loss"])
    eng.backward()
    worst = max(abs(float(eng.view(n, eng.grads).double().norm()) - w) / w for n, w in ref["gn"].items())
    print(f"[C5 per-GPU shape, fp8 forward] worst probed gradient-norm error {worst:.3e}")
    record("C5_per_gpu_fp8_forward", logit_mae=mae, rel_err=rel, argmax_agreement=agree, worst_grad_norm_rel=worst)
    assert worst < 1.1e-2, worst                                       # measured 6.9e-3
    g_eval = float(eng.optimizer_step(lr=0.0)[0])
    eng.set_dropout(0.1, seed=11)
    cmds, pars = eng.forward(frames[:, :-1], an, cad)
    loss_t, _ = eng.loss(cmds, pars, actions[:, 1:], U.LABEL_W)
    eng.backward()
    g_train = float(eng.optimizer_step(lr=1e-5)[0])
    assert np.isfinite(float(loss_t[0])) and np.isfinite(g_train) and 0.3 * g_eval < g_train < 3.0 * g_eval, (g_eval, g_train)


@pytest.mark.parametrize("dtype,tol", [(L.VCAD_F32, 1e-4), (L.VCAD_BF16X3, 1e-4), (L.VCAD_BF16, 6e-3)], ids=["f32", "bf16x3", "bf16"])
def test_horizon_beyond_192_matches_oracle(dtype, tol):
    """r04: horizons past the released dataset's maximum (186) up to the reference's `max_ep_len` (model/autoregressive_transformer.py:17,80) — here
    T = 250 (four 64-step blocks: the block-streaming decoder attention in bf16 mode, sixteen-piece wave-per-row kernels in the fp32 modes;
    timestep rows up to 249): two clips against the fp32 oracle — logits, arg-max (exact in f32 / bf16x3), loss, probed gradient norms."""
    T = 250
    ref = oracle_two_clips(T)
    eng = build(dtype)
    frames, actions, cad = tiled(ref["batch"], 1)
    cmds, pars = eng.forward(frames[:, :-1], O.normalize_actions(actions[:, :-1]), cad)
    c, p = cmds.cpu(), pars.cpu()
    rel = U.relerr(p, ref["pars"])
    agree = float((p.argmax(-1) == ref["pars"].argmax(-1)).float().mean())
    record(f"T250_{dtype}", rel_err=rel, argmax_agreement=agree)
    assert rel < tol and U.relerr(c, ref["cmds"]) < tol, (rel, U.relerr(c, ref["cmds"]))
    if dtype != L.VCAD_BF16:
        assert torch.equal(p.argmax(-1), ref["pars"].argmax(-1)) and torch.equal(c.argmax(-1), ref["cmds"].argmax(-1))
    else:
        assert agree > 0.985, agree
    loss, _ = eng.loss(cmds, pars, actions[:, 1:], U.LABEL_W)
    assert abs(float(loss[0]) - ref["loss"]) < (2e-5 if dtype != L.VCAD_BF16 else 1.5e-3) * abs(ref["loss"])
    eng.backward()
    gt = 3e-3 if dtype != L.VCAD_BF16 else 6e-3
    for n, want in ref["gn"].items():
        got = float(eng.view(n, eng.grads).double().norm())
        assert abs(got - want) <= gt * want + 1e-9, (n, got, want)
