"""End-to-end parity of the HIP train step against the committed goldens (captured from the imported reference,
tests/golden/make_goldens.py) and, for cases without goldens, against the oracle restatement.  Through the C ABI."""
import json
import os

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


ENGINE_GEMM_FLAGS = 0     # set by the gemm_dma_mode fixture; every engine build() makes carries it
ENGINES = []


def build(dtype, cfg=O.CANONICAL_CONFIG):
    eng = NativeEngine(make_config(dtype=dtype, **{k: cfg[k] for k in CFG_KEYS}), DEV)
    eng.set_gemm_flags(ENGINE_GEMM_FLAGS)
    ENGINES.append(eng)
    shapes = O.param_shapes(cfg)
    assert set(eng.table) == set(shapes)
    for k, s in shapes.items():
        eng.view(k).copy_(synth.make_param_torch(k, s, DEV))
    eng.sync_shadow()
    return eng


def sl(t, n=64):
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long().to(f.device)
    return f[idx].cpu().numpy()


def run_case(eng, gold, B, T, seed, lengths, full_params):
    batch = synth.make_batch_torch(B, T, seed, DEV, lengths)
    frames, actions, cad = batch["frames"], batch["actions"], batch["cad_image"]
    cmds, pars = eng.forward(frames[:, :-1], O.normalize_actions(actions[:, :-1]), cad)
    gc = torch.from_numpy(gold["cmds"]).to(DEV); gp = torch.from_numpy(gold["params"]).to(DEV)
    pcmp = pars if full_params else pars[:, :, :, ::8]
    return batch, cmds, pars, gc, gp, pcmp


@pytest.mark.parametrize("case", ["c1_full", "c1_ragged"])
def test_f32_step_matches_reference_goldens(golden_dir, case):
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))["cases"][case]
    gold = np.load(os.path.join(golden_dir, case + ".npz"))
    eng = build(L.VCAD_F32)
    batch, cmds, pars, gc, gp, pcmp = run_case(eng, gold, meta["B"], meta["T"], meta["seed"], meta["lengths"], case == "c1_full")
    # gate: 1e-3 relative on logits (norm-wise and worst element vs the logit scale), bit-exact argmax
    assert U.relerr(cmds, gc) < 1e-4 and U.relerr(pcmp, gp) < 1e-4, (U.relerr(cmds, gc), U.relerr(pcmp, gp))
    assert float((pcmp - gp).abs().max()) < 1e-3 * float(gp.abs().max())
    assert np.array_equal(pars.argmax(-1).cpu().numpy(), gold["params_argmax"])
    assert np.array_equal(cmds.argmax(-1).cpu().numpy(), gold["cmds_argmax"])
    loss, met = eng.loss(cmds, pars, batch["actions"][:, 1:], U.LABEL_W)
    assert abs(float(loss[0]) - float(gold["loss"])) < 1e-4 * abs(float(gold["loss"]))
    gm = json.loads(str(gold["metrics_json"]))
    m = met.tolist()
    assert m[L.MET_CMD_COUNT:L.MET_CMD_COUNT + 5] == gm["cmd_counts"] and m[L.MET_CMD_CORRECT:L.MET_CMD_CORRECT + 5] == gm["cmd_corrects"]
    assert m[L.MET_PAR_COUNT:L.MET_PAR_COUNT + 6] == gm["param_counts"] and m[L.MET_PAR_CORRECT:L.MET_PAR_CORRECT + 6] == gm["param_corrects"]
    assert m[L.MET_CORRECT] == gm["correct_predictions"] and m[L.MET_TOTAL] == gm["total_predictions"]
    assert m[L.MET_CMD_CORRECT_TOPK] == gm["cmd_correct_topk"] and m[L.MET_PAR_COUNT_TOPK] == gm["param_counts_topk"]
    # backward: per-tensor gradient norms for all 300+ live tensors + slices
    eng.backward()
    names = [str(n) for n in gold["grad_names"]]
    bad = []
    for n, gn in zip(names, gold["grad_norms"]):
        mine = float(eng.view(n, eng.grads).double().norm())
        if abs(mine - gn) > 2e-3 * gn + 1e-9:
            bad.append((n, mine, float(gn)))
    assert not bad, bad[:8]
    for k in gold.files:
        if k.startswith("gslice:"):
            n = k[len("gslice:"):]
            ref = gold[k]; got = sl(eng.view(n, eng.grads))
            assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-9, (n, np.abs(got - ref).max(), np.abs(ref).max())
    norm = eng.optimizer_step(lr=1e-5)
    assert abs(float(norm[0]) - float(gold["total_grad_norm"])) < 1e-3 * float(gold["total_grad_norm"])
    for k in gold.files:
        if k.startswith("pslice:"):
            n = k[len("pslice:"):]
            assert np.abs(sl(eng.view(n)) - gold[k]).max() < 2e-6, n


@pytest.mark.parametrize("case", ["c1_full", "c1_ragged"])
def test_bf16x3_step_in_tolerance_of_reference_goldens(golden_dir, case):
    """VCAD_BF16X3 — fp32 tensors, every Linear and the ViT attention as three hi/lo-split bf16 MFMAs on pre-split tensors (r04), decoder attention on the f32 matrix cores — against the goldens
    of the imported reference: north_star's gate (logits within 1e-3 relative, arg-max bit-exact) with a decade to spare, and the whole
    step (loss, metrics, 309 gradient norms, clip norm, post-Adam weights) at the split's accuracy."""
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))["cases"][case]
    gold = np.load(os.path.join(golden_dir, case + ".npz"))
    eng = build(L.VCAD_BF16X3)
    batch, cmds, pars, gc, gp, pcmp = run_case(eng, gold, meta["B"], meta["T"], meta["seed"], meta["lengths"], case == "c1_full")
    rel_c, rel_p = U.relerr(cmds, gc), U.relerr(pcmp, gp)
    print(f"\n[measured bf16x3 {case}] logits rel cmd {rel_c:.3e} params {rel_p:.3e}  max abs {float((pcmp - gp).abs().max()):.3e} (logit scale {float(gp.abs().max()):.2f})")
    assert rel_c < 1e-4 and rel_p < 1e-4, (rel_c, rel_p)
    assert float((pcmp - gp).abs().max()) < 1e-3 * float(gp.abs().max())
    assert np.array_equal(pars.argmax(-1).cpu().numpy(), gold["params_argmax"])
    assert np.array_equal(cmds.argmax(-1).cpu().numpy(), gold["cmds_argmax"])
    loss, met = eng.loss(cmds, pars, batch["actions"][:, 1:], U.LABEL_W)
    assert abs(float(loss[0]) - float(gold["loss"])) < 1e-4 * abs(float(gold["loss"]))
    gm = json.loads(str(gold["metrics_json"]))
    m = met.tolist()
    assert m[L.MET_PAR_COUNT:L.MET_PAR_COUNT + 6] == gm["param_counts"] and m[L.MET_PAR_CORRECT:L.MET_PAR_CORRECT + 6] == gm["param_corrects"]
    assert m[L.MET_CORRECT] == gm["correct_predictions"] and m[L.MET_TOTAL] == gm["total_predictions"]
    eng.backward()
    rels = [abs(float(eng.view(str(n), eng.grads).double().norm()) - gn) / (gn + 1e-12) for n, gn in zip(gold["grad_names"], gold["grad_norms"])]
    print(f"[measured bf16x3 {case}] grad-norm rel err: median {np.median(rels):.3e} max {np.max(rels):.3e}")
    assert np.max(rels) < 3e-3, np.max(rels)
    norm = eng.optimizer_step(lr=1e-5)
    assert abs(float(norm[0]) - float(gold["total_grad_norm"])) < 1e-3 * float(gold["total_grad_norm"])
    for k in gold.files:
        if k.startswith("pslice:"):
            n = k[len("pslice:"):]
            assert np.abs(sl(eng.view(n)) - gold[k]).max() < 2e-6, n


@pytest.mark.parametrize("dtype,tol", [(L.VCAD_F32, 1e-4), (L.VCAD_BF16X3, 1e-4), (L.VCAD_BF16, 8e-3)], ids=["f32", "bf16x3", "bf16"])
@pytest.mark.parametrize("case", ["long_t186", "long_t70"])
def test_long_horizon_step_matches_reference_goldens(golden_dir, case, dtype, tol):
    """B = 1 at T = 186 (maximum horizon: three 64-key blocks, band mask far off the diagonal tile, timestep rows to 185) and T = 70 against
    goldens of the IMPORTED reference at those lengths (tests/golden/make_goldens.py, r03): logits, arg-max, loss, metrics, all 309
    gradient norms and the clip norm.  fp32 and bf16x3 must meet north_star's gate (1e-3 relative, arg-max exact); bf16 is gated at
    ~2x what it measures."""
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))["cases"][case]
    gold = np.load(os.path.join(golden_dir, case + ".npz"))
    eng = build(dtype)
    batch, cmds, pars, gc, gp, pcmp = run_case(eng, gold, meta["B"], meta["T"], meta["seed"], None, False)
    rel_c, rel_p = U.relerr(cmds, gc), U.relerr(pcmp, gp)
    agree = float((pars.argmax(-1).cpu().numpy() == gold["params_argmax"]).mean())
    print(f"\n[measured {case} dtype={dtype}] logits rel cmd {rel_c:.3e} params {rel_p:.3e} arg-max agreement {agree:.4f}")
    assert rel_c < tol and rel_p < tol, (rel_c, rel_p)
    if dtype != L.VCAD_BF16:
        assert float((pcmp - gp).abs().max()) < 1e-3 * float(gp.abs().max())
        assert agree == 1.0 and np.array_equal(cmds.argmax(-1).cpu().numpy(), gold["cmds_argmax"])
    else:
        assert agree >= 0.97
    loss, met = eng.loss(cmds, pars, batch["actions"][:, 1:], U.LABEL_W)
    ltol = 1e-4 if dtype != L.VCAD_BF16 else 2e-2
    assert abs(float(loss[0]) - float(gold["loss_fwd"])) < ltol * abs(float(gold["loss_fwd"]))
    if dtype != L.VCAD_BF16:
        gm = json.loads(str(gold["metrics_json"])); m = met.tolist()
        assert m[L.MET_PAR_COUNT:L.MET_PAR_COUNT + 6] == gm["param_counts"] and m[L.MET_PAR_CORRECT:L.MET_PAR_CORRECT + 6] == gm["param_corrects"]
        assert m[L.MET_CORRECT] == gm["correct_predictions"] and m[L.MET_TOTAL] == gm["total_predictions"]
    eng.backward()
    rels = [abs(float(eng.view(str(n), eng.grads).double().norm()) - gn) / (gn + 1e-12) for n, gn in zip(gold["grad_names"], gold["grad_norms"])]
    print(f"[measured {case} dtype={dtype}] grad-norm rel err: median {np.median(rels):.3e} max {np.max(rels):.3e}")
    assert np.max(rels) < (3e-3 if dtype != L.VCAD_BF16 else 5e-2), np.max(rels)
    norm = eng.optimizer_step(lr=1e-5)
    assert abs(float(norm[0]) - float(gold["total_grad_norm"])) < (1e-3 if dtype != L.VCAD_BF16 else 3e-2) * float(gold["total_grad_norm"])


def test_f32_window1_forward(golden_dir):
    gold = np.load(os.path.join(golden_dir, "win1.npz"))
    cfg = dict(O.CANONICAL_CONFIG); cfg["window_size"] = 1
    eng = build(L.VCAD_F32, cfg)
    batch, cmds, pars, gc, gp, pcmp = run_case(eng, gold, 2, 8, 3, None, False)
    assert U.relerr(cmds, gc) < 1e-4 and U.relerr(pcmp, gp) < 1e-4
    assert np.array_equal(pars.argmax(-1).cpu().numpy(), gold["params_argmax"])
    loss, _ = eng.loss(cmds, pars, batch["actions"][:, 1:], U.LABEL_W)
    assert abs(float(loss[0]) - float(gold["loss"])) < 1e-4 * abs(float(gold["loss"]))


@pytest.fixture(params=[-1, 1], ids=["gemm-auto", "gemm-dma-forced"])
def gemm_dma_mode(request):
    """bf16 tests run twice: automatic kernel choice (these small batches stay on the register-staged GEMM) and with every
    legal bf16 GEMM forced through the persistent DMA-fed kernel (the one the C2 bench shapes take)."""
    global ENGINE_GEMM_FLAGS
    ENGINE_GEMM_FLAGS = (L.GEMM_DMA_ALWAYS | L.GEMM_DYNAMIC) if request.param == 1 else 0      # (forced mode also draws the items with tickets, as the data-parallel trainer does)
    del ENGINES[:]
    used = lambda: sum(e.kernel_launches(L.KERNEL_GEMM_DMA) for e in ENGINES)
    yield request.param, used
    ENGINE_GEMM_FLAGS = 0
    del ENGINES[:]


def test_bf16_step_close_to_goldens(golden_dir, gemm_dma_mode):
    """Throughput mode: bf16 MFMA / fp32 accumulate.  Reported (not gated at 1e-3): logit MAE, norm-wise error, argmax agreement."""
    gold = np.load(os.path.join(golden_dir, "c1_full.npz"))
    eng = build(L.VCAD_BF16)
    batch, cmds, pars, gc, gp, _ = run_case(eng, gold, 2, 8, 1, None, True)
    mae = float((pars - gp).abs().mean()); rel = U.relerr(pars, gp)
    agree = float((pars.argmax(-1).cpu().numpy() == gold["params_argmax"]).mean())
    print(f"\n[measured bf16 vs fp32 reference] params-logit MAE {mae:.3e}  norm-wise rel {rel:.3e}  argmax agreement {agree:.3f}  cmd rel {U.relerr(cmds, gc):.3e}")
    # gates at ~1.5x what this mode measures (r02 driver run: rel 3.9e-3, MAE 7.3e-3, 95 of 96 arg-maxes): a regression of the throughput mode fails here
    assert rel < 6e-3 and mae < 1.2e-2 and agree >= 0.97, (rel, mae, agree)
    loss, _ = eng.loss(cmds, pars, batch["actions"][:, 1:], U.LABEL_W)
    assert abs(float(loss[0]) - float(gold["loss"])) < 2e-2 * abs(float(gold["loss"]))
    eng.backward()
    names = [str(n) for n in gold["grad_names"]]
    rels = []
    for n, gn in zip(names, gold["grad_norms"]):
        rels.append(abs(float(eng.view(n, eng.grads).double().norm()) - gn) / (gn + 1e-12))
    print(f"[measured bf16] grad-norm rel err: median {np.median(rels):.3e} max {np.max(rels):.3e}")
    assert np.median(rels) < 1e-2
    norm = eng.optimizer_step(lr=1e-5)
    assert abs(float(norm[0]) - float(gold["total_grad_norm"])) < 5e-2 * float(gold["total_grad_norm"])
    if gemm_dma_mode[0] == 1:
        assert gemm_dma_mode[1]() > 20, "forced mode did not reach the DMA kernel"


def test_fp8_forward_mode_close_to_goldens(golden_dir):
    """VCAD_FP8 (BASELINE configs[4]'s "fp8 MFMA" variant): ViT Linears on the block-scaled fp8 matrix cores, everything else bf16.
    Reported against the imported reference's fp32 goldens: logit MAE, norm-wise error, argmax agreement; the bf16 backward runs on top."""
    gold = np.load(os.path.join(golden_dir, "c1_full.npz"))
    eng = build(L.VCAD_BF16)
    eng.set_fp8(True)
    batch, cmds, pars, gc, gp, _ = run_case(eng, gold, 2, 8, 1, None, True)
    mae = float((pars - gp).abs().mean()); rel = U.relerr(pars, gp)
    agree = float((pars.argmax(-1).cpu().numpy() == gold["params_argmax"]).mean())
    print(f"\n[measured fp8 forward vs fp32 reference] params-logit MAE {mae:.3e}  norm-wise rel {rel:.3e}  argmax agreement {agree:.3f}  cmd rel {U.relerr(cmds, gc):.3e}")
    # ~1.5x the measured values (r02: rel 8.1e-3, MAE 1.5e-2, arg-max 97.9 %)
    assert rel < 1.3e-2 and mae < 2.3e-2 and agree >= 0.95, (rel, mae, agree)
    loss, _ = eng.loss(cmds, pars, batch["actions"][:, 1:], U.LABEL_W)
    assert abs(float(loss[0]) - float(gold["loss"])) < 5e-2 * abs(float(gold["loss"]))
    eng.backward()
    names = [str(n) for n in gold["grad_names"]]
    rels = [abs(float(eng.view(n, eng.grads).double().norm()) - gn) / (gn + 1e-12) for n, gn in zip(names, gold["grad_norms"])]
    print(f"[measured fp8 forward] grad-norm rel err: median {np.median(rels):.3e} max {np.max(rels):.3e}")
    assert np.median(rels) < 1.5e-2            # (r02: 5e-3)
    eng.optimizer_step(lr=1e-5)                                  # weights changed: the fp8 copies are rebuilt by the next forward
    c2, p2 = eng.forward(batch["frames"][:, :-1].contiguous(), O.normalize_actions(batch["actions"][:, :-1]), batch["cad_image"])
    assert torch.isfinite(p2).all()


def test_f32_causality_and_batch_independence():
    """Size-independent properties (reference SURVEY §3.3 probe): step t ignores inputs at steps > t; clips are independent."""
    eng = build(L.VCAD_F32)
    B, T = 3, 12
    batch = synth.make_batch_torch(B, T, 21, DEV)
    fr = batch["frames"][:, :-1].contiguous(); an = O.normalize_actions(batch["actions"][:, :-1]); cad = batch["cad_image"]
    c0, p0 = eng.forward(fr, an, cad); c0 = c0.clone(); p0 = p0.clone()
    fr2 = fr.clone(); an2 = an.clone()
    fr2[:, 7:] = -fr2[:, 7:]; an2[:, 7:] = 0.3
    c1, p1 = eng.forward(fr2, an2, cad)
    assert torch.equal(c0[:, :7], c1[:, :7]) and torch.equal(p0[:, :7], p1[:, :7])
    assert not torch.equal(p0[:, 7:], p1[:, 7:])
    perm = torch.tensor([2, 0, 1], device=DEV)
    c2, p2 = eng.forward(fr[perm].contiguous(), an[perm].contiguous(), cad[perm].contiguous())
    assert U.relerr(p2, p0[perm]) < 1e-6


def test_fused_loss_matches_reference_loss_cases(golden_dir):
    """compute_loss on synthetic logits, use_mse True and False (class-weighted CE with the NaN-skip), incl. the
    'all predictions inside the window -> constant 0' and 'no valid targets' branches (reference trainer.py:895-896, :961)."""
    cases = json.load(open(os.path.join(golden_dir, "loss_cases.json")))
    cw_json = json.load(open(os.path.join(golden_dir, "class_weights.json")))
    names = ["x", "y", "Key Pressed", "Times Key Pressed", "Scroll Amount", "Typed Value"]
    cw = torch.tensor([cw_json[k] for k in names], dtype=torch.float32, device=DEV).contiguous()
    lw = cw_json["Label"]
    B, T = cases["B"], cases["T"]
    eng = build(L.VCAD_F32)
    b = synth.make_batch_torch(B, T, 1, DEV)
    eng.forward(b["frames"][:, :-1], O.normalize_actions(b["actions"][:, :-1]), b["cad_image"])      # sizes the workspace for (B, T)
    for name, c in cases["cases"].items():
        seed, mode = c["seed"], c["mode"]
        acts = torch.from_numpy(synth.make_actions(B, T + 1, seed, lengths=cases["lengths"]))[:, 1:].clone()
        cm = torch.from_numpy(synth.hash_uniform(1000 + seed, B * T * 5).reshape(B, T, 5) * 3)
        pm = torch.from_numpy(synth.hash_uniform(2000 + seed, B * T * 6000).reshape(B, T, 6, 1000) * 3)
        if mode == "inside":
            pm.scatter_(-1, acts[..., 1:].long().clamp(min=0).unsqueeze(-1), 50.0)
        if mode == "empty":
            acts[..., 5:] = -1
        if mode == "edge":
            m = acts[..., 1:] >= 0
            acts[..., 1:][m] = torch.where(acts[..., 1:][m] > 500, torch.tensor(999.0), torch.tensor(998.0))
        cm, pm, acts = cm.to(DEV), pm.to(DEV).contiguous(), acts.to(DEV)
        from videocad_amd.trainer import metrics_from_counters
        out, met = eng.loss(cm, pm, acts, lw, use_mse=True)
        assert abs(float(out[0]) - c["loss_use_mse"]) < 2e-5 * max(1.0, abs(c["loss_use_mse"])), (name, float(out[0]), c["loss_use_mse"])
        assert metrics_from_counters(met.tolist()) == c["metrics_use_mse"], name
        out, met = eng.loss(cm, pm, acts, lw, use_mse=False, class_weights=cw)
        assert abs(float(out[0]) - c["loss_no_mse"]) < 2e-5 * max(1.0, abs(c["loss_no_mse"])), (name, float(out[0]), c["loss_no_mse"])
        assert metrics_from_counters(met.tolist()) == c["metrics_no_mse"], name


def _masks_for(eng, B, T):
    import importlib.util
    spec = importlib.util.spec_from_file_location("engine_emu_helpers", os.path.join(os.path.dirname(__file__), "test_engine_emu.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod.engine_masks(eng, O.CANONICAL_CONFIG, B, T)


@pytest.mark.parametrize("dtype,tol_logit,tol_grad,B,T", [(L.VCAD_F32, 1e-4, 2e-3, 2, 4), (L.VCAD_BF16, 3e-2, 6e-2, 2, 4),
                                                          (L.VCAD_BF16, 3e-2, 6e-2, 1, 70)],
                         ids=["f32-T4", "bf16-T4", "bf16-T70-long-decoder-attention"])
def test_train_mode_dropout_matches_oracle_with_same_masks(dtype, tol_logit, tol_grad, B, T, gemm_dma_mode):
    """Canonical model, p = 0.1 at every site: the engine's masks are exported (vcad_dropout_mask) and applied by the
    oracle as explicit multipliers.  fp32 mode must agree tightly; bf16 mode (MFMA attention path with in-register masks)
    within bf16 tolerance."""
    if dtype == L.VCAD_F32 and gemm_dma_mode[0] == 1:
        pytest.skip("the DMA kernel is bf16 only")
    if T > 64 and gemm_dma_mode[0] == 1:
        pytest.skip("one GEMM mode is enough for the long-horizon case")
    eng = build(dtype)
    eng.set_dropout(0.1, seed=77)
    batch = synth.make_batch(B, T, seed=8)
    weights = {k: eng.view(k).cpu().numpy() for k in eng.table}
    ot = O.OracleTrainer(weights)
    ot.masks = _masks_for(eng, B, T)
    oloss, ometrics, ocmds, opars = ot.loss_and_grads(batch)
    fr = torch.from_numpy(batch["frames"]).to(DEV); ac = torch.from_numpy(batch["actions"]).to(DEV); cad = torch.from_numpy(batch["cad_image"]).to(DEV)
    cmds, pars = eng.forward(fr[:, :-1], O.normalize_actions(ac[:, :-1]), cad)
    assert U.relerr(pars, opars) < tol_logit and U.relerr(cmds, ocmds) < tol_logit, (U.relerr(pars, opars), U.relerr(cmds, ocmds))
    loss, _ = eng.loss(cmds, pars, ac[:, 1:], U.LABEL_W)
    assert abs(float(loss[0]) - float(oloss)) < max(tol_logit, 1e-4) * abs(float(oloss))
    eng.backward()
    rels = []
    for k in weights:
        og = ot.P[k].grad
        if float(og.norm()) > 1e-7:
            rels.append((U.relerr(eng.view(k, eng.grads), og), k))
    worst = max(rels)
    med = float(np.median([r for r, _ in rels]))
    print(f"\n[dropout, dtype={dtype}] grad rel err: median {med:.3e}, worst {worst}")
    assert med < tol_grad and worst[0] < 40 * tol_grad, worst
    # eval mode is unaffected by the armed seed once p = 0
    eng.set_dropout(0.0, 0)
    c0, p0 = eng.forward(fr[:, :-1], O.normalize_actions(ac[:, :-1]), cad)
    ot.masks = None
    with torch.no_grad():
        oc, op, _ = ot.forward(batch)
    assert U.relerr(p0, op) < tol_logit
    if gemm_dma_mode[0] == 1:
        assert gemm_dma_mode[1]() > 20, "forced mode did not reach the DMA kernel"


@pytest.mark.parametrize("case", ["states_only", "actions_only"])
def test_f32_other_wirings_match_reference_goldens(golden_dir, case):
    """cad_and_past_10_states (tgt = UI embeddings, band self-attention) and cad_and_past_5_actions (no frame encoder, no
    timestep embedding, memory = tanh(CAD embedding)) — reference forward branches :198-213 — against goldens from the
    imported reference: logits, arg-max, loss, metrics, the set of live parameters and their gradient norms."""
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))["cases"][case]
    gold = np.load(os.path.join(golden_dir, case + ".npz"))
    rcfg = json.load(open(os.path.join(golden_dir, "model_configs.json")))[meta["config"]]
    cfg = dict(O.CANONICAL_CONFIG)
    cfg.update(window_size=rcfg["window_size"], enable_past_actions=rcfg.get("enable_past_actions", False),
               enable_past_states=rcfg.get("enable_past_states", False), enable_timestep_embedding=rcfg.get("enable_timestep_embedding", False))
    keys = CFG_KEYS + ("enable_past_actions", "enable_past_states", "enable_timestep_embedding")
    eng = NativeEngine(make_config(dtype=L.VCAD_F32, **{k: cfg[k] for k in keys}), DEV)
    shapes = O.param_shapes(cfg)
    assert set(eng.table) == set(shapes)
    for k, s in shapes.items():
        eng.view(k).copy_(synth.make_param_torch(k, s, DEV))
    batch, cmds, pars, gc, gp, pcmp = run_case(eng, gold, meta["B"], meta["T"], meta["seed"], None, False)
    assert U.relerr(cmds, gc) < 1e-4 and U.relerr(pcmp, gp) < 1e-4
    assert np.array_equal(pars.argmax(-1).cpu().numpy(), gold["params_argmax"])
    loss, met = eng.loss(cmds, pars, batch["actions"][:, 1:], U.LABEL_W)
    assert abs(float(loss[0]) - float(gold["loss"])) < 1e-4 * abs(float(gold["loss"]))
    from videocad_amd.trainer import metrics_from_counters
    assert metrics_from_counters(met.tolist()) == json.loads(str(gold["metrics_json"]))
    eng.backward()
    live = [str(n) for n in gold["grad_names"]]
    for n, gn in zip(live, gold["grad_norms"]):
        mine = float(eng.view(n, eng.grads).double().norm())
        assert abs(mine - gn) <= 2e-3 * gn + 1e-9, (n, mine, float(gn))
    for n in set(shapes) - set(live):                       # parameters the wiring never touches
        assert float(eng.view(n, eng.grads).abs().max()) == 0.0, n
    norm = eng.optimizer_step(lr=1e-5)
    assert abs(float(norm[0]) - float(gold["total_grad_norm"])) < 1e-3 * float(gold["total_grad_norm"])


@pytest.mark.parametrize("dtype,tol,gtol", [(L.VCAD_F32, 1e-4, 2e-3), (L.VCAD_BF16X3, 1e-4, 3e-3), (L.VCAD_BF16, 6e-3, 6e-2)])
def test_multiview_branch_matches_reference_goldens(golden_dir, dtype, tol, gtol):
    """f4 remainder: the multiview branch (reference model/autoregressive_transformer.py:72-74,167-170) on the canonical model with num_views = 2 against
    one train step of the imported reference (tests/golden/multiview_2.npz): logits, arg-max (exact in f32 / bf16x3), loss, every gradient norm —
    the CAD tower runs B (1 + V) images here, embed_multiview and the third image_projection block are live."""
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))["cases"]["multiview_2"]
    gold = np.load(os.path.join(golden_dir, "multiview_2.npz"))
    V = meta["num_views"]
    cfg = dict(O.CANONICAL_CONFIG); cfg.update(num_views=V)
    eng = NativeEngine(make_config(dtype=dtype, **{k: cfg[k] for k in CFG_KEYS + ("num_views",)}), DEV)
    shapes = O.param_shapes(cfg)
    assert set(eng.table) == set(shapes)
    for k, sh in shapes.items():
        eng.view(k).copy_(synth.make_param_torch(k, sh, DEV))
    eng.sync_shadow()
    batch = synth.make_batch_torch(meta["B"], meta["T"], meta["seed"], DEV, None, num_views=V)
    frames, actions, cad, mv = batch["frames"], batch["actions"], batch["cad_image"], batch["multiview_images"]
    cmds, pars = eng.forward(frames[:, :-1], O.normalize_actions(actions[:, :-1]), cad, mv)
    gc = torch.from_numpy(gold["cmds"]).to(DEV); gp = torch.from_numpy(gold["params"]).to(DEV)
    assert U.relerr(cmds, gc) < tol and U.relerr(pars[:, :, :, ::8], gp) < tol, (U.relerr(cmds, gc), U.relerr(pars[:, :, :, ::8], gp))
    agree = float((pars.argmax(-1).cpu().numpy() == gold["params_argmax"]).mean())
    assert agree == 1.0 if dtype != L.VCAD_BF16 else agree >= 0.95, agree
    loss, _ = eng.loss(cmds, pars, actions[:, 1:], U.LABEL_W)
    assert abs(float(loss[0]) - float(gold["loss"])) < (1e-4 if dtype != L.VCAD_BF16 else 2e-2) * abs(float(gold["loss"]))
    eng.backward()
    errs = []
    for n, gn in zip([str(x) for x in gold["grad_names"]], gold["grad_norms"]):
        errs.append((abs(float(eng.view(n, eng.grads).double().norm()) - gn) / (gn + 1e-12), n))
    assert max(errs)[0] < gtol * (1 if dtype != L.VCAD_BF16 else 4) and sorted(errs)[len(errs) // 2][0] < gtol, max(errs)
    # sampled gradient elements of the two tensors only this branch touches
    for key, name in (("grad_embed_multiview", "embed_multiview.weight"), ("grad_image_projection", "image_projection.weight")):
        assert U.relerr(torch.from_numpy(sl(eng.view(name, eng.grads), 256)), torch.from_numpy(gold[key])) < (5e-3 if dtype != L.VCAD_BF16 else 0.15), name
    # uint8 pixels for all three image inputs: the CAD-tower staging copies bytes, the patchify kernel normalises (same result as fp32 of the same pixels)
    if dtype == L.VCAD_F32:
        u8 = lambda t: ((t * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8)
        f8, c8, m8 = u8(frames[:, :-1]), u8(cad), u8(mv)
        back = lambda t: ((t.cpu().to(torch.float32) / 255.0 - 0.5) / 0.5).to(DEV)      # on the host: IEEE division, as the kernel does (torch's GPU division by a constant is not)
        ca, pa_ = eng.forward(f8, O.normalize_actions(actions[:, :-1]), c8, m8)
        cb, pb = eng.forward(back(f8), O.normalize_actions(actions[:, :-1]), back(c8), back(m8))
        assert torch.equal(pa_, pb) and torch.equal(ca, cb)


NHEAD8_MODES = [("f32", 1e-4, 2e-3), ("f16", 1e-3, 8e-3), ("bf16x3", 1e-4, 3e-3), ("bf16", 8e-3, 6e-2)]


@pytest.mark.parametrize("mode,tol,gtol", NHEAD8_MODES, ids=[m[0] for m in NHEAD8_MODES])
@pytest.mark.parametrize("case", ["nhead8_large", "nhead8_multiview3"])
def test_nhead8_configs_full_step_matches_reference_goldens(golden_dir, case, mode, tol, gtol):
    """r05: `cad_past_10_actions_and_states_large` (nhead 8 -> decoder head dim 128) and `..._large_multiview_only` (nhead 8,
    num_views 3: the CAD tower runs B (1 + 3) images, embed_multiview is [1024, 1536]) — reference model_configs/transformer_experiments.json:146,165 —
    against ONE FULL TRAIN STEP of the imported reference (tests/golden/nhead8_*.npz): logits, arg-max, loss, metrics, every gradient norm,
    sampled gradient elements, the clip norm and the post-Adam weights, through the C ABI.  f32: 1e-4 / exact arg-max; f16: north_star's 1e-3 /
    exact arg-max; bf16x3 like f32; bf16 (the throughput mode) measured and gated at its own level."""
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))["cases"][case]
    gold = np.load(os.path.join(golden_dir, case + ".npz"))
    V = meta["num_views"]
    dtype = {"f32": L.VCAD_F32, "f16": L.VCAD_F16, "bf16x3": L.VCAD_BF16X3, "bf16": L.VCAD_BF16}[mode]
    cfg = dict(O.CANONICAL_CONFIG); cfg.update(nhead=meta["nhead"], num_views=V)
    eng = NativeEngine(make_config(dtype=dtype, **{k: cfg[k] for k in CFG_KEYS + ("num_views",)}), DEV)
    shapes = O.param_shapes(cfg)
    assert set(eng.table) == set(shapes)
    for k, sh in shapes.items():
        eng.view(k).copy_(synth.make_param_torch(k, sh, DEV))
    eng.sync_shadow()
    batch = synth.make_batch_torch(meta["B"], meta["T"], meta["seed"], DEV, None, num_views=V) if V else synth.make_batch_torch(meta["B"], meta["T"], meta["seed"], DEV, None)
    frames, actions, cad, mv = batch["frames"], batch["actions"], batch["cad_image"], batch.get("multiview_images")
    cmds, pars = eng.forward(frames[:, :-1], O.normalize_actions(actions[:, :-1]), cad, mv) if V else eng.forward(frames[:, :-1], O.normalize_actions(actions[:, :-1]), cad)
    gc = torch.from_numpy(gold["cmds"]).to(DEV); gp = torch.from_numpy(gold["params"]).to(DEV)
    rel_c, rel_p = U.relerr(cmds, gc), U.relerr(pars[:, :, :, ::8], gp)
    agree = float((pars.argmax(-1).cpu().numpy() == gold["params_argmax"]).mean())
    print(f"\n[measured {mode} {case}] logits rel cmd {rel_c:.3e} params {rel_p:.3e}  arg-max agreement {agree:.5f}")
    assert rel_c < tol * (2.5 if mode == "f16" else 1.0) and rel_p < tol, (rel_c, rel_p)
    if mode != "bf16":
        assert agree == 1.0 and np.array_equal(cmds.argmax(-1).cpu().numpy(), gold["cmds_argmax"])
    else:
        assert agree >= 0.95, agree
    loss, met = eng.loss(cmds, pars, actions[:, 1:], U.LABEL_W)
    assert abs(float(loss[0]) - float(gold["loss"])) < {"f32": 1e-4, "bf16x3": 1e-4, "f16": 1e-3, "bf16": 2e-2}[mode] * abs(float(gold["loss"]))
    if mode in ("f32", "bf16x3"):
        from videocad_amd.trainer import metrics_from_counters
        assert metrics_from_counters(met.tolist()) == json.loads(str(gold["metrics_json"]))
    eng.backward()
    live = [str(n) for n in gold["grad_names"]]
    assert set(live) == set(shapes)
    rels = [abs(float(eng.view(n, eng.grads).double().norm()) - gn) / (gn + 1e-12) for n, gn in zip(live, gold["grad_norms"])]
    print(f"[measured {mode} {case}] grad-norm rel err: median {np.median(rels):.3e} max {np.max(rels):.3e}")
    assert np.max(rels) < gtol * (4 if mode == "bf16" else 1) and np.median(rels) < gtol / (1 if mode in ("f32", "bf16") else 3), (np.median(rels), np.max(rels), live[int(np.argmax(rels))])
    if mode == "f32":
        for k in gold.files:
            if k.startswith("gslice:"):
                ref = gold[k]; got = sl(eng.view(k[len("gslice:"):], eng.grads))
                assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-9, k
    norm = eng.optimizer_step(lr=1e-5)
    assert abs(float(norm[0]) - float(gold["total_grad_norm"])) < {"f32": 1e-3, "bf16x3": 1e-3, "f16": 4e-3, "bf16": 3e-2}[mode] * float(gold["total_grad_norm"])
    if mode in ("f32", "bf16x3"):
        for k in gold.files:
            if k.startswith("pslice:"):
                assert np.abs(sl(eng.view(k[len("pslice:"):])) - gold[k]).max() < 2e-6, k


def test_staged_backward_with_side_stream_is_bitwise_the_whole_backward():
    """The data-parallel order (stages 0-1, the CAD ViT's stage on the side stream, the frame ViT's stages, join) must produce exactly the
    gradients of the single-call backward: same kernels, only the stream they are issued on differs."""
    eng = build(L.VCAD_BF16)
    B, T = 2, 8
    batch = synth.make_batch_torch(B, T, 11, DEV, None)
    fr, ac, cad = batch["frames"], batch["actions"], batch["cad_image"]

    def run(staged):
        eng.set_dropout(0.1, seed=5)
        cmds, pars = eng.forward(fr[:, :-1], O.normalize_actions(ac[:, :-1]), cad)
        eng.loss(cmds, pars, ac[:, 1:], U.LABEL_W)
        eng.grads.zero_()
        if not staged:
            eng.backward()
        else:
            assert len(eng.buckets) == 5 and eng.side_stage == 2
            for st in range(eng.side_stage):
                eng.backward(stage=st)
            eng.backward(stage=eng.side_stage, side=True)
            for st in range(eng.side_stage + 1, len(eng.buckets)):
                eng.backward(stage=st)
            eng.join_side()
        torch.cuda.synchronize()
        return eng.grads.clone()

    g_whole, g_staged = run(False), run(True)
    assert bool(torch.equal(g_whole, g_staged)), float((g_whole - g_staged).abs().max())
    assert float(g_whole.abs().sum()) > 0


@pytest.mark.parametrize("pa,ps,tse", [(True, False, False), (False, True, True), (False, False, False)])
def test_bf16_train_mode_other_wirings(pa, ps, tse):
    """Throughput-mode train step (dropout on: deferred grouped decoder wgrads, LayerNorm-fused masked copies, side stream only when
    the frame ViT exists) on the three non-canonical wirings: its gradients must agree with the SAME engine run in eval-style
    ordering — p = 0.1 masks are part of both runs, but the second goes through the staged entry point (no side stream, no
    deferral to it) — and two identical runs must be bitwise equal."""
    cfg = dict(O.CANONICAL_CONFIG)
    cfg.update(enable_past_actions=pa, enable_past_states=ps, enable_timestep_embedding=tse)
    keys = CFG_KEYS + ("enable_past_actions", "enable_past_states", "enable_timestep_embedding")
    eng = NativeEngine(make_config(dtype=L.VCAD_BF16, **{k: cfg[k] for k in keys}), DEV)
    for k, sh in O.param_shapes(cfg).items():
        eng.view(k).copy_(synth.make_param_torch(k, sh, DEV))
    eng.sync_shadow()
    batch = synth.make_batch_torch(2, 6, 21, DEV, None)
    fr, ac, cad = batch["frames"], batch["actions"], batch["cad_image"]

    def run(staged):
        eng.set_dropout(0.1, seed=9)
        cmds, pars = eng.forward(fr[:, :-1], O.normalize_actions(ac[:, :-1]), cad)
        loss, _ = eng.loss(cmds, pars, ac[:, 1:], U.LABEL_W)
        eng.grads.zero_()
        if staged:
            for st in range(len(eng.buckets)):
                eng.backward(stage=st)
        else:
            eng.backward()
        torch.cuda.synchronize()
        return float(loss[0]), eng.grads.clone()

    l1, g1 = run(False); l2, g2 = run(False); l3, g3 = run(True)
    assert np.isfinite(l1) and bool(torch.isfinite(g1).all()) and float(g1.abs().sum()) > 0
    assert l1 == l2 and bool(torch.equal(g1, g2)), "train step is not deterministic"
    assert bool(torch.equal(g1, g3)), float((g1 - g3).abs().max())


def test_bucket_callback_orders_the_exchange_behind_the_right_stream():
    """vcad_set_bucket_callback on the device: the hook gets, per bucket, the stream its finalising launches are on (the caller's, or the library's side stream
    for the CAD ViT's bucket and — train mode — for bucket 0's deferred weight gradients); an "exchange" enqueued on a third stream behind an event recorded
    there (here: doubling the range, = all-reduce(SUM) over two identical ranks) followed by grad_scale = 1/2 must give bitwise the plain step."""
    import ctypes as C
    B, T = 2, 8
    batch = synth.make_batch_torch(B, T, 11, DEV, None)
    fr, ac, cad = batch["frames"], batch["actions"], batch["cad_image"]

    def step(hook):
        eng = build(L.VCAD_BF16)
        comm = torch.cuda.Stream(device=DEV)
        seen = []
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int64, C.c_void_p)

        def cb(user, bucket, grads, count, stream):
            lo, hi = eng.buckets[bucket]
            src = torch.cuda.ExternalStream(stream, device=DEV) if stream else torch.cuda.current_stream(DEV)
            ev = torch.cuda.Event(); ev.record(src)
            comm.wait_event(ev)
            with torch.cuda.stream(comm):
                eng.grads[lo:hi].mul_(2.0)
            seen.append((bucket, int(stream or 0)))
            return 0
        keep = CB(cb)
        if hook:
            L.check(eng.lib, eng.lib.vcad_set_bucket_callback(eng.h, C.cast(keep, C.c_void_p), None), "set_bucket_callback")
        eng.set_dropout(0.1, seed=5)
        cmds, pars = eng.forward(fr[:, :-1], O.normalize_actions(ac[:, :-1]), cad)
        eng.loss(cmds, pars, ac[:, 1:], U.LABEL_W)
        eng.backward()
        torch.cuda.current_stream(DEV).wait_stream(comm)
        eng.optimizer_step(lr=1e-4, grad_scale=0.5 if hook else 1.0)
        torch.cuda.synchronize()
        return eng.params.clone(), seen

    p_plain, _ = step(False)
    p_hook, seen = step(True)
    assert [b for b, _ in seen] == [0, 1, 2, 3, 4]
    cur = torch.cuda.current_stream(DEV).cuda_stream
    assert seen[1][1] == cur and seen[3][1] == cur and seen[4][1] == cur
    assert seen[0][1] == seen[2][1] and seen[0][1] != cur            # bucket 0 (deferred wgrads) and the CAD ViT's bucket are final on the side stream
    assert bool(torch.equal(p_plain, p_hook)), float((p_plain - p_hook).abs().max())
