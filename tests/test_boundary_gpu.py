"""The drop-in surface on a real MI355X: ModelFactory -> model(inputs) / trainer._process_batch against the goldens."""
import json
import os

import numpy as np
import pytest
import torch

import oputil as U
from oracle import restatement as O
from videocad_amd import synth
from videocad_amd.model_factory import ModelFactory
from videocad_amd.trainer import create_trainer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
CANON = json.load(open(os.path.join(HERE, "golden", "model_configs.json")))["cad_past_10_actions_and_states_timestep_embedding"]


def sl(t, n=64):
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long().to(f.device)
    return f[idx].cpu().numpy()


def make(dtype, tmp_path):
    import shutil
    shutil.copy(os.path.join(HERE, "golden", "class_weights.json"), os.path.join(str(tmp_path), "class_weights.json"))
    os.chdir(tmp_path)                                   # the trainer reads ./class_weights.json and writes logs/ like the reference
    sd = {k: synth.make_param_torch(k, s, DEV) for k, s in O.param_shapes().items()}
    model, mtype = ModelFactory().create_model(CANON["model_name"], dict(CANON, compute_dtype=dtype), DEV, state_dict=sd)
    model.eval()            # goldens were captured with dropout off (the reference's _process_batch does not switch modes)
    pk = {"loader": [], "sampler": None}
    tr = create_trainer(pk, pk, pk, model, {"lr": 1e-5, "use_mse": True, "experiment_name": "t"}, DEV, mtype, rank=0)
    return model, tr


def test_process_batch_matches_reference_goldens_f32(golden_dir, tmp_path):
    gold = np.load(os.path.join(golden_dir, "c1_full.npz"))
    model, tr = make("f32", tmp_path)
    batch = synth.make_batch_torch(2, 8, 1, "cpu")                      # host batch, like the loader hands it over
    loss, metrics = tr._process_batch(batch)
    assert abs(float(loss) - float(gold["loss"])) < 1e-4 * float(gold["loss"])
    assert metrics == json.loads(str(gold["metrics_json"]))
    for k in gold.files:
        if k.startswith("pslice:"):
            n = k[len("pslice:"):]
            assert np.abs(sl(dict(model.named_parameters())[n]) - gold[k]).max() < 2e-6, n


def test_reference_style_loop_through_autograd_f32(golden_dir, tmp_path):
    """The reference's own sequence (trainer.py:480-496) on our classes: zero_grad, model(inputs), compute_loss,
    loss.backward(), clip_grad_norm_, torch.optim.Adam.step()."""
    gold = np.load(os.path.join(golden_dir, "c1_full.npz"))
    model, tr = make("f32", tmp_path)
    opt = torch.optim.Adam(model.parameters(), lr=1e-5)
    batch = synth.make_batch_torch(2, 8, 1, DEV)
    opt.zero_grad()
    bd = tr.prepare_batch(batch)
    cmds, pars = model(tr._prepare_model_inputs(bd, False))
    assert np.array_equal(pars.argmax(-1).cpu().numpy(), gold["params_argmax"])
    loss, metrics = tr.compute_loss((cmds, pars), bd["actions"][:, 1:])
    loss.backward()
    total = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    assert abs(float(total) - float(gold["total_grad_norm"])) < 1e-3 * float(gold["total_grad_norm"])
    opt.step()
    for k in gold.files:
        if k.startswith("pslice:"):
            n = k[len("pslice:"):]
            assert np.abs(sl(dict(model.named_parameters())[n]) - gold[k]).max() < 2e-6, n
    # a second forward must see the torch-optimiser update (bf16 shadow / fp32 weights are re-read)
    c2, p2 = model(tr._prepare_model_inputs(bd, False))
    assert not torch.equal(p2, pars)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_bf16_trainer_runs(tmp_path, dtype):
    model, tr = make(dtype, tmp_path)
    model.train()           # dropout 0.1 active: the train loop of the reference
    batch = synth.make_batch_torch(2, 6, 5, "cpu")
    l0, _ = tr._process_batch(batch)
    for _ in range(3):
        l1, m = tr._process_batch(batch)
    assert torch.isfinite(l1) and float(l1) < float(l0) + 1e-3          # lr 1e-5: loss must not blow up


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-4), ("bf16", 3e-2), ("f16", 1e-3)])
def test_cached_sequential_inference_matches_oracle_prefix_runs(tmp_path, dtype, tol):
    """f1 (reference model/autoregressive_transformer.py:222-275): step t of the cached run == last row of the ORACLE's forward
    on the prefix [0..t] — what the reference's O(T^2) loop computes — and the literal loop (cached=False) agrees too."""
    model, tr = make(dtype, tmp_path)
    B, T = 2, 5
    b = synth.make_batch(B, T - 1, seed=6)
    frames, cad = torch.from_numpy(b["frames"]).to(DEV), torch.from_numpy(b["cad_image"]).to(DEV)
    P = {k: torch.from_numpy(synth.make_param(k, s)) for k, s in O.param_shapes().items()}
    cmds, pars = model.sequential_inference(frames, cad, action=False)
    assert cmds.shape == (B, T, 5) and pars.shape == (B, T, 6, 1000)
    for t in range(T):
        with torch.no_grad():
            oc, op = O.model_forward(P, frames[:, : t + 1].cpu(), torch.zeros(B, t + 1, 7), cad.cpu())[:2]
        # (the command logits of one step are ten numbers: their norm-wise error scatters around the 12 000-element parameter logits' — 2.5x for fp16's gate)
        assert U.relerr(pars[:, t], op[:, -1]) < tol and U.relerr(cmds[:, t], oc[:, -1]) < tol * (2.5 if dtype == "f16" else 1.0), (t, U.relerr(pars[:, t], op[:, -1]), U.relerr(cmds[:, t], oc[:, -1]))
        if dtype != "bf16":
            assert torch.equal(pars[:, t].argmax(-1).cpu(), op[:, -1].argmax(-1))
    c0, p0 = model.sequential_inference(frames, cad, action=False, cached=False)
    assert U.relerr(p0, pars) < {"f32": 2e-5, "bf16": 2e-2, "f16": 2.5e-3}[dtype]
    # action feedback: equals ONE teacher-forced oracle forward on the actions the run fed itself
    c2, p2 = model.sequential_inference(frames, cad, action=True)
    fed = [torch.zeros(B, 1, 7, device=DEV)]
    for t in range(T - 1):
        fed.append(model._next_action(c2[:, t:t + 1], p2[:, t:t + 1]))
    with torch.no_grad():
        oc, op = O.model_forward(P, frames.cpu(), torch.cat(fed, 1).cpu(), cad.cpu())[:2]
    assert U.relerr(p2, op) < tol and U.relerr(c2, oc) < tol * (2.5 if dtype == "f16" else 1.0)


@pytest.mark.parametrize("tag,cfg_name", [("canon", None), ("nhead8_large", "cad_past_10_actions_and_states_large")])
@pytest.mark.parametrize("dtype,tol", [("f32", 1e-4), ("f16", 1e-3), ("bf16", 3e-2)])
def test_sequential_inference_matches_the_reference_fixture(golden_dir, tmp_path, tag, cfg_name, dtype, tol):
    """f1 pinned to the reference's OWN function: tests/golden/seqinf.npz holds what the imported reference's `sequential_inference`
    (model/autoregressive_transformer.py:222-275) returned with action=False (the branch that runs upstream; action=True raises in apply_action_mask) for the canonical
    and the nhead-8 configuration (tests/golden/make_goldens.py --only-seqinf).  The cached incremental path must reproduce it step for step."""
    import shutil
    gold = np.load(os.path.join(golden_dir, "seqinf.npz"))
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))["seqinf"][tag]
    shutil.copy(os.path.join(HERE, "golden", "class_weights.json"), os.path.join(str(tmp_path), "class_weights.json"))
    os.chdir(tmp_path)
    rcfg = CANON if cfg_name is None else json.load(open(os.path.join(golden_dir, "model_configs.json")))[cfg_name]
    ocfg = dict(O.CANONICAL_CONFIG); ocfg.update(nhead=rcfg["nhead"], window_size=rcfg["window_size"])
    sd = {k: synth.make_param_torch(k, s, DEV) for k, s in O.param_shapes(ocfg).items()}
    model, _ = ModelFactory().create_model(rcfg["model_name"], dict(rcfg, compute_dtype=dtype), DEV, state_dict=sd)
    model.eval()
    B, T = meta["B"], meta["T"]
    b = synth.make_batch(B, T - 1, seed=meta["seed"])
    frames, cad = torch.from_numpy(b["frames"]).to(DEV), torch.from_numpy(b["cad_image"]).to(DEV)
    cmds, pars = model.sequential_inference(frames, cad, action=False)
    gc, gp = torch.from_numpy(gold[f"{tag}:a0:cmds"]), torch.from_numpy(gold[f"{tag}:a0:params"])
    rc, rp = U.relerr(cmds, gc), U.relerr(pars[:, :, :, ::8], gp)
    print(f"\n[seqinf {tag} {dtype}] rel cmds {rc:.3e} params {rp:.3e}")
    assert rp < tol and rc < tol * (2.5 if dtype == "f16" else 1.0), (tag, rc, rp)
    agree = float((pars.argmax(-1).cpu().numpy() == gold[f"{tag}:a0:params_argmax"]).mean())
    assert np.array_equal(cmds.argmax(-1).cpu().numpy(), gold[f"{tag}:a0:cmds_argmax"]) or dtype == "bf16"
    # parameter arg-max: exact in fp32; the fixture's smallest top-1 / top-2 gap is 3.9e-4 (nhead8_large, meta.json) — inside fp16's 4.7e-4 logit error, so the
    # 16-bit modes may flip that near-tie (96 arg-maxes per case: one flip = 0.9896)
    assert agree == 1.0 if dtype == "f32" else agree >= (0.979 if dtype == "f16" else 0.95), (tag, dtype, agree)


def test_uint8_input_path_is_bit_identical_and_staged_from_pinned_memory(tmp_path):
    """f2 / a2 / a18: pkl-style uint8 RGB frames -> PIL-exact gray on the host -> pinned uint8 batch -> double-buffered H2D ->
    normalisation inside the patchify kernel  ==  the reference's fp32 batch, bit for bit (loss, metrics, updated weights)."""
    from videocad_amd import data as D
    rng = np.random.default_rng(3)
    items_u8, items_f32 = [], []
    for n in (5, 7):
        rgb = rng.integers(0, 256, (n, 224, 224, 3), dtype=np.uint8)
        cad = torch.from_numpy(D.cv2_bgr2gray_u8(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8))).unsqueeze(0)
        act = torch.from_numpy(synth.make_actions(1, n, 30 + n)[0])
        items_u8.append({"frames": D.frames_from_rgb(rgb, True), "actions": act, "cad_image": cad, "multiview_images": None})
        items_f32.append({"frames": D.frames_from_rgb(rgb, False), "actions": act, "cad_image": D.normalize_u8(cad), "multiview_images": None})
    bu, bf = D.collate_with_padding(items_u8), D.collate_with_padding(items_f32)
    assert bu["frames"].is_pinned() and bf["frames"].is_pinned() and bu["frames"].shape == (2, 7, 1, 224, 224)
    assert bu["frames"].numel() * 4 == bf["frames"].numel() * bf["frames"].element_size()          # a quarter of the PCIe bytes
    res = []
    for batch in (bu, bf):
        model, tr = make("bf16", tmp_path)
        staged = list(D.DeviceStager([batch, batch], DEV))
        assert staged[0]["frames"].device.type == "cuda" and staged[0]["frames"].dtype == batch["frames"].dtype
        l0, m0 = tr._process_batch(staged[0]); l1, m1 = tr._process_batch(staged[1])
        res.append((float(l0), float(l1), m0, model.embed_state.weight.detach().clone(), model.state_embedding_model.to_patch_embedding[1].weight.detach().clone()))
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and res[0][2] == res[1][2]
    assert torch.equal(res[0][3], res[1][3]) and torch.equal(res[0][4], res[1][4])


def test_evaluate_and_find_first_mistake_match_the_reference_fixture(golden_dir, tmp_path):
    """f3 on the GPU: `trainer.evaluate` (fused loss + on-device counters, one D2H) and `trainer.find_first_mistake` on a two-batch loader
    (one ragged) against what the IMPORTED reference's trainer produced for the same weights and batches (tests/golden/eval_cases.json:
    reference trainer.py:713-750, 1131-1260) — every accumulated count / percentage and the complete mistake bookkeeping; the loss against
    the oracle's (the reference's evaluate does not return one)."""
    gold = json.load(open(os.path.join(golden_dir, "eval_cases.json")))
    model, tr = make("f32", tmp_path)
    loader = [synth.make_batch_torch(b["B"], b["T"], b["seed"], "cpu", b["lengths"]) for b in gold["batches"]]
    tr.train_loader = tr.val_loader = tr.test_loader = loader
    ev = tr.evaluate(model, mode="test")
    for k, v in gold["evaluate"].items():
        assert k in ev, k
        assert (abs(ev[k] - v) < 1e-9 * max(1.0, abs(v))) if isinstance(v, float) else ev[k] == v, (k, ev[k], v)
    ot = O.OracleTrainer({k: synth.make_param(k, s) for k, s in O.param_shapes().items()})
    losses = []
    for b in gold["batches"]:
        with torch.no_grad():
            c, p, tgt = ot.forward(synth.make_batch(b["B"], b["T"], b["seed"], b["lengths"]))
            losses.append(float(O.compute_loss(c, p, tgt)[0]))
    assert abs(ev["loss"] - sum(losses) / len(losses)) < 1e-4 * abs(sum(losses) / len(losses))
    ffm = tr.find_first_mistake(model, mode="test", tol=3)
    assert json.loads(json.dumps(ffm)) == gold["find_first_mistake"]
    assert os.path.exists(os.path.join("logs", "t", "test.json"))                  # evaluate dumps where the reference does (:746-749)


def test_rgb8_and_uint8_batches_match_the_oracle_on_the_host_converted_batch(tmp_path):
    """f2, oracle-direct: the SAME pixels three ways — stored uint8 RGB [B,S,H,W,3] (vcad_forward_rgb8: PIL luma + ToTensor +
    Normalize in the patchify kernel), host-grayed uint8 (vcad_forward_u8), host-normalised fp32 (the reference's contract) — give bit-identical
    logits, and those match the ORACLE evaluated on the reference's host pipeline output."""
    from videocad_amd import data as D
    model, tr = make("f32", tmp_path)
    eng = model._engine
    rng = np.random.default_rng(11)
    rgb = rng.integers(0, 256, (2, 6, 224, 224, 3), dtype=np.uint8)
    cad8 = torch.from_numpy(D.cv2_bgr2gray_u8(rng.integers(0, 256, (2, 224, 224, 3), dtype=np.uint8))).unsqueeze(1)
    act = torch.from_numpy(synth.make_actions(2, 6, 17))
    an = O.normalize_actions(act[:, :-1]).to(DEV)
    host = D.frames_from_rgb(rgb.reshape(-1, 224, 224, 3), as_uint8=False).reshape(2, 6, 1, 224, 224)          # PIL convert('L') -> ToTensor -> Normalize
    gray8 = D.frames_from_rgb(rgb.reshape(-1, 224, 224, 3), as_uint8=True).reshape(2, 6, 1, 224, 224)
    outs = []
    for fr, cad in ((torch.from_numpy(rgb), cad8), (gray8, cad8), (host, D.normalize_u8(cad8))):
        c, p = eng.forward(fr.to(DEV)[:, :-1], an, cad.to(DEV))
        outs.append((c.clone(), p.clone()))
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1]) and torch.equal(outs[1][1], outs[2][1])
    P = {k: torch.from_numpy(synth.make_param(k, s)) for k, s in O.param_shapes().items()}
    with torch.no_grad():
        oc, op = O.model_forward(P, host[:, :-1], an.cpu(), D.normalize_u8(cad8))[:2]
    assert U.relerr(outs[0][0], oc) < 1e-4 and U.relerr(outs[0][1], op) < 1e-4
    assert bool((outs[0][1].argmax(-1).cpu() == op.argmax(-1)).all())


def test_checkpoint_roundtrip_through_the_factory(tmp_path):
    """f4: save in the reference's format, reload through ModelFactory.create_model(state_dict=...) with DDP prefixes, resume."""
    model, tr = make("bf16", tmp_path)
    batch = synth.make_batch_torch(2, 4, 8, DEV)
    tr._process_batch(batch)
    ck = tr.save_checkpoint(0, 1.25)
    path = os.path.join("checkpoints", "t", "epoch_1.pt")
    assert os.path.exists(path)
    disk = torch.load(path, map_location="cpu")
    assert set(disk) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"} and disk["loss"] == 1.25
    sd = {"module._orig_mod." + k: v for k, v in disk["model_state_dict"].items()}
    model2, mtype = ModelFactory().create_model(CANON["model_name"], dict(CANON, compute_dtype="bf16"), DEV, state_dict=sd)
    model2.eval()
    pk = {"loader": [], "sampler": None}
    tr2 = create_trainer(pk, pk, pk, model2, {"lr": 1e-5, "use_mse": True, "experiment_name": "t2"}, DEV, mtype, rank=0)
    tr2.load_checkpoint(path)
    l1, _ = tr._process_batch(batch); l2, _ = tr2._process_batch(batch)
    assert float(l1) == float(l2) and torch.equal(model.embed_image.weight, model2.embed_image.weight)


def test_head_dim_128_config_runs(tmp_path):
    """nhead = 8 (reference final_experiments.json: cad_and_past_10_states, cad_and_past_5_actions; the *_large configs)."""
    import shutil
    shutil.copy(os.path.join(HERE, "golden", "class_weights.json"), os.path.join(str(tmp_path), "class_weights.json")); os.chdir(tmp_path)
    cfg = dict(CANON, nhead=8, compute_dtype="f32")
    ocfg = dict(O.CANONICAL_CONFIG, nhead=8)
    sd = {k: synth.make_param_torch(k, s, DEV) for k, s in O.param_shapes(ocfg).items()}
    model, _ = ModelFactory().create_model("x", cfg, DEV, state_dict=sd)
    model.eval()
    b = synth.make_batch(2, 5, seed=12)
    fr, ac, cad = (torch.from_numpy(b[k]).to(DEV) for k in ("frames", "actions", "cad_image"))
    with torch.no_grad():
        cmds, pars = model({"frames": fr[:, :-1], "actions": O.normalize_actions(ac[:, :-1]), "cad_image": cad})
        oc, op = O.model_forward({k: v.cpu() for k, v in sd.items()}, fr[:, :-1].cpu(), O.normalize_actions(ac[:, :-1]).cpu(), cad.cpu(), ocfg)[:2]
    assert U.relerr(pars, op) < 1e-4 and U.relerr(cmds, oc) < 1e-4
