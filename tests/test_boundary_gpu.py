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
    os.chdir(tmp_path)
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


def test_bf16_trainer_runs_and_sequential_inference(tmp_path):
    model, tr = make("bf16", tmp_path)
    model.train()           # dropout 0.1 active: the train loop of the reference
    batch = synth.make_batch_torch(2, 6, 5, "cpu")
    l0, _ = tr._process_batch(batch)
    for _ in range(3):
        l1, m = tr._process_batch(batch)
    assert torch.isfinite(l1) and float(l1) < float(l0) + 1e-3          # lr 1e-5: loss must not blow up
    model.eval()
    b = synth.make_batch_torch(1, 3, 6, DEV)
    cmds, pars = model.sequential_inference(b["frames"][:, :3], b["cad_image"], action=True)
    assert cmds.shape == (1, 3, 5) and pars.shape == (1, 3, 6, 1000)
    with torch.no_grad():
        full = model({"frames": b["frames"][:, :3], "actions": torch.zeros(1, 3, 7, device=DEV), "cad_image": b["cad_image"]})
    c0, p0 = model.sequential_inference(b["frames"][:, :3], b["cad_image"], action=False)
    assert U.relerr(p0, full[1]) < 2e-2                                   # causal model: prefix runs agree with the full run
