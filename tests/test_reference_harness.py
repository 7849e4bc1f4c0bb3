"""Drop-in proof on the REFERENCE'S OWN harness: `/root/reference/experiment.py` is imported unmodified, with
exactly the two imports a maintainer would swap —

    from model.model_factory import ModelFactory     ->  videocad_amd.model_factory
    from trainer import create_trainer               ->  videocad_amd.trainer

— and `Experiment.run_experiment_with_params` (reference experiment.py:75-125: factory -> optional torch.compile -> .to(device) /
DDP wrap -> create_trainer -> train -> evaluate -> results.json) runs end to end on a two-batch synthetic loader, the kernels under
the CPU emulator (build container only: the reference's Python never travels to the GPU box, so the test skips where
/root/reference is absent).  The numbers it produces are checked against the oracle.

What this pins: constructor kwargs straight from the reference's JSON entry (extra keys tolerated), `create_trainer` signature and
packets, `trainer.train(epochs)` returning the model, `trainer.evaluate(model)` returning a JSON-serialisable dict, the files the
harness writes (params.json, training_config.json, results.json), and that a `torch.compile` wrapper around the model is harmless
(the trainer drives the native module's engine; the wrapper's forward is never traced).  The DDP branch (:94-109) needs CUDA
(`device_ids=[0]`) and is covered by `trainer.unwrap` + the GradSync gloo tests instead."""
import importlib
import json
import os
import shutil
import sys
import types

import pytest
import torch

import oputil as U
from oracle import restatement as O
from videocad_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "experiment.py")), reason="the reference lives in the build container only")


@pytest.fixture
def reference_experiment(monkeypatch):
    """`experiment` of the reference, imported with videocad_amd standing in for `model.model_factory` and `trainer`."""
    import transformers  # noqa: F401   (must precede the torchvision stub)
    import videocad_amd.model_factory as vmf
    import videocad_amd.trainer as vtr
    saved = dict(sys.modules)
    monkeypatch.syspath_prepend(os.path.join(HERE, "golden", "stubs"))
    monkeypatch.syspath_prepend(REF)
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    pkg = types.ModuleType("model"); pkg.__path__ = []          # the swapped package: only model.model_factory is asked for
    pkg.model_factory = vmf
    sys.modules.update({"model": pkg, "model.model_factory": vmf, "trainer": vtr})
    sys.modules.pop("experiment", None)
    exp = importlib.import_module("experiment")
    assert exp.ModelFactory is vmf.ModelFactory and exp.create_trainer is vtr.create_trainer
    yield exp
    stubs = os.path.join(HERE, "golden", "stubs")
    for k, m in list(sys.modules.items()):            # drop ONLY what came from the reference tree / the stubs (torch's own lazily imported modules stay)
        f = getattr(m, "__file__", None) or ""
        if k not in saved and (f.startswith(REF) or f.startswith(stubs)):
            del sys.modules[k]
    for k in ("model", "model.model_factory", "trainer", "experiment"):
        if k in saved:
            sys.modules[k] = saved[k]
        else:
            sys.modules.pop(k, None)


@pytest.mark.parametrize("compile_flag", [False, True])
def test_reference_experiment_runs_on_videocad_amd(reference_experiment, tmp_path, monkeypatch, compile_flag):
    exp = reference_experiment
    monkeypatch.chdir(tmp_path)
    shutil.copy(os.path.join(HERE, "golden", "class_weights.json"), "class_weights.json")      # the trainer reads it CWD-relative (reference trainer.py:822)
    # the reference's JSON entry, depth-reduced so the emulator finishes in seconds (vit_depth is this build's test extension)
    params = dict(json.load(open(os.path.join(HERE, "golden", "model_configs.json")))["cad_past_10_actions_and_states_timestep_embedding"])
    params.update(num_decoder_layers=1, window_size=2, max_ep_len=8, vit_depth=1, compute_dtype="f32", dropout=0.0)
    ocfg = dict(O.CANONICAL_CONFIG); ocfg.update(vit_depth=1, num_decoder_layers=1, window_size=2, max_ep_len=8)
    batches = []
    for seed in (21, 22):
        b = synth.make_batch(1, 2, seed=seed)
        batches.append({k: (torch.from_numpy(v) if v is not None else None) for k, v in b.items()})
    pk = {"loader": batches, "sampler": None}
    tcfg = {"lr": 1e-5, "use_mse": True, "epochs": 1, "compile": compile_flag, "enable_parallel": False, "sequential": False,
            "save_frequency": 1, "val_frequency": 1, "early_stopping_enabled": False}
    with U.emulated():
        ex = exp.Experiment(pk, pk, pk, "cpu", 0, training_config=tcfg, rank=0)
        # same initial weights as the oracle: create_model draws torch's default init, so load the hash-generated set through the
        # harness's own checkpoint path (experiment.py:61-71 `"state_dict": path`)
        shapes = O.param_shapes(ocfg)
        weights = {k: synth.make_param(k, s) for k, s in shapes.items()}
        torch.save({"model_state_dict": {"module._orig_mod." + k: torch.from_numpy(v) for k, v in weights.items()}}, "init.pt")
        params["state_dict"] = "init.pt"
        ex.run_experiment_with_params(params, name="dropin")
    logs = [d for d in os.listdir("logs") if d.startswith("dropin_")]
    assert len(logs) == 1
    d = os.path.join("logs", logs[0])
    for f in ("params.json", "training_config.json", "results.json"):
        assert os.path.exists(os.path.join(d, f)), f
    res = json.load(open(os.path.join(d, "results.json")))
    assert res["total_predictions"] > 0 and "cmd_accuracy" in res and "loss" in res
    assert os.path.exists(os.path.join("checkpoints", logs[0], "epoch_1.pt"))
    # the evaluation the harness stored == the oracle's on the same two batches after the same two optimiser steps
    ot = O.OracleTrainer(weights, ocfg)
    for b in (21, 22):
        ot.step(synth.make_batch(1, 2, seed=b))
    tot = 0.0; correct = 0; total = 0
    for b in (21, 22):
        nb = synth.make_batch(1, 2, seed=b)
        with torch.no_grad():
            cmds, pars, tgt = ot.forward(nb)
            loss, m = O.compute_loss(cmds, pars, tgt)
        tot += float(loss); correct += m["correct_predictions"]; total += m["total_predictions"]
    assert abs(res["loss"] - tot / 2) < 1e-4 * abs(tot / 2), (res["loss"], tot / 2)
    assert res["correct_predictions"] == correct and res["total_predictions"] == total
