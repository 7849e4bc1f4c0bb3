"""Drop-in boundary on the CPU: state_dict contract, ModelFactory / trainer surface, loud failure
without a GPU, and — with the kernels running under the emulator — the autograd bridge and the native train step."""
import json
import os

import numpy as np
import pytest
import torch

import oputil as U
from oracle import restatement as O
from videocad_amd import lib as L
from videocad_amd import synth
from videocad_amd.model_factory import ModelFactory, ModelType
from videocad_amd.trainer import create_trainer

HERE = os.path.dirname(os.path.abspath(__file__))
CANON = json.load(open(os.path.join(HERE, "golden", "model_configs.json")))["cad_past_10_actions_and_states_timestep_embedding"]


def test_c_abi_exports_every_declared_symbol():
    import ctypes
    lib = ctypes.CDLL(U.ensure_hip_lib())             # loads without a GPU (no compute calls here); built on demand in a fresh checkout
    L.declare(lib)
    assert b"gfx950" in lib.vcad_version()
    hdr = open(os.path.join(HERE, "..", "include", "vcad.h")).read()
    import re
    product, ab = re.split(r"#ifdef VCAD_AB\n", hdr)[0], re.findall(r"#ifdef VCAD_AB\n(.*?)#endif", hdr, re.S)
    product += re.split(r"#ifdef VCAD_AB\n.*?#endif", hdr, flags=re.S)[1]
    declared = set(re.findall(r"\b(vcad_[a-z0-9_]+)\s*\(", product))
    # the A/B-only selectors (csrc/ab.h) are declared under VCAD_AB and must NOT be in the shipped library; nor any process-global switch
    ab_names = set(re.findall(r"\b(vcad_debug_[a-z0-9_]+)\s*\(", ab[0]))
    assert ab_names == set(L.AB_PROTOTYPES) and not any(n.startswith("vcad_debug") for n in declared)
    assert not any(hasattr(lib, n) for n in ab_names), [n for n in ab_names if hasattr(lib, n)]
    assert declared and all(hasattr(lib, n) for n in declared), [n for n in declared if not hasattr(lib, n)]
    assert declared <= set(L.PROTOTYPES) | {"vcad_config", "vcad_engine"}, declared - set(L.PROTOTYPES)
    # the fp16-storage build of the same sources: the same entry points, and it says which format it stores
    lib16 = L.declare(ctypes.CDLL(L.LIB_PATH_F16))
    assert all(hasattr(lib16, n) for n in declared) and not any(hasattr(lib16, n) for n in ab_names)
    assert lib.vcad_storage_format() == b"bf16" and lib16.vcad_storage_format() == b"f16" and b"f16" in lib16.vcad_version()


def test_state_dict_matches_appendix_b_and_factory_surface():
    model, mtype = ModelFactory().create_model(CANON["model_name"], dict(CANON), "cpu")
    assert mtype == ModelType.MULTI_CLASSES
    sd = model.state_dict()
    shapes = O.param_shapes()
    assert set(sd) == set(shapes)
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    assert sum(v.numel() for v in sd.values()) == 126_963_573           # live parameters
    assert hasattr(model, "state_embedding_model") and hasattr(model, "cad_embedding_model")   # reference trainer.py:244-245
    assert model.transformer_decoder.layers[3].self_attn.in_proj_weight.shape == (3072, 1024)
    # views of one flat buffer
    assert model.embed_state.weight.data_ptr() == model._engine.view("embed_state.weight").data_ptr()
    # load with DDP/compile prefixes + unknown (dead GPT-2) keys, strict=False like the reference factory
    w = {"module._orig_mod.embed_state.bias": torch.full((1024,), 0.5), "transformer.h.0.ln_1.weight": torch.zeros(3)}
    model2, _ = ModelFactory().create_model("x", dict(CANON), "cpu", state_dict=w)
    assert float(model2.embed_state.bias[7]) == 0.5
    with pytest.raises(RuntimeError):
        model2.load_state_dict({"embed_state.bias": torch.zeros(1024)}, strict=True)
    # no CPU fallback: the product path fails loudly without a ROCm device
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model({"frames": torch.zeros(1, 1, 1, 224, 224), "actions": torch.zeros(1, 1, 7), "cad_image": torch.zeros(1, 1, 224, 224)})
    # a checkpoint in the pre-1.2 vit-pytorch layout (PreNorm wrappers; the reference pins no version) is different arithmetic: refused, not silently ignored by strict=False
    legacy = {"module.state_embedding_model.transformer.layers.0.0.fn.to_qkv.weight": torch.zeros(3072, 512), "embed_state.bias": torch.zeros(1024)}
    with pytest.raises(RuntimeError, match="pre-1.2 vit-pytorch layout"):
        ModelFactory().create_model("x", dict(CANON), "cpu", state_dict=legacy)
    with pytest.raises(RuntimeError, match="pre-1.2 vit-pytorch layout"):
        model2.load_state_dict({"cad_embedding_model.to_patch_embedding.1.weight": torch.zeros(512, 1024)}, strict=False)
    model2.load_state_dict({"cad_embedding_model.to_patch_embedding.1.weight": torch.ones(1024), "state_embedding_model.transformer.layers.2.0.norm.weight": torch.ones(512)}, strict=False)   # the current layout's keys pass
    for bad in (dict(CANON, encoder="resnet"), dict(CANON, num_views=2, enable_past_actions=False), dict(CANON, window_size=0)):   # (views + states without actions: the reference's own shapes do not match)
        with pytest.raises((NotImplementedError, AssertionError)):
            ModelFactory().create_model("x", bad, "cpu")


def small_model(emu, **over):
    cfg = dict(CANON); cfg.update(num_decoder_layers=1, window_size=2, max_ep_len=8, compute_dtype="f32", vit_depth=1); cfg.update(over)
    ocfg = dict(O.CANONICAL_CONFIG); ocfg.update(vit_depth=1, num_decoder_layers=1, window_size=2, max_ep_len=8)
    return cfg, ocfg


@pytest.fixture
def emu():
    with U.emulated() as e:
        yield e


def _cwd_with_class_weights(tmp_path):
    import shutil
    shutil.copy(os.path.join(HERE, "golden", "class_weights.json"), os.path.join(str(tmp_path), "class_weights.json"))
    os.chdir(str(tmp_path))


def test_autograd_bridge_and_trainer_step_under_emulator(emu, tmp_path, monkeypatch):
    """model(inputs) -> torch loss -> loss.backward() gives the oracle's gradients; trainer._process_batch == oracle step."""
    monkeypatch.chdir(tmp_path)
    _cwd_with_class_weights(tmp_path)
    # depth-reduced model (vit_depth=1, 1 decoder layer), B=1, T=2: the emulator executes every lane as a fiber
    cfg, ocfg = small_model(emu)
    model, mtype = ModelFactory().create_model("autoregressive", cfg, "cpu")
    shapes = O.param_shapes(ocfg)
    weights = {k: synth.make_param(k, s) for k, s in shapes.items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=True)
    model.eval()                                   # parity with the (eval-mode) oracle; train-mode dropout is tested with explicit masks
    batch = synth.make_batch(1, 2, seed=4)
    tb = {k: (torch.from_numpy(v) if v is not None else None) for k, v in batch.items()}
    pk = {"loader": [tb], "sampler": None}
    tr = create_trainer(pk, pk, pk, model, {"lr": 1e-5, "use_mse": True, "experiment_name": "t"}, "cpu", mtype, rank=0)

    ot = O.OracleTrainer(weights, ocfg)
    oloss, ometrics, ocmds, opars = ot.loss_and_grads(batch)
    # (1) reference-style sequence: model(inputs) -> compute_loss -> backward, gradients through the autograd bridge
    bd = tr.prepare_batch(tb)
    preds = model(tr._prepare_model_inputs(bd, False))
    assert U.relerr(preds[1], opars) < 1e-5
    loss, metrics = tr.compute_loss(preds, bd["actions"][:, 1:])
    assert metrics == ometrics
    loss.backward()
    worst = max((U.relerr(p.grad, ot.P[n].grad), n) for n, p in model.named_parameters() if float(ot.P[n].grad.norm()) > 0)
    assert worst[0] < 2e-4, worst
    # (2) fused native step == oracle step
    total = ot.apply_grads({k: p.grad for k, p in ot.P.items()})
    loss2, metrics2 = tr._process_batch(tb)
    assert abs(float(loss2) - float(oloss)) < 1e-5 * abs(float(oloss)) and metrics2 == ometrics
    d = max(float((p.detach() - ot.P[n].detach()).abs().max()) for n, p in model.named_parameters())
    assert d < 2e-6, d


def test_multiview_model_through_factory_and_trainer_under_emulator(emu, tmp_path, monkeypatch):
    """The multiview branch at the reference's surface: `num_views` in the JSON entry, `multiview_images` in the loader's batch (reference
    trainer.py:320-322,515-516) -> create_model / create_trainer / model(inputs) / _process_batch; numbers against the oracle."""
    monkeypatch.chdir(tmp_path)
    _cwd_with_class_weights(tmp_path)
    cfg, ocfg = small_model(emu, num_views=2)
    ocfg["num_views"] = 2
    model, mtype = ModelFactory().create_model("autoregressive", cfg, "cpu")
    assert model.num_views == 2 and model.num_inputs == 3 and tuple(model.embed_multiview.weight.shape) == (1024, 1024)
    weights = {k: synth.make_param(k, s) for k, s in O.param_shapes(ocfg).items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=True)
    model.eval()
    batch = synth.make_batch(1, 2, seed=14, num_views=2)
    tb = {k: (torch.from_numpy(v) if v is not None else None) for k, v in batch.items()}
    pk = {"loader": [tb], "sampler": None}
    tr = create_trainer(pk, pk, pk, model, {"lr": 1e-5, "use_mse": True, "experiment_name": "t"}, "cpu", mtype, rank=0)
    ot = O.OracleTrainer(weights, ocfg)
    oloss, ometrics, ocmds, opars = ot.loss_and_grads(batch)
    bd = tr.prepare_batch(tb)
    inputs = tr._prepare_model_inputs(bd, False)
    assert inputs["multiview_images"].shape == (1, 2, 1, 224, 224)
    preds = model(inputs)
    assert U.relerr(preds[1], opars) < 1e-5
    with pytest.raises(RuntimeError, match="multiview"):
        model({k: v for k, v in inputs.items() if k != "multiview_images"})
    loss, metrics = tr.compute_loss(preds, bd["actions"][:, 1:])
    loss.backward()
    worst = max((U.relerr(p.grad, ot.P[n].grad), n) for n, p in model.named_parameters() if float(ot.P[n].grad.norm()) > 0)
    assert worst[0] < 2e-4, worst
    ot.apply_grads({k: p.grad for k, p in ot.P.items()})
    loss2, metrics2 = tr._process_batch(tb)
    assert abs(float(loss2) - float(oloss)) < 1e-5 * abs(float(oloss)) and metrics2 == ometrics
    d = max(float((p.detach() - ot.P[n].detach()).abs().max()) for n, p in model.named_parameters())
    assert d < 2e-6, d
    with pytest.raises(RuntimeError, match="multiview"):
        model.sequential_inference(bd["frames"][:, :-1], bd["cad_image"])          # as in the reference, inference has no multiview input


def _q_get(q, procs, timeout=900):
    """q.get that gives up as soon as a worker has died (instead of sitting out the whole timeout)"""
    import queue, time
    t0 = time.time()
    while time.time() - t0 < timeout:
        try:
            return q.get(timeout=2)
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                raise AssertionError(f"a worker died: exit codes {[p.exitcode for p in procs]}")
    raise AssertionError("workers timed out")


def _ddp_worker(rank, world, port, tmp, q, wrap=False, dtype="f32", extra=None, seed0=10):
    import torch.distributed as dist
    _cwd_with_class_weights(tmp)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        L._lib = U.load_emu()                        # this process only ever runs the emulator build
        cfg, ocfg = small_model(None)
        if dtype == "f16":
            L._lib_f16 = U.load_emu("f16"); cfg = dict(cfg, compute_dtype="f16")
        torch.manual_seed(1234 + rank)               # every rank draws DIFFERENT initial weights (the reference seeds nothing) ...
        model, mtype = ModelFactory().create_model("autoregressive", cfg, "cpu")
        shapes = O.param_shapes(ocfg)
        if rank == 0:                                 # ... and only rank 0 holds the weights the check expects
            model.load_state_dict({k: torch.from_numpy(synth.make_param(k, s)) for k, s in shapes.items()}, strict=True)
        model.eval()
        native = model
        calls = {"n": 0, "elems": 0}
        if wrap:
            # exactly what reference experiment.py:104-109 does to the model before create_trainer (device_ids only exist on a GPU)
            from torch.nn.parallel import DistributedDataParallel
            model = DistributedDataParallel(model, find_unused_parameters=True)
            real = dist.all_reduce
            def counted(t, *a, **k):
                calls["n"] += 1; calls["elems"] += t.numel()
                return real(t, *a, **k)
            dist.all_reduce = counted
        batch = synth.make_batch(1, 2, seed=seed0 + rank)
        tb = {k: (torch.from_numpy(v) if v is not None else None) for k, v in batch.items()}
        pk = {"loader": [tb], "sampler": None}
        tr = create_trainer(pk, pk, pk, model, dict({"lr": 1e-5, "use_mse": True, "experiment_name": "t"}, **(extra or {})), "cpu", mtype, rank=rank)
        assert tr.gradsync.world == world
        assert native._engine.gemm_flags & L.GEMM_DYNAMIC, "data-parallel runs draw the persistent GEMM's items with tickets (RCCL holds CUs)"
        # GradSync's constructor broadcast rank 0's parameters (what the DDP wrap did in the reference, experiment.py:104-109)
        w0 = torch.from_numpy(synth.make_param("embed_state.weight", shapes["embed_state.weight"]))
        assert torch.equal(native.embed_state.weight.detach(), w0), "initial parameters were not broadcast from rank 0"
        calls["n"] = calls["elems"] = 0
        tr._process_batch(tb)
        if wrap:
            # ONE gradient exchange per step: GradSync's collectives cover the flat gradient buffer exactly once, and the DDP wrapper's own reducer
            # never ran (its forward is never entered, so no autograd hook fired: no .grad was produced on any parameter)
            eng = native._engine
            assert tr.native is native and calls["n"] == len(eng.buckets), calls
            assert calls["elems"] == eng.buckets[-1][1] - eng.buckets[0][0], calls
            assert all(p.grad is None for p in model.parameters())
            if rank == 0:                                                       # the summed gradient itself (Adam's first step hides a doubled gradient)
                q.put({"__grads__": {n: eng.view(n, eng.grads).detach().clone().numpy() for n in shapes}})
        if extra and rank == 0:                      # exchange variants: the summed gradient buffer itself + how many collectives carried it
            eng = native._engine
            q.put({"__flat__": eng.grads.detach().clone().numpy(), "collectives": tr.gradsync.collectives, "wire": str(eng.wire_dtype),
                   "plans": [tr.gradsync.plan(lo, hi)[1] for lo, hi in eng.buckets], "buckets": list(eng.buckets)})
        elif dtype == "f16" and rank == 0:
            eng = native._engine
            q.put({"__grads__": {n: eng.view(n, eng.grads).detach().clone().numpy() for n in shapes}, "scale": eng.grad_scale})
        elif rank == 0:
            q.put({n: p.detach().clone().numpy() for n, p in native.named_parameters()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_ddp_wrapped_model_does_one_gradient_exchange(tmp_path):
    """`north_star`: "drops into experiment.py unchanged" — so the wrap at reference experiment.py:104-109
    (`DistributedDataParallel(model, find_unused_parameters=True)`) WILL sit around the native module (whose parameters are views of one flat
    buffer).  `create_trainer` unwraps it; the step's only gradient exchange is GradSync's (one collective sweep over the flat buffer — DDP's
    reducer never runs because its forward is never entered), and the summed gradient / the updated weights equal the oracle's mean-gradient step."""
    import torch.multiprocessing as mp
    U.load_emu()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, str(tmp_path), q, True)) for r in range(world)]
    [p.start() for p in procs]
    first = _q_get(q, procs); second = _q_get(q, procs)
    [p.join(timeout=300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    grads, got = (first["__grads__"], second) if "__grads__" in first else (second["__grads__"], first)
    _, ocfg = small_model(None)
    shapes = O.param_shapes(ocfg)
    ot = O.OracleTrainer({k: synth.make_param(k, s) for k, s in shapes.items()}, ocfg)
    gsum = None
    for r in range(world):
        ot.loss_and_grads(synth.make_batch(1, 2, seed=10 + r))
        g = {k: p.grad.clone() for k, p in ot.P.items()}
        gsum = g if gsum is None else {k: gsum[k] + g[k] for k in g}
    # the flat buffer holds the SUM over ranks (1/world is folded into the Adam kernel): a second, redundant all-reduce would show as 2x here
    worst = max((U.relerr(torch.from_numpy(grads[n]), gsum[n]), n) for n in grads if float(gsum[n].norm()) > 1e-6)
    assert worst[0] < 2e-4, worst
    ot.apply_grads({k: v / world for k, v in gsum.items()})
    d = max((float(np.abs(got[n] - ot.P[n].detach().numpy())[np.abs(gsum[n].numpy()) > 2e-6].max(initial=0.0)), n) for n in got)
    assert d[0] < 2e-6, d


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_data_parallel_step_matches_mean_of_gradients(tmp_path, world):
    """world_size = 2 and 4 over gloo on the CPU (kernels under the emulator): rank-local backward, bucketed all-reduce(SUM) of
    the flat gradient buffer, 1/world folded into Adam  ==  clip+Adam on the mean of the two ranks' oracle gradients
    (what DDP does at reference experiment.py:104-109)."""
    import torch.multiprocessing as mp
    U.load_emu()                                    # build once before forking
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    [p.start() for p in procs]
    got = _q_get(q, procs)
    [p.join(timeout=300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    _, ocfg = small_model(None)
    shapes = O.param_shapes(ocfg)
    weights = {k: synth.make_param(k, s) for k, s in shapes.items()}
    ot = O.OracleTrainer(weights, ocfg)
    gsum = None
    for r in range(world):
        ot.loss_and_grads(synth.make_batch(1, 2, seed=10 + r))
        g = {k: p.grad.clone() for k, p in ot.P.items()}
        gsum = g if gsum is None else {k: gsum[k] + g[k] for k in g}
    gavg = {k: v / world for k, v in gsum.items()}
    ot.apply_grads(gavg)
    # Adam's first step is lr * g / (|g| + eps): elements whose gradient is numerical noise around zero (e.g. the key
    # bias of every attention: softmax is shift-invariant, so its true gradient is 0) are ill-conditioned by
    # construction — in the reference too — and are only required to move by at most lr.
    worst_sig, worst_noise = ("", 0.0), ("", 0.0)
    for n in got:
        diff = np.abs(got[n] - ot.P[n].detach().numpy())
        sig = np.abs(gavg[n].numpy()) > 1e-6
        if sig.any() and diff[sig].max() > worst_sig[1]:
            worst_sig = (n, float(diff[sig].max()))
        if (~sig).any() and diff[~sig].max() > worst_noise[1]:
            worst_noise = (n, float(diff[~sig].max()))
    assert worst_sig[1] < 2e-6, worst_sig
    assert worst_noise[1] <= 2.1e-5, worst_noise


def test_gloo_data_parallel_step_on_the_fp16_build(tmp_path):
    """two ranks over gloo, kernels of the fp16-storage build under the emulator: what the collectives sum is the TRUE gradient of each rank (every bucket is
    divided by the gradient scale before it is handed over), so the flat buffer holds the sum of the two ranks' oracle gradients at fp16 accuracy"""
    import torch.multiprocessing as mp
    U.load_emu(); U.load_emu("f16")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 11
    # Batch seeds 20 / 21 (the fp32 twins use 10 / 11): these batches have TWO token rows, so one ReLU unit of the decoder's feed-forward whose pre-activation
    # lies within fp16 rounding of zero flips between engine and oracle and moves every gradient upstream of it by ~1e-2 — a property of the two-row batch, not
    # of the exchange under test (seed 11 hits such a unit since r06 moved the rounding points of the last ViT layer: unit 416 of linear1, gradient 0 vs 0.069,
    # every other element within 1e-4; seed 12 flips another one).  20 / 21 have no unit that close to zero.
    SEED0 = 20
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, str(tmp_path), q, False, "f16", None, SEED0)) for r in range(world)]
    [p.start() for p in procs]
    got = _q_get(q, procs)
    [p.join(timeout=300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert got["scale"] == 1024.0                       # automatic gradient scale: 2 x pow2ceil(B * T), floor 1024 (two rows here; 4096 at the benchmark's 2 048)
    _, ocfg = small_model(None)
    shapes = O.param_shapes(ocfg)
    ot = O.OracleTrainer({k: synth.make_param(k, s) for k, s in shapes.items()}, ocfg)
    gsum = None
    for r in range(world):
        ot.loss_and_grads(synth.make_batch(1, 2, seed=SEED0 + r))
        g = {k: p.grad.clone() for k, p in ot.P.items()}
        gsum = g if gsum is None else {k: gsum[k] + g[k] for k in g}
    errs = sorted((U.relerr(torch.from_numpy(got["__grads__"][n]), gsum[n]), n) for n in gsum if float(gsum[n].norm()) > 1e-6)
    assert errs[len(errs) // 2][0] < 4e-3 and errs[-1][0] < 0.05, (errs[len(errs) // 2], errs[-1])


def _run_exchange(tmp_path, world, dtype, extra, tag):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 40 + tag
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, str(tmp_path), q, False, dtype, extra)) for r in range(world)]
    [p.start() for p in procs]
    got = _q_get(q, procs)
    [p.join(timeout=300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return got


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_gloo_exchange_variants_bound_the_change_in_the_summed_gradient(tmp_path, dtype):
    """r05: the two knobs of GradSync on a 2-rank gloo group, kernels under the emulator — against the default exchange
    (fp32 buckets, one all_reduce(SUM) each: what the reference's DDP does, experiment.py:104-109) on the same ranks and batches:
      grad_exchange = "rs_ag": reduce_scatter + all_gather of every bucket — the SAME sums (two addends per element: bit-identical);
      grad_wire = "half": buckets travel in the library's 16-bit storage format — bf16 for the bf16 / f32 library (8 significant bits per addend
        and for the sum), fp16 with a power-of-two scale from the all-reduced max |g| for the fp16 library (11 bits): the norm-wise change of the
        summed gradient buffer is bounded at 6e-3 / 8e-4, per bucket too; with "auto" only buckets of at least grad_rs_min_mb take rs_ag."""
    U.load_emu()
    if dtype == "f16":
        U.load_emu("f16")
    base = _run_exchange(tmp_path, 2, dtype, {"grad_wire": "fp32", "grad_exchange": "all_reduce", "experiment_name": "x0"}, 0)
    nb = len(base["plans"])
    assert base["collectives"] == nb and set(base["plans"]) == {"all_reduce"}
    ref = torch.from_numpy(base["__flat__"])
    assert float(ref.norm()) > 0
    if dtype == "f32":                       # (the exchange form does not depend on the engine's dtype: once is enough)
        rs = _run_exchange(tmp_path, 2, dtype, {"grad_wire": "fp32", "grad_exchange": "rs_ag", "experiment_name": "x1"}, 1)
        assert rs["collectives"] == 2 * nb and set(rs["plans"]) == {"rs_ag"}
        assert torch.equal(torch.from_numpy(rs["__flat__"]), ref)
    half = _run_exchange(tmp_path, 2, dtype, {"grad_wire": "half", "grad_exchange": "auto", "grad_rs_min_mb": 20.0, "experiment_name": "x2"}, 2)
    assert half["wire"] == ("torch.float16" if dtype == "f16" else "torch.bfloat16")
    assert "rs_ag" in half["plans"] and "all_reduce" in half["plans"]          # the big bucket takes rs_ag, the small ones all_reduce
    extra = nb if dtype == "f16" else 0                                          # (fp16: one 4-byte MAX all-reduce per bucket fixes its scale)
    assert half["collectives"] == nb + half["plans"].count("rs_ag") + extra
    got = torch.from_numpy(half["__flat__"])
    tol = 8e-4 if dtype == "f16" else 6e-3
    assert U.relerr(got, ref) < tol, U.relerr(got, ref)
    assert bool(torch.isfinite(got).all())
    # per bucket as well: a small-magnitude bucket must not drown in a scale chosen for a large one (the scale is per exchanged range)
    for lo, hi in half["buckets"]:
        if float(ref[lo:hi].norm()) > 0:
            assert U.relerr(got[lo:hi], ref[lo:hi]) < 2 * tol, (lo, hi, U.relerr(got[lo:hi], ref[lo:hi]))
