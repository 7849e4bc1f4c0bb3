"""Data parallelism on real devices (SURVEY §8 e / a17): two RCCL ranks, one process per GPU, against the mean of the oracle's
per-rank gradients — the GPU twin of test_two_rank_gloo_data_parallel_step_matches_mean_of_gradients — plus bench.py's own
N-rank launch.  Skipped on a 1-GPU box (the driver's multi-GPU node runs them)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oputil as U
from oracle import restatement as O
from videocad_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
CANON = json.load(open(os.path.join(HERE, "golden", "model_configs.json")))["cad_past_10_actions_and_states_timestep_embedding"]
PROBES = ["predict_action_class_0_999.weight", "transformer_decoder.layers.3.linear1.weight", "embed_state.weight",
          "state_embedding_model.transformer.layers.2.0.to_qkv.weight", "cad_embedding_model.to_patch_embedding.2.weight",
          "state_embedding_model.to_patch_embedding.2.weight"]
two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL over xGMI)")


def _rank_main(rank, world, port, tmp, q):
    import torch.distributed as dist
    from videocad_amd.model_factory import ModelFactory
    from videocad_amd.trainer import create_trainer
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    os.chdir(tmp)
    dev = torch.device(f"cuda:{rank}")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        torch.manual_seed(100 + rank)                           # ranks start from DIFFERENT random weights; rank 0 then loads the known ones
        model, mtype = ModelFactory().create_model("autoregressive", dict(CANON, compute_dtype="f32"), dev)
        if rank == 0:
            model.load_state_dict({k: synth.make_param_torch(k, s, dev) for k, s in O.param_shapes().items()}, strict=True)
        model.eval()                                            # deterministic step (dropout has its own tests)
        pk = {"loader": [], "sampler": None}
        tr = create_trainer(pk, pk, pk, model, {"lr": 1e-5, "use_mse": True, "experiment_name": f"r{rank}",
                                                "class_weights_path": os.path.join(HERE, "golden", "class_weights.json")}, dev, mtype, rank=rank)
        assert tr.gradsync.world == world and tr.gradsync.stream is not None
        batch = synth.make_batch_torch(1, 2, 10 + rank, "cpu")
        loss, _ = tr._process_batch(batch)
        torch.cuda.synchronize()
        if rank == 0:
            q.put({n: dict(model.named_parameters())[n].detach().cpu().numpy() for n in PROBES})
        dist.barrier()
    finally:
        dist.destroy_process_group()


@two_gpus
def test_two_rank_rccl_step_matches_mean_of_oracle_gradients(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    [p.start() for p in procs]
    got = q.get(timeout=900)
    [p.join(timeout=300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    shapes = O.param_shapes()
    ot = O.OracleTrainer({k: synth.make_param(k, s) for k, s in shapes.items()})
    gsum = None
    for r in range(2):
        ot.loss_and_grads(synth.make_batch(1, 2, seed=10 + r))
        g = {k: p.grad.clone() for k, p in ot.P.items()}
        gsum = g if gsum is None else {k: gsum[k] + g[k] for k in g}
    gavg = {k: v / 2 for k, v in gsum.items()}
    ot.apply_grads(gavg)
    for n in PROBES:
        diff = np.abs(got[n] - ot.P[n].detach().numpy())
        sig = np.abs(gavg[n].numpy()) > 1e-6                     # Adam's first step is ill-conditioned where the gradient is numerical noise
        assert diff[sig].max() < 2e-6, (n, float(diff[sig].max()))
        assert diff.max() <= 2.1e-5, (n, float(diff.max()))


def _bench(extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--seq", "8",
                          "--no-cpu-baseline", "--no-seq186", "--no-pcie"] + extra, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_line_single_gpu_contract():
    j = _bench([])
    assert j["n_gpus"] == 1 and j["unit"] == "frames/s" and j["value"] > 0 and j["config"]["step"].endswith("train_step")
    assert j["roofline"]["bound"] == "mfma" and 0 < j["roofline"]["frac"] < 1
    assert j["kernel_breakdown"]["attention"]["GBps"] and j["parity"]["argmax_agreement"] > 0.85 and j["parity"]["logit_mae"] < 5e-2


@two_gpus
def test_bench_gpus_2_starts_two_rccl_ranks():
    j = _bench(["--gpus", "2"])
    assert j["n_gpus"] == 2 and j["comm"]["rccl_ranks"] == 2 and j["config"]["global_batch"] == 4 and j["scaling"] == "weak"
