#!/usr/bin/env python
"""bench.py — training frames/s of the VideoCAD behaviour-cloning step on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: one rank per GPU over RCCL — under torch.distributed.run the ranks
                                                          come from the launcher; run directly, bench.py spawns its N ranks itself)

One step = videocad_amd.trainer.BaseTrainer.train_step (what _process_batch calls) on one synthetic loader-shaped batch resident in HBM:
forward -> fused loss -> backward -> (gradient all-reduce over RCCL, overlapped, N > 1) -> clip(1.0) -> Adam.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seq", type=int, default=64)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32", "bf16x3"],
                    help="bf16 = throughput mode (headline, BASELINE's dtype); f16 = the same kernels on fp16 storage (libvcad_hip_f16.so: in tolerance); "
                         "bf16x3 = fp32 tensors, hi/lo-split bf16 MFMAs (in tolerance); f32 = exact fp32 MFMA")
    ap.add_argument("--dropout", type=float, default=0.1, help="train-mode dropout probability (reference default 0.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-seq186", action="store_true", help="skip the extra seq_len=186 measurement reported beside the headline config")
    ap.add_argument("--gemm-dma", type=int, default=-1, help="A/B switch: 0 = register-staged GEMM only, -1 = automatic (default)")
    ap.add_argument("--gemm-mid", type=int, default=-1, help="A/B: 0 = no six-stage DMA-ring kernel for mid-size GEMMs, -1 = automatic")
    ap.add_argument("--fp8", action="store_true", help="VCAD_FP8 forward mode: the ViT Linear layers on the MXFP8 matrix cores (not the headline number)")
    ap.add_argument("--no-side", type=int, default=0, help="A/B: 1 = no library side stream (CAD ViT and deferred wgrads serialised on the caller's stream)")
    ap.add_argument("--gemm-wide", type=int, default=-1, help="A/B switch: 0 = 256x128 tile only, -1 = automatic (default)")
    ap.add_argument("--uint8-frames", action="store_true", help="feed uint8 grayscale pixels (normalised inside the patchify kernel) instead of fp32 frames")
    ap.add_argument("--no-modes", action="store_true", help="skip the short bf16x3 / f32 / fp8 legs reported beside the headline (each with its parity block)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive (pinned host -> HBM staged) measurement reported beside the headline")
    ap.add_argument("--grad-wire", default="fp32", choices=["fp32", "half"], help="N > 1: gradient buckets on the wire in fp32 (reference) or in the library's 16-bit storage format")
    ap.add_argument("--grad-exchange", default="all_reduce", choices=["all_reduce", "rs_ag", "auto"], help="N > 1: one all_reduce per bucket, or reduce_scatter + all_gather (auto: buckets >= 32 MB)")
    ap.add_argument("--profile-only", action="store_true", help="rocprofv3 runs: only the headline workload's train steps (no seq-186 / PCIe / parity / CPU-baseline legs)")
    args = ap.parse_args(argv)
    if args.profile_only:
        args.no_cpu_baseline = args.no_seq186 = args.no_pcie = args.no_parity = args.no_modes = True

    from videocad_amd.bench_impl import launch
    launch(args)


if __name__ == "__main__":
    main()
