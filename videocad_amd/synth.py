"""Deterministic integer-hash generators for weights and loader-shaped batches.

Nothing large is ever committed: weights and inputs are regenerated bit-identically
from (name, flat index) / (seed, flat index) on any machine (this container, the GPU
box), following SURVEY.md §8(c)/(d).  splitmix64 -> top 24 bits -> fp32 in [-1, 1).

The batch layout reproduces the collated tensor contract of the reference loader
(reference data_loader/data_loader.py:313-366, trainer.py:291-324):
  frames  [B, S, 1, 224, 224] f32 in [-1, 1)     (S = T + 1)
  actions [B, S, 7] f32: cmd in {0..4}, params in {0..999} at the slots the command
          mask allows (reference trainer.py:258-264), -1 elsewhere, row 0 all zeros
          (reference generate_dataset.py:180-182); padded rows are -1 everywhere
  cad_image [B, 1, 224, 224] f32, timesteps [B, S] i64
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_u64(key: int, n: int, offset: int = 0) -> np.ndarray:
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + np.uint64(key & 0xFFFFFFFFFFFFFFFF))


def hash_uniform(key: int, n: int, offset: int = 0) -> np.ndarray:
    """fp32 uniform in [-1, 1), exactly representable (24-bit mantissa grid)."""
    u = hash_u64(key, n, offset) >> np.uint64(40)          # 24 bits
    return (u.astype(np.float32) * np.float32(2.0 ** -23)) - np.float32(1.0)


def hash_randint(key: int, n: int, high: int) -> np.ndarray:
    return (hash_u64(key, n) >> np.uint64(11)) % np.uint64(high)


# ----------------------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------------------

def init_scale(name: str, shape) -> tuple[float, float]:
    """(scale, shift) for a parameter: value = shift + scale * U[-1,1).

    Mirrors the magnitudes of the PyTorch defaults the reference relies on (SURVEY.md
    Appendix A 'Init'): Linear ~ U(+-1/sqrt(fan_in)); LayerNorm weight ~ 1, bias ~ 0 (perturbed
    so that gamma/beta code paths are exercised); embeddings / cls / pos O(1).  The two head
    matrices are scaled up so top-1/top-2 logit gaps are comfortably above fp32 noise."""
    leaf = name.rsplit(".", 1)[-1]
    if "norm" in name or "to_patch_embedding.1." in name or "to_patch_embedding.3." in name \
            or name.endswith("net.0.weight") or name.endswith("net.0.bias"):
        return (0.1, 1.0) if leaf == "weight" else (0.05, 0.0)
    if name.endswith("pos_embedding") or name.endswith("cls_token"):
        return (1.0, 0.0)
    if name.startswith("timestep_embedding"):
        return (1.0, 0.0)
    if len(shape) == 2:
        fan_in = shape[1]
        s = 1.0 / np.sqrt(fan_in)
        if name.startswith("predict_action_class"):
            s *= 4.0
        return (float(s), 0.0)
    return (0.02, 0.0)   # biases


def make_param(name: str, shape, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    scale, shift = init_scale(name, shape)
    u = hash_uniform(fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF), n)
    return (np.float32(shift) + np.float32(scale) * u).astype(np.float32).reshape(shape)


# ----------------------------------------------------------------------------------------
# batches
# ----------------------------------------------------------------------------------------

ACTION_MASK = np.array([[1, 1, 0, 0, 0, 0],
                        [0, 0, 1, 1, 0, 0],
                        [0, 0, 0, 0, 1, 0],
                        [0, 0, 0, 0, 0, 1],
                        [0, 0, 0, 0, 0, 0]], dtype=np.int64)   # reference trainer.py:258-264


def make_actions(B: int, S: int, seed: int, lengths=None) -> np.ndarray:
    key = fnv1a64("actions") ^ (seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
    cmd = hash_randint(key, B * S, 5).astype(np.int64).reshape(B, S)
    par = hash_randint(key ^ 0x5555, B * S * 6, 1000).astype(np.int64).reshape(B, S, 6)
    mask = ACTION_MASK[cmd]                                    # [B,S,6]
    par = np.where(mask == 1, par, -1)
    act = np.concatenate([cmd[..., None], par], axis=-1).astype(np.float32)
    act[:, 0, :] = 0.0                                         # first row all zeros
    if lengths is not None:
        for b, L in enumerate(lengths):
            act[b, L:, :] = -1.0                               # collate_with_padding pad value
    return act


def make_batch(B: int, T: int, seed: int, lengths=None, img: int = 224, num_views: int = 0) -> dict:
    """Loader-shaped batch (numpy). `lengths` (per-clip valid S<=T+1) gives ragged clips padded with -1.
    num_views > 0 adds 'multiview_images' [B, V, 1, img, img] (reference data loader contract, autoregressive_transformer.py:131)."""
    S = T + 1
    kf = fnv1a64("frames") ^ (seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
    kc = fnv1a64("cad_image") ^ (seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
    frames = hash_uniform(kf, B * S * img * img).reshape(B, S, 1, img, img)
    if lengths is not None:
        for b, L in enumerate(lengths):
            frames[b, L:] = -1.0
    cad = hash_uniform(kc, B * img * img).reshape(B, 1, img, img)
    return {
        "frames": frames,
        "actions": make_actions(B, S, seed, lengths),
        "cad_image": cad,
        "timesteps": np.tile(np.arange(S, dtype=np.int64), (B, 1)),
        "multiview_images": (hash_uniform(fnv1a64("multiview_images") ^ (seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF), B * num_views * img * img)
                             .reshape(B, num_views, 1, img, img) if num_views > 0 else None),
    }


# ----------------------------------------------------------------------------------------
# torch twins (bit-identical to the numpy generators; run on any device, used on the GPU box)
# ----------------------------------------------------------------------------------------

def _s64(x: int) -> int:
    x &= 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(x, s: int):
    import torch
    return (x >> s) & ((1 << (64 - s)) - 1)


def hash_u64_torch(key: int, n: int, device, offset: int = 0):
    """splitmix64 in wrapping int64 arithmetic (logical shifts emulated by masking)."""
    import torch
    idx = torch.arange(offset, offset + n, dtype=torch.int64, device=device)
    x = idx * _s64(0xD1342543DE82EF95) + _s64(key)
    x = x + _s64(0x9E3779B97F4A7C15)
    z = (x ^ _lsr(x, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def hash_uniform_torch(key: int, n: int, device, offset: int = 0):
    import torch
    u = _lsr(hash_u64_torch(key, n, device, offset), 40)
    return u.to(torch.float32) * (2.0 ** -23) - 1.0


def make_param_torch(name: str, shape, device, seed: int = 0):
    import torch
    n = int(np.prod(shape))
    scale, shift = init_scale(name, shape)
    u = hash_uniform_torch(fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF), n, device)
    return (torch.tensor(shift, dtype=torch.float32, device=device) + torch.tensor(scale, dtype=torch.float32, device=device) * u).reshape(tuple(shape))


def make_batch_torch(B: int, T: int, seed: int, device, lengths=None, img: int = 224, num_views: int = 0) -> dict:
    """Same tensors as make_batch, generated on `device` (actions are tiny and come from the numpy path)."""
    import torch
    S = T + 1
    kf = fnv1a64("frames") ^ (seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
    kc = fnv1a64("cad_image") ^ (seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
    frames = hash_uniform_torch(kf, B * S * img * img, device).reshape(B, S, 1, img, img)
    if lengths is not None:
        for b, L in enumerate(lengths):
            frames[b, L:] = -1.0
    cad = hash_uniform_torch(kc, B * img * img, device).reshape(B, 1, img, img)
    return {"frames": frames, "actions": torch.from_numpy(make_actions(B, S, seed, lengths)).to(device), "cad_image": cad,
            "timesteps": torch.arange(S, device=device).repeat(B, 1),
            "multiview_images": (hash_uniform_torch(fnv1a64("multiview_images") ^ (seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF), B * num_views * img * img, device)
                                 .reshape(B, num_views, 1, img, img) if num_views > 0 else None)}
