"""Host side of the input path: the collate twin against a literal restatement of the reference's
pad / stack semantics (reference data_loader/data_loader.py:313-366), PIL-exact grayscale, the uint8 <-> fp32 equivalence the
patchify kernel relies on, and the staging iterator on the CPU."""
import numpy as np
import pytest
import torch

from videocad_amd import data as D


def _items(lengths, u8, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for n in lengths:
        fr = torch.randint(0, 256, (n, 1, 8, 8), generator=g, dtype=torch.uint8)
        cad = torch.randint(0, 256, (1, 8, 8), generator=g, dtype=torch.uint8)
        out.append({"frames": fr if u8 else D.normalize_u8(fr), "actions": torch.randint(0, 999, (n, 7), generator=g).float(),
                    "cad_image": cad if u8 else D.normalize_u8(cad), "timesteps": torch.arange(n) + 5, "multiview_images": None})
    return out


def _reference_collate(batch):
    """the reference's algorithm, literally: pad each tensor with -1 by cat, fresh arange timesteps, stack"""
    mx = max(it["frames"].shape[0] for it in batch)

    def pad(a):
        n = mx - a.shape[0]
        return torch.cat([a, torch.full((n, *a.shape[1:]), -1, dtype=a.dtype)], 0) if n > 0 else a
    return {"frames": torch.stack([pad(it["frames"]) for it in batch]), "actions": torch.stack([pad(it["actions"]) for it in batch]),
            "cad_image": torch.stack([it["cad_image"] for it in batch]), "timesteps": torch.stack([torch.arange(mx) for _ in batch]),
            "multiview_images": None}


def test_collate_matches_reference_semantics_fp32():
    items = _items([5, 3, 4], u8=False)
    got, ref = D.collate_with_padding(items, pin=False), _reference_collate(items)
    for k in ("frames", "actions", "cad_image", "timesteps"):
        assert got[k].dtype == ref[k].dtype and torch.equal(got[k], ref[k]), k
    assert got["multiview_images"] is None
    assert float(got["frames"][1, 3:].max()) == -1.0 and float(got["actions"][1, 3:].max()) == -1.0     # ragged clip padded with -1
    assert torch.equal(got["timesteps"][0], torch.arange(5))                                             # not the items' own timesteps
    # equal lengths: nothing padded; single item
    same = D.collate_with_padding(_items([4, 4], u8=False), pin=False)
    assert same["frames"].shape == (2, 4, 1, 8, 8)
    assert D.collate_with_padding(_items([2], u8=False), pin=False)["actions"].shape == (1, 2, 7)
    mv = _items([2, 2], u8=False)
    for it in mv:
        it["multiview_images"] = torch.zeros(3, 1, 8, 8)
    assert D.collate_with_padding(mv, pin=False)["multiview_images"].shape == (2, 3, 1, 8, 8)
    mv[1]["multiview_images"] = None
    assert D.collate_with_padding(mv, pin=False)["multiview_images"] is None                             # reference :360-363


def test_uint8_batch_normalises_to_the_fp32_batch_bit_exactly():
    """uint8 collate (pad = pixel 0) followed by the kernel's (u/255 - 0.5)/0.5 == the fp32 collate (pad = -1.0)."""
    iu, ifl = _items([5, 2], u8=True, seed=3), _items([5, 2], u8=False, seed=3)
    bu, bf = D.collate_with_padding(iu, pin=False), D.collate_with_padding(ifl, pin=False)
    assert bu["frames"].dtype == torch.uint8 and bu["cad_image"].dtype == torch.uint8
    assert torch.equal(D.normalize_u8(bu["frames"]), bf["frames"]) and torch.equal(D.normalize_u8(bu["cad_image"]), bf["cad_image"])
    assert torch.equal(bu["actions"], bf["actions"])
    # ToTensor + Normalize(0.5, 0.5) restated with explicit ops on every pixel value
    u = torch.arange(256, dtype=torch.uint8)
    assert torch.equal(D.normalize_u8(u), (u.float().div(255.0) - 0.5) / 0.5)
    assert float(D.normalize_u8(torch.zeros(1, dtype=torch.uint8))) == -1.0


def test_grayscale_matches_pil_exactly():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (5, 32, 32, 3), dtype=np.uint8)
    a[0, 0, :8] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [1, 1, 1], [254, 255, 255], [128, 127, 129]]
    ref = np.stack([np.array(Image.fromarray(f).convert("L")) for f in a])
    assert np.array_equal(D.pil_grayscale_u8(a), ref)
    t = D.frames_from_rgb(a, as_uint8=True)
    assert t.shape == (5, 1, 32, 32) and t.dtype == torch.uint8
    assert torch.equal(D.frames_from_rgb(a, as_uint8=False), D.normalize_u8(t))
    # the CAD path's cv2 formula: grey pixels map to themselves, the weights sum to 1 << 14
    g = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, 1)
    assert np.array_equal(D.cv2_bgr2gray_u8(g), np.arange(256, dtype=np.uint8)) and 1868 + 9617 + 4899 == 1 << 14


def test_stager_passthrough_on_cpu():
    batches = [D.collate_with_padding(_items([3, 2], u8=True, seed=s), pin=False) for s in range(3)]
    st = D.DeviceStager(batches, "cpu")
    assert len(st) == 3
    out = list(st)
    assert len(out) == 3 and out[0]["frames"].dtype == torch.uint8 and out[0]["actions"].dtype == torch.float32
    assert out[2]["timesteps"].dtype == torch.long and torch.equal(out[1]["frames"], batches[1]["frames"])
    assert "multiview_images" not in out[0]
