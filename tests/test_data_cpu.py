"""Host side of the input path: the collate twin against a literal restatement of the reference's
pad / stack semantics (reference data_loader/data_loader.py:313-366), PIL-exact grayscale, the uint8 <-> fp32 equivalence the
patchify kernel relies on, and the staging iterator on the CPU."""
import os

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


# ------------------------------------------------------------------------------------------------ on-disk clips (f2 / f4)
def _write_dataset(root, n_clips=3, S=224):
    """a synthetic dataset directory in the reference's layout: <root>/<id[:4]>/<id>_data.pkl + <id>_frame.png"""
    import os, pickle
    from PIL import Image
    rng = np.random.default_rng(7)
    ids, raw = [], {}
    for c in range(n_clips):
        cid = f"{1000 + c:04d}{c:04d}"
        n = 3 + c
        frames = rng.integers(0, 256, (n, S, S, 3), dtype=np.uint8)
        actions = rng.integers(0, 999, (n, 7)).astype(np.float64); actions[:, 3] = 5
        rec = D.finalize_clip(frames, actions, list(range(n)))
        os.makedirs(os.path.join(root, cid[:4]), exist_ok=True)
        with open(os.path.join(root, cid[:4], cid + "_data.pkl"), "wb") as f:
            pickle.dump(rec, f)
        render = rng.integers(0, 256, (S, S, 3), dtype=np.uint8)
        Image.fromarray(render).save(os.path.join(root, cid[:4], cid + "_frame.png"))
        ids.append(cid); raw[cid] = (rec, render)
    return ids, raw


def _reference_getitem(rec, render_rgb):
    """reference data_loader.py:434-508, literally, with PIL doing what torchvision's transforms do (Resize is the identity at 224):
    Image.fromarray(frame) -> Grayscale(1) -> ToTensor -> Normalize([0.5], [0.5]); CAD: BGR2GRAY -> /255 -> Normalize"""
    from PIL import Image
    fr = []
    for frame in rec["frames"]:
        g = np.array(Image.fromarray(frame).convert("L"))                        # torchvision Grayscale(1) = PIL convert('L')
        fr.append(((torch.from_numpy(g).float().div(255.0) - 0.5) / 0.5).unsqueeze(0))
    bgr = render_rgb[..., ::-1].astype(np.uint32)                                # what cv2.imread returns
    gray = ((bgr[..., 0] * 1868 + bgr[..., 1] * 9617 + bgr[..., 2] * 4899 + (1 << 13)) >> 14).astype(np.uint8)
    cad = (torch.from_numpy(gray.astype(np.float32) / 255.0).unsqueeze(0) - 0.5) / 0.5
    return {"frames": torch.stack(fr), "actions": torch.from_numpy(rec["actions"].astype(np.float32)), "cad_image": cad,
            "timesteps": torch.arange(len(fr))}


def test_finalize_clip_is_the_reference_record():
    """reference generate_dataset.py:180-199: duplicated first frame + zero action row, cut after the END_ACTION row"""
    fr = np.arange(4 * 2 * 2 * 3, dtype=np.uint8).reshape(4, 2, 2, 3)
    ac = np.tile(np.arange(7, dtype=np.float64), (4, 1)); ac[:, 3] = [1, 2, D.END_ACTION, 4]
    rec = D.finalize_clip(fr, ac, [10, 11, 12, 13])
    assert rec["frames"].shape[0] == rec["actions"].shape[0] == 4                 # 1 (prepended) + 3 (cut after the end action)
    assert np.array_equal(rec["frames"][0], fr[0]) and np.array_equal(rec["frames"][1], fr[0]) and not rec["actions"][0].any()
    assert rec["actions"][-1, 3] == D.END_ACTION and list(rec["timesteps"]) == [10, 10, 11, 12]
    full = D.finalize_clip(fr, np.zeros((4, 7)), [0, 1, 2, 3])
    assert full["frames"].shape[0] == 5


def test_pkl_dataset_items_match_the_reference_getitem(tmp_path):
    pytest.importorskip("PIL.Image")
    ids, raw = _write_dataset(str(tmp_path))
    ds = {m: D.PklClipDataset(str(tmp_path), mode=m) for m in ("f32", "gray8", "rgb8")}
    assert len(ds["f32"]) == 3 and ds["f32"].ids == sorted(ids)
    for i, cid in enumerate(sorted(ids)):
        ref = _reference_getitem(*raw[cid])
        it = ds["f32"][i]
        for k in ("frames", "actions", "cad_image", "timesteps"):
            assert it[k].dtype == ref[k].dtype and torch.equal(it[k], ref[k]), (cid, k)
        g8 = ds["gray8"][i]
        assert g8["frames"].dtype == torch.uint8 and torch.equal(D.normalize_u8(g8["frames"]), ref["frames"]) and torch.equal(D.normalize_u8(g8["cad_image"]), ref["cad_image"])
        r8 = ds["rgb8"][i]
        assert r8["frames"].dtype == torch.uint8 and r8["frames"].shape[1:] == (224, 224, 3) and np.array_equal(r8["frames"].numpy(), raw[cid][0]["frames"])
    # through the collate: ragged clips padded with pixel 0 (= -1.0 after the in-kernel normalisation) in both uint8 layouts
    b = D.collate_with_padding([ds["rgb8"][0], ds["rgb8"][2]], pin=False)
    assert b["frames"].shape == (2, 6, 224, 224, 3) and int(b["frames"][0, 4:].max()) == 0 and b["cad_image"].shape == (2, 1, 224, 224)
    with pytest.raises(IndexError):
        ds["f32"][3]


def test_pkl_dataset_multiview_items(tmp_path):
    """reference data_loader.py:416-431, 479-490: `view_ids` -> [V,1,H,W] views read from <multiview_dir>/<id[:4]>/<id>_<view>.png, each
    BGR2GRAY -> /255 -> Normalize like the CAD image; stacked by the collate into [B,V,1,H,W] (what `vcad_set_multiview` takes)"""
    Image = pytest.importorskip("PIL.Image")
    ids, raw = _write_dataset(str(tmp_path / "data"))
    rng = np.random.default_rng(5)
    views = {}
    for cid in ids:
        os.makedirs(tmp_path / "mv" / cid[:4], exist_ok=True)
        for v in ("a", "b"):
            img = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8); views[(cid, v)] = img
            Image.fromarray(img).save(str(tmp_path / "mv" / cid[:4] / f"{cid}_{v}.png"))
    with pytest.raises(ValueError):
        D.PklClipDataset(str(tmp_path / "data"), mode="f32", view_ids=["a"])                    # no multiview_dir
    ds = D.PklClipDataset(str(tmp_path / "data"), mode="f32", view_ids=["a", "b"], multiview_dir=str(tmp_path / "mv"))
    d8 = D.PklClipDataset(str(tmp_path / "data"), mode="gray8", view_ids=["a", "b"], multiview_dir=str(tmp_path / "mv"))
    for i, cid in enumerate(sorted(ids)):
        mv = ds[i]["multiview_images"]
        assert mv.shape == (2, 1, 224, 224) and mv.dtype == torch.float32
        for j, v in enumerate(("a", "b")):
            bgr = views[(cid, v)][..., ::-1].astype(np.uint32)
            gray = ((bgr[..., 0] * 1868 + bgr[..., 1] * 9617 + bgr[..., 2] * 4899 + (1 << 13)) >> 14).astype(np.uint8)
            ref = (torch.from_numpy(gray.astype(np.float32) / 255.0).unsqueeze(0) - 0.5) / 0.5
            assert torch.equal(mv[j], ref)
        assert d8[i]["multiview_images"].dtype == torch.uint8 and torch.equal(D.normalize_u8(d8[i]["multiview_images"]), mv)
    b = D.collate_with_padding([ds[0], ds[1]], pin=False)
    assert b["multiview_images"].shape == (2, 2, 1, 224, 224)
    os.remove(str(tmp_path / "mv" / sorted(ids)[0][:4] / f"{sorted(ids)[0]}_b.png"))
    with pytest.raises(ValueError):
        ds[0]
