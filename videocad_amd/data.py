"""Frame batching into pinned host memory and on to HBM (north_star "image_loader.py frame batching into
pinned HBM").  Everything here is host-side plumbing around ONE contract, the collated batch of the reference loader:

    frames [B,S,1,224,224], actions [B,S,7] f32, cad_image [B,1,224,224], timesteps [B,S] i64, multiview_images None

  * `pad_array` / `collate_with_padding` — twins of reference data_loader/data_loader.py:313-366 (pad every clip to the batch
    maximum with -1, `timesteps = arange(max_len)`), writing straight into PINNED buffers (the reference gets pinning from
    `DataLoader(pin_memory=True)`, :186-196, through one more copy).
  * uint8 mode — the stored frames are uint8 (pkl `frames uint8 [N,224,224,3]`, reference :434-447); the reference converts
    every frame to fp32 on the host (PIL Grayscale -> ToTensor -> Normalize(0.5, 0.5)) and ships 4 bytes per pixel over PCIe
    (trainer.py:308).  Here the host only does the integer part — `pil_grayscale_u8`, PIL's exact ITU-R 601-2 integer luma —
    and the batch stays uint8: 1 byte per pixel over PCIe and in HBM; `(u/255 - 0.5)/0.5` happens inside the patchify kernel
    (csrc/norm.h, `vcad_forward_u8`) with the same fp32 operations, so the patch vectors are bit-identical.  Padding value
    -1.0 is pixel 0 in that encoding ((0/255 - 0.5)/0.5 == -1.0 exactly).
  * `DeviceStager` — double-buffered H2D staging on its own stream: batch i+1 is copied while step i computes.
"""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------ pixel conversions (host, integer)
def pil_grayscale_u8(rgb: np.ndarray) -> np.ndarray:
    """PIL `Image.convert('L')` / torchvision `Grayscale()` on uint8 [..., 3] (reference data_loader.py:444-445, main.py:105):
    L = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16  (ITU-R 601-2 luma in 16.16 fixed point; verified bit-exact against PIL)."""
    a = rgb.astype(np.uint32)
    return ((a[..., 0] * 19595 + a[..., 1] * 38470 + a[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def cv2_bgr2gray_u8(bgr: np.ndarray) -> np.ndarray:
    """OpenCV `cvtColor(COLOR_BGR2GRAY)` on uint8 [..., 3] (the CAD image path, reference data_loader.py:471): OpenCV's published
    14-bit fixed-point form Y = (B*1868 + G*9617 + R*4899 + (1 << 13)) >> 14.  (cv2 is not in this image: restated, not probed.)"""
    a = bgr.astype(np.uint32)
    return ((a[..., 0] * 1868 + a[..., 1] * 9617 + a[..., 2] * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def normalize_u8(gray_u8: torch.Tensor) -> torch.Tensor:
    """ToTensor + Normalize(0.5, 0.5) (reference main.py:103-108) — what the patchify kernel applies to uint8 pixels."""
    return (gray_u8.to(torch.float32) / 255.0 - 0.5) / 0.5


def frames_from_rgb(frames_rgb_u8: np.ndarray, as_uint8: bool = True) -> torch.Tensor:
    """pkl frames uint8 [N,H,W,3] -> [N,1,H,W]: uint8 gray (as_uint8) or the reference's normalised fp32."""
    g = torch.from_numpy(pil_grayscale_u8(frames_rgb_u8)).unsqueeze(1)
    return g if as_uint8 else normalize_u8(g)


# ------------------------------------------------------------------------------------------------ on-disk clips
END_ACTION = 950            # reference generate_dataset.py:184: a key value of 950 ends the clip


def finalize_clip(frames: np.ndarray, actions: np.ndarray, timesteps) -> dict:
    """The record reference generate_dataset.py:180-199 pickles for one clip, from the per-action arrays `convert_logs_to_vectors` /
    `extract_frames_from_actions` produced: the first frame is duplicated and an all-zero action row prepended (so step 0 predicts the first
    real action from the initial screen), then everything is cut after the first row whose key column (index 3) holds END_ACTION.
    -> {'frames' uint8 [N,H,W,3], 'actions' float [N,7] (0..999, -1 = unused slot), 'timesteps' [N]}"""
    frames = np.vstack([frames[:1], frames])
    actions = np.vstack([np.zeros((1, 7)), actions])
    timesteps = np.array(list(timesteps[:1]) + list(timesteps))
    end = np.where(actions[:, 3] == END_ACTION)[0]
    if len(end) > 0:
        frames, actions, timesteps = frames[: end[0] + 1], actions[: end[0] + 1], timesteps[: end[0] + 1]
    assert len(frames) == len(actions), "Number of frames and actions must be the same"
    return {"frames": frames, "actions": actions, "timesteps": timesteps}


class PklClipDataset:
    """Reader of the reference's dataset directory (data_loader/data_loader.py:293-312 file discovery, :399-508 item path;
    sequence_retriver.py:25-33; image_loader.py:31-44): `<root>/.../<id>_data.pkl` = {'frames' uint8 [N,224,224,3], 'actions' [N,7],
    'timesteps'} next to the clip's CAD render `<image_dir>/<id[:4]>/<id><suffix>` (PNG).  Items feed `collate_with_padding`.

    mode  'rgb8'  frames stay the STORED uint8 RGB [N,H,W,3] — zero per-pixel host work; PIL's luma + ToTensor + Normalize run inside the
                  patchify kernel (vcad_forward_rgb8); 3 bytes per pixel over PCIe
          'gray8' frames uint8 gray [N,1,H,W] (host: PIL's integer luma only); 1 byte per pixel (vcad_forward_u8)
          'f32'   the reference's item, fp32 [N,1,H,W] normalised to [-1, 1] (4 bytes per pixel)
    The CAD image is cv2-read BGR -> BGR2GRAY -> /255 -> Normalize(0.5, 0.5) in the reference (:471-476); here: gray uint8 [1,H,W] for the uint8
    modes (normalised in-kernel), the same fp32 for 'f32'.  Stored frames and renders must already be image_size (the released dataset is
    `data_resized`): the reference's Resize / cv2.resize are identities there and are not restated."""

    def __init__(self, dataset_path: str, image_dir: Optional[str] = None, mode: str = "rgb8", image_suffix: str = "_frame.png", image_size=(224, 224),
                 view_ids=None, multiview_dir: Optional[str] = None):
        import os
        assert mode in ("rgb8", "gray8", "f32")
        self.mode, self.image_size, self.image_suffix = mode, tuple(image_size), image_suffix
        self.image_dir = image_dir if image_dir is not None else dataset_path
        self.view_ids = list(view_ids) if view_ids else []                      # reference :418-431: `<multiview_dir>/<id[:4]>/<id>_<view>.png`
        self.multiview_dir = multiview_dir
        if self.view_ids and not multiview_dir:
            raise ValueError("PklClipDataset: view_ids needs multiview_dir (the reference leaves base_dir unbound without it, data_loader.py:421-423)")
        files, png_ids = [], set()
        for root, _dirs, fs in os.walk(dataset_path):
            files += [os.path.join(root, f) for f in fs if f.endswith("_data.pkl")]
            png_ids.update(f.split("_")[0] for f in fs if f.endswith(".png"))
        self.data_files = sorted(files)                                         # reference :307-310: sorted, one render per clip id
        if png_ids:
            # the reference pairs the idx-th sorted pkl PATH with the idx-th sorted PNG id found under the same tree (:297-311, :411) and asserts
            # the two lists have one length; kept, so a directory layout whose two orders differ resolves exactly as it does there
            self.ids = sorted(png_ids)
            assert len(self.data_files) == len(self.ids), "Number of data files and image files must be the same"
        else:                                                                   # renders live elsewhere (image_dir): the clip id is the pkl's own prefix
            self.ids = [os.path.basename(f).split("_")[0] for f in self.data_files]

    def __len__(self):
        return len(self.data_files)

    def _cad_gray_u8(self, clip_id: str) -> np.ndarray:
        import os
        from PIL import Image
        path = os.path.join(self.image_dir, clip_id[:4], clip_id + self.image_suffix)
        rgb = np.asarray(Image.open(path).convert("RGB"))                       # cv2.imread delivers the same pixels as BGR
        if rgb.shape[:2] != self.image_size[::-1]:
            raise ValueError(f"{path}: {rgb.shape[:2]} render, expected {self.image_size[::-1]} (resize offline)")
        return cv2_bgr2gray_u8(rgb[..., ::-1])

    def __getitem__(self, idx: int) -> dict:
        import pickle
        if idx < 0 or idx >= len(self.data_files):
            raise IndexError("Index out of range")
        with open(self.data_files[idx], "rb") as f:
            data = pickle.load(f)
        frames = np.ascontiguousarray(data["frames"]); actions = np.asarray(data["actions"])
        if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[3] != 3 or frames.shape[1:3] != self.image_size[::-1]:
            raise ValueError(f"{self.data_files[idx]}: frames {frames.dtype} {frames.shape}, expected uint8 [N,{self.image_size[1]},{self.image_size[0]},3]")
        cad = torch.from_numpy(self._cad_gray_u8(self.ids[idx])).unsqueeze(0)
        if self.mode == "rgb8":
            ft = torch.from_numpy(frames)
        elif self.mode == "gray8":
            ft = frames_from_rgb(frames, as_uint8=True)
        else:
            ft = frames_from_rgb(frames, as_uint8=False); cad = normalize_u8(cad)
        return {"frames": ft, "actions": torch.from_numpy(actions.astype(np.float32)), "cad_image": cad,
                "multiview_images": self._multiview(self.ids[idx]), "timesteps": torch.arange(ft.shape[0])}

    def _multiview(self, clip_id: str):
        """reference data_loader.py:416-431, 479-490: the views of a clip, each cv2-read -> BGR2GRAY -> /255 -> Normalize: [V,1,H,W], uint8 gray in
        the uint8 modes (normalised inside the patchify kernel like the CAD image), normalised fp32 in 'f32'; None without view_ids."""
        if not self.view_ids:
            return None
        import os
        from PIL import Image
        views = []
        for v in self.view_ids:
            path = os.path.join(self.multiview_dir, clip_id[:4], f"{clip_id}_{v}.png")
            if not os.path.exists(path):
                raise ValueError(f"Missing view {v} for file {clip_id}")
            rgb = np.asarray(Image.open(path).convert("RGB"))
            if rgb.shape[:2] != self.image_size[::-1]:
                raise ValueError(f"{path}: {rgb.shape[:2]} render, expected {self.image_size[::-1]} (resize offline)")
            views.append(torch.from_numpy(cv2_bgr2gray_u8(rgb[..., ::-1])).unsqueeze(0))
        mv = torch.stack(views)
        return mv if self.mode != "f32" else normalize_u8(mv)


# ------------------------------------------------------------------------------------------------ collate
def _pad_value(dtype: torch.dtype):
    return 0 if dtype == torch.uint8 else -1           # pixel 0 == -1.0 after normalisation


def pad_array(max_len: int, array: torch.Tensor) -> torch.Tensor:
    """reference data_loader.py:313-318"""
    pad = max_len - array.shape[0]
    if pad > 0:
        array = torch.cat([array, torch.full((pad, *array.shape[1:]), _pad_value(array.dtype), dtype=array.dtype)], dim=0)
    return array


def collate_with_padding(batch: List[dict], pin: bool = True) -> dict:
    """reference data_loader.py:321-366, writing into pinned buffers (one copy instead of cat + stack + pin).
    Items: {'frames' [S_i,1,H,W] f32|u8, 'actions' [S_i,7], 'cad_image' [1,H,W] f32|u8, optional 'multiview_images'}."""
    B = len(batch)
    max_len = max(int(it["frames"].shape[0]) for it in batch)
    f0, a0, c0 = batch[0]["frames"], batch[0]["actions"], batch[0]["cad_image"]
    pin = bool(pin and torch.cuda.is_available())

    def buf(shape, dtype, fill=None):
        t = torch.empty(shape, dtype=dtype, pin_memory=pin)
        if fill is not None:
            t.fill_(fill)
        return t

    frames = buf((B, max_len, *f0.shape[1:]), f0.dtype, _pad_value(f0.dtype))
    actions = buf((B, max_len, *a0.shape[1:]), torch.float32, -1)
    cad = buf((B, *c0.shape), c0.dtype)
    for b, it in enumerate(batch):
        n = int(it["frames"].shape[0])
        frames[b, :n].copy_(it["frames"]); actions[b, :n].copy_(it["actions"]); cad[b].copy_(it["cad_image"])
    out = {"frames": frames, "actions": actions, "cad_image": cad,
           "timesteps": torch.arange(max_len).repeat(B, 1)}                     # :337 — fresh arange, whatever the items carried
    mv = [it.get("multiview_images", None) for it in batch]
    out["multiview_images"] = torch.stack(mv) if all(m is not None for m in mv) else None
    return out


# ------------------------------------------------------------------------------------------------ H2D staging
class DeviceStager:
    """Wraps a loader of collated (ideally pinned) batches: yields device-resident batch dicts, the NEXT batch's H2D copy running
    on a private stream while the caller computes on the current one (reference trainer.py:307-311 copies synchronously on the
    compute stream).  Two device buffer sets alternate; a batch's buffers are handed back to the copy stream only after the
    compute stream has passed the point where the following batch was requested."""

    KEYS = ("frames", "actions", "cad_image", "timesteps", "multiview_images")

    def __init__(self, loader: Iterable, device, frames_dtype: Optional[torch.dtype] = None):
        self.loader, self.device = loader, torch.device(device)
        self.frames_dtype = frames_dtype
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch: dict) -> dict:
        out = {}
        for k in self.KEYS:
            v = batch.get(k, None)
            if v is None:
                continue
            if k == "timesteps":
                out[k] = v.to(self.device, dtype=torch.long, non_blocking=True)
            elif v.dtype == torch.uint8:
                out[k] = v.to(self.device, non_blocking=True)                     # stays uint8: normalised in the patchify kernel
            else:
                out[k] = v.to(self.device, dtype=torch.float, non_blocking=True)
        return out

    def __iter__(self) -> Iterator[dict]:
        it = iter(self.loader)
        if self.copy_stream is None:
            for b in it:
                yield self._stage(b)
            return
        cur = torch.cuda.current_stream(self.device)

        def fetch():
            try:
                b = next(it)
            except StopIteration:
                return None
            self.copy_stream.wait_stream(cur)          # buffers freed by the allocator are reused in stream order of `cur`
            with torch.cuda.stream(self.copy_stream):
                d = self._stage(b)
            ev = torch.cuda.Event(); ev.record(self.copy_stream)
            return d, ev, b                            # keep the pinned source alive until the copy has been waited for

        nxt = fetch()
        while nxt is not None:
            d, ev, _src = nxt
            cur.wait_event(ev)
            for v in d.values():
                v.record_stream(cur)
            nxt = fetch()                              # enqueue the next copy before the caller starts computing on `d`
            yield d
