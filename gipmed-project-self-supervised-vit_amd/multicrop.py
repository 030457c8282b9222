"""Random multi-crop boxes for the DINO step (host side of gv_crop_resize).

The reference has no DINO loop; the recipe is the paper's: 2 global crops of scale
(0.4, 1) resized to 224 and 8 local crops of scale (0.05, 0.4) resized to 96, each flipped
left-right with p = 0.5.  Boxes follow torchvision's RandomResizedCrop.get_params (10 tries of a
uniform area and a log-uniform aspect ratio in (3/4, 4/3), then a uniform integer origin,
centred fallback).  Only the 6 integers per crop travel to the device; the resampling itself is
one HIP kernel over tiles that are already resident in HBM.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np
import torch


class MultiCropSampler:
    def __init__(self, batch: int, tile: int = 256, n_global: int = 2, n_local: int = 8, global_scale=(0.4, 1.0),
                 local_scale=(0.05, 0.4), ratio=(3.0 / 4.0, 4.0 / 3.0), flip_p: float = 0.5, seed: int = 0):
        self.B, self.tile, self.G, self.Lc = batch, tile, n_global, n_local
        self.gs, self.ls, self.ratio, self.flip_p = tuple(global_scale), tuple(local_scale), tuple(ratio), flip_p
        self.rng = np.random.default_rng(seed)

    def _boxes(self, n: int, scale) -> np.ndarray:
        """n boxes at once (this runs inside the launch loop, once per step: a Python loop over 640 crops cost 6 ms of host
        time): the 10 tries of every crop are drawn together, the first one that fits is taken, the square tile's fallback is
        the whole tile; then a uniform integer origin.  -> int64 [n, 4] = (y0, x0, h, w)."""
        H = W = self.tile
        lr = (math.log(self.ratio[0]), math.log(self.ratio[1]))
        target = H * W * self.rng.uniform(scale[0], scale[1], (n, 10))
        r = np.exp(self.rng.uniform(lr[0], lr[1], (n, 10)))
        w = np.rint(np.sqrt(target * r)).astype(np.int64)
        h = np.rint(np.sqrt(target / r)).astype(np.int64)
        ok = (w > 0) & (w <= W) & (h > 0) & (h <= H)
        first = ok.argmax(1)
        rows = np.arange(n)
        hit = ok[rows, first]
        h = np.where(hit, h[rows, first], H)
        w = np.where(hit, w[rows, first], W)
        y0 = np.floor(self.rng.random(n) * (H - h + 1)).astype(np.int64)
        x0 = np.floor(self.rng.random(n) * (W - w + 1)).astype(np.int64)
        return np.stack([np.where(hit, y0, 0), np.where(hit, x0, 0), h, w], 1)

    def sample(self, device=None):
        """-> (global boxes int32 [G*B, 6], local boxes int32 [L*B, 6]), rows crop-major: (tile, y0, x0, h, w, flip)."""
        out = []
        for n, scale in ((self.G, self.gs), (self.Lc, self.ls)):
            rows = np.empty((n * self.B, 6), np.int32)
            rows[:, 0] = np.tile(np.arange(self.B), n)
            rows[:, 1:5] = self._boxes(n * self.B, scale)
            rows[:, 5] = self.rng.random(n * self.B) < self.flip_p
            t = torch.from_numpy(rows)
            out.append(t.to(device, non_blocking=True) if device is not None else t)
        return tuple(out)


class ViewAugmentSampler:
    """Per-crop draws of DINO's view augmentation (DataAugmentationDINO; the pixel work is gv_crop_augment's):
      every crop:   ColorJitter(brightness 0.4, contrast 0.4, saturation 0.2, hue 0.1) with p = 0.8 (the four operations in a
                    random order with uniform factors, torchvision semantics), grayscale with p = 0.2;
      global crop 1: Gaussian blur p = 1.0;  global crop 2: blur p = 0.1, solarise (threshold 128) p = 0.2;  local crops: blur p = 0.5.
    The blur is the 3x3 Gaussian of the reference's recipes (transformations.py:140-147, torchvision GaussianBlur(3)) with
    sigma ~ U(0.1, 2.0); DINO itself blurs with PIL's radius-based filter -- same role, smaller support.  Rows are crop-major
    like MultiCropSampler's boxes.  Same distributions as torchvision's modules, this build's own random stream."""

    DT = np.dtype([("n_color", "<i4"), ("order", "<i4", (4,)), ("bf", "<f4"), ("cf", "<f4"), ("sf", "<f4"), ("hue", "<i4"), ("gray", "<i4"),
                   ("blur", "<i4"), ("kc", "<f4"), ("ks", "<f4"), ("solar", "<i4")])

    def __init__(self, batch: int, n_global: int = 2, n_local: int = 8, jitter_p: float = 0.8, jitter=(0.4, 0.4, 0.2, 0.1), gray_p: float = 0.2,
                 blur_p=(1.0, 0.1, 0.5), solar_p=(0.0, 0.2, 0.0), blur_sigma=(0.1, 2.0), solar_threshold: int = 128, seed: int = 0):
        self.B, self.G, self.Lc = batch, n_global, n_local
        self.jp, self.j, self.gp, self.bp, self.sp, self.bs, self.st = jitter_p, tuple(jitter), gray_p, tuple(blur_p), tuple(solar_p), tuple(blur_sigma), solar_threshold
        self.rng = np.random.default_rng(seed)

    def _draw(self, n_crops: int, first_crop: int) -> np.ndarray:
        r, n = self.rng, n_crops * self.B
        a = np.zeros(n, self.DT)
        a["bf"] = a["cf"] = a["sf"] = 1.0
        a["solar"] = -1
        on = r.random(n) < self.jp
        a["n_color"] = np.where(on, 4, 0)
        a["order"] = r.permuted(np.tile(np.arange(4, dtype=np.int32), (n, 1)), axis=1)
        b, c, s, h = self.j
        a["bf"], a["cf"], a["sf"] = r.uniform(max(0.0, 1 - b), 1 + b, n), r.uniform(max(0.0, 1 - c), 1 + c, n), r.uniform(max(0.0, 1 - s), 1 + s, n)
        a["hue"] = (r.uniform(-h, h, n) * 255).astype(np.int64) % 256
        a["gray"] = r.random(n) < self.gp
        crop = first_crop + np.arange(n) // self.B                         # global crop 0, global crop 1, local crops
        kind = np.minimum(crop, 2) if first_crop == 0 else np.full(n, 2)
        bp, sp = np.asarray(self.bp)[kind], np.asarray(self.sp)[kind]
        sg = r.uniform(self.bs[0], self.bs[1], n).astype(np.float32)
        side = np.exp(np.float32(-0.5) * np.square(np.float32(1.0) / sg).astype(np.float32)).astype(np.float32)   # torchvision _get_gaussian_kernel1d(3, sigma)
        tot = (side + np.float32(1.0) + side).astype(np.float32)
        a["blur"] = r.random(n) < bp
        a["kc"], a["ks"] = (np.float32(1.0) / tot).astype(np.float32), (side / tot).astype(np.float32)
        a["solar"] = np.where(r.random(n) < sp, self.st, -1)
        return a

    def sample_host(self):
        """-> (global records [G*B], local records [L*B]) as structured numpy arrays (oracle format: to_dicts)."""
        return self._draw(self.G, 0), self._draw(self.Lc, self.G)

    @staticmethod
    def to_dicts(a: np.ndarray):
        return [dict(order=[int(o) for o in r["order"][: int(r["n_color"])]], bf=float(r["bf"]), cf=float(r["cf"]), sf=float(r["sf"]), hue=int(r["hue"]),
                     gray=bool(r["gray"]), blur=(float(r["kc"]), float(r["ks"])) if r["blur"] else None, solar=int(r["solar"])) for r in a]

    @staticmethod
    def pack(a: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy())

    def sample(self, device=None):
        """-> (global, local) packed gv_view_params records as uint8 tensors on ``device``."""
        out = [self.pack(x) for x in self.sample_host()]
        return tuple(t.to(device, non_blocking=True) if device is not None else t for t in out)
