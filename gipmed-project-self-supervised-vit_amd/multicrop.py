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

    def _box(self, scale) -> Tuple[int, int, int, int]:
        H = W = self.tile
        area = H * W
        lr = (math.log(self.ratio[0]), math.log(self.ratio[1]))
        for _ in range(10):
            target = area * self.rng.uniform(scale[0], scale[1])
            r = math.exp(self.rng.uniform(lr[0], lr[1]))
            w, h = int(round(math.sqrt(target * r))), int(round(math.sqrt(target / r)))
            if 0 < w <= W and 0 < h <= H:
                return int(self.rng.integers(0, H - h + 1)), int(self.rng.integers(0, W - w + 1)), h, w
        return 0, 0, H, W            # square tile: the centred fallback is the whole tile

    def sample(self, device=None):
        """-> (global boxes int32 [G*B, 6], local boxes int32 [L*B, 6]), rows crop-major: (tile, y0, x0, h, w, flip)."""
        out = []
        for n, scale in ((self.G, self.gs), (self.Lc, self.ls)):
            rows = np.empty((n * self.B, 6), np.int32)
            for c in range(n):
                for i in range(self.B):
                    y0, x0, h, w = self._box(scale)
                    rows[c * self.B + i] = (i, y0, x0, h, w, int(self.rng.random() < self.flip_p))
            t = torch.from_numpy(rows)
            out.append(t.to(device, non_blocking=True) if device is not None else t)
        return tuple(out)
