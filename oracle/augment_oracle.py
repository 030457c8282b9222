"""CPU restatement of the multi-crop input stage (TEST INFRASTRUCTURE ONLY -- never imported by the product).

The reference has no DINO loop and augments on the CPU with PIL (transformations.py:103-209); SURVEY 8f
rank 1 asks for the crop pipeline on the device.  Spec restated here (parity unpinned at the reference,
pinned against torch in tests/test_oracle.py):

* box sampling = torchvision RandomResizedCrop.get_params: up to 10 tries of area = A * U(scale),
  log-uniform aspect ratio in (3/4, 4/3), w = round(sqrt(area * r)), h = round(sqrt(area / r)); accept if
  it fits, then a uniform integer origin; otherwise the centred fallback box (DINO paper: global crops
  scale (0.4, 1) -> 224, local crops scale (0.05, 0.4) -> 96; horizontal flip with p = 0.5);
* resampling = torchvision tensor-mode resized_crop without antialias: F.interpolate on float32,
  mode='bilinear', align_corners=False, then round half to even, clamp, cast to uint8.
"""
from __future__ import annotations

import math

import numpy as np


def sample_box(rng: np.random.Generator, H: int, W: int, scale, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    area = H * W
    log_r = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target = area * rng.uniform(scale[0], scale[1])
        r = math.exp(rng.uniform(log_r[0], log_r[1]))
        w = int(round(math.sqrt(target * r)))
        h = int(round(math.sqrt(target / r)))
        if 0 < w <= W and 0 < h <= H:
            y0 = int(rng.integers(0, H - h + 1))
            x0 = int(rng.integers(0, W - w + 1))
            return y0, x0, h, w
    in_ratio = W / H
    if in_ratio < ratio[0]:
        w = W; h = int(round(w / ratio[0]))
    elif in_ratio > ratio[1]:
        h = H; w = int(round(h * ratio[1]))
    else:
        w, h = W, H
    return (H - h) // 2, (W - w) // 2, h, w


def crop_resize(tiles: np.ndarray, boxes: np.ndarray, out: int) -> np.ndarray:
    """tiles u8 [n_tiles, H, W, 3]; boxes int [n, 6] = (tile, y0, x0, h, w, flip) -> u8 [n, out, out, 3]."""
    f32 = np.float32
    res = np.empty((len(boxes), out, out, 3), np.uint8)
    o = np.arange(out, dtype=f32)
    for n, (t, y0, x0, h, w, flip) in enumerate(np.asarray(boxes).tolist()):
        img = tiles[t, y0:y0 + h, x0:x0 + w].astype(f32)

        def axis(size):
            s = f32(size) / f32(out)
            src = np.maximum(s * (o + f32(0.5)) - f32(0.5), f32(0.0)).astype(f32)
            i0 = src.astype(np.int64)
            i1 = i0 + (i0 < size - 1)
            l1 = (src - i0.astype(f32)).astype(f32)
            return i0, i1, (f32(1.0) - l1).astype(f32), l1

        iy0, iy1, ly0, ly1 = axis(h)
        ix0, ix1, lx0, lx1 = axis(w)
        top = lx0[None, :, None] * img[iy0][:, ix0] + lx1[None, :, None] * img[iy0][:, ix1]
        bot = lx0[None, :, None] * img[iy1][:, ix0] + lx1[None, :, None] * img[iy1][:, ix1]
        v = ly0[:, None, None] * top + ly1[:, None, None] * bot
        v = np.clip(np.rint(v.astype(f32)), 0, 255).astype(np.uint8)
        res[n] = v[:, ::-1] if flip else v
    return res
