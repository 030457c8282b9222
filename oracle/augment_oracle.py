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


# --------------------------------------------------------------------------- #
# tile augmentation recipes of the reference (transformations.py:10-100, 131-197), restated for uint8 NHWC tiles.
# The reference runs them on the CPU through PIL / torchvision / skimage; here every operation takes its random draws
# as explicit parameters, so that the device kernel (gv_augment) can be checked BYTE FOR BYTE against this file.
# Pinned against PIL itself where PIL defines the operation (tests/test_oracle.py): brightness / contrast / saturation
# are PIL's ImageEnhance blend (float32, truncation) exactly; HSV -> RGB is exact over all 2^24 inputs; RGB -> HSV agrees
# with PIL on S and V exactly and on the H byte for 99.6 % of all colours (+-1 otherwise: PIL mixes float and double);
# the zoom is PIL's NEAREST affine in 16.16 fixed point.  Gaussian noise uses this build's own counter-based generator
# (the reference draws from numpy's global stream, transformations.py:71-88): same distribution, not the same stream.
# --------------------------------------------------------------------------- #
F32 = np.float32


def _L(img):            # PIL "L" conversion
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16


def _blend(deg, img, f):
    """PIL Image.blend(degenerate, image, factor) as ImageEnhance uses it: float32, clip, truncate."""
    f = F32(f)
    d = deg.astype(F32)
    t = d + (f * (img.astype(F32) - d)).astype(F32)
    return np.clip(t, 0, 255).astype(np.uint8)


def rgb_to_hsv(img):
    r, g, b = (img[..., i].astype(np.int32) for i in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    gray = maxc == minc
    cr = (maxc - minc).astype(F32)
    crs = np.where(gray, F32(1), cr)
    s = cr / np.where(maxc == 0, 1, maxc).astype(F32)
    rc, gc, bc = ((maxc - c).astype(F32) / crs for c in (r, g, b))
    h = np.where(r == maxc, bc - gc, np.where(g == maxc, (F32(2.0) + rc) - bc, (F32(4.0) + gc) - rc)).astype(F32)
    t = (h / F32(6.0) + F32(1.0)).astype(F32)
    h = (t - np.trunc(t)).astype(F32)
    uh = np.clip((h * F32(255.0)).astype(np.int32), 0, 255)
    us = np.clip((s * F32(255.0)).astype(np.int32), 0, 255)
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def hsv_to_rgb(hsv):
    h, s, v = (hsv[..., i].astype(np.int64) for i in range(3))
    hh = (h.astype(F32) * F32(6.0) / F32(255.0)).astype(F32)
    i = np.floor(hh).astype(np.int64)
    f = (hh - i.astype(F32)).astype(F32)
    fs = (s.astype(F32) / F32(255.0)).astype(F32)
    vv = v.astype(F32)
    rnd = lambda x: np.floor(x.astype(F32) + F32(0.5)).astype(np.int64)
    one = F32(1.0)
    p = rnd(vv * (one - fs)); q = rnd(vv * (one - (fs * f).astype(F32))); t = rnd(vv * (one - (fs * (one - f)).astype(F32)))
    i6 = i % 6
    r = np.choose(i6, [v, q, p, p, t, v]); g = np.choose(i6, [t, v, v, q, p, p]); b = np.choose(i6, [p, p, t, v, v, q])
    out = np.where((s == 0)[..., None], np.stack([v, v, v], -1), np.stack([r, g, b], -1))
    return np.clip(out, 0, 255).astype(np.uint8)


def color_op(img, op, p):
    """op 0 brightness, 1 contrast, 2 saturation, 3 hue (torchvision ColorJitter on PIL images)."""
    if op == 0:
        return _blend(np.zeros_like(img), img, p["bf"])
    if op == 1:
        L = _L(img)
        mean = int(float(L.sum()) / L.size + 0.5)
        return _blend(np.full_like(img, mean), img, p["cf"])
    if op == 2:
        return _blend(np.repeat(_L(img)[..., None], 3, -1).astype(np.uint8), img, p["sf"])
    hsv = rgb_to_hsv(img)
    hsv[..., 0] = ((hsv[..., 0].astype(np.int64) + int(p["hue"])) & 255).astype(np.uint8)
    return hsv_to_rgb(hsv)


def blur3(img, kc, ks):
    """torchvision gaussian_blur(kernel 3) on a uint8 tensor: float32 3x3 correlation with reflect padding, taps added in
    raster order, round half to even (kc / ks: centre / side weight of the normalised 1-D kernel, float32)."""
    k1 = np.array([ks, kc, ks], F32)
    pad = np.pad(img.astype(F32), ((1, 1), (1, 1), (0, 0)), mode="reflect")
    H, W = img.shape[:2]
    acc = np.zeros(img.shape, F32)
    for dy in range(3):
        for dx in range(3):
            acc = (acc + (F32(k1[dy] * k1[dx]) * pad[dy:dy + H, dx:dx + W]).astype(F32)).astype(F32)
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


def noise_index(seed, H, W):
    """counter-based index into the 1024-entry normal table: murmur3 finaliser of (seed, pixel, channel)."""
    u = np.arange(H * W * 3, dtype=np.uint64).reshape(H, W, 3) + 1
    u = (np.uint64(seed) + np.uint64(0x9E3779B9) * u) & np.uint64(0xFFFFFFFF)
    u ^= u >> np.uint64(16); u = (u * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    u ^= u >> np.uint64(13); u = (u * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    u ^= u >> np.uint64(16)
    return (u >> np.uint64(22)).astype(np.int64)


def add_noise(img, sigma, seed, ztable):
    """MyGaussianNoiseTransform (transformations.py:71-88): x/255 + N(0, sigma), clip to [0, 1], (255 x).astype(uint8)."""
    H, W = img.shape[:2]
    z = ztable[noise_index(seed, H, W)].astype(F32)
    x = (img.astype(F32) / F32(255.0)).astype(F32)
    x = np.clip((x + (F32(sigma) * z).astype(F32)).astype(F32), F32(0), F32(1))
    return (F32(255.0) * x).astype(F32).astype(np.uint8)


def zoom_fixed(s, size):
    """16.16 fixed-point coefficients of PIL's NEAREST affine for torchvision RandomAffine(degrees=0, scale=s) about the
    image centre: source index = (a2 + a0 * dst) >> 16."""
    a = 1.0 / s
    c = a * (-size * 0.5) + size * 0.5
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return fix(a), fix(c + a * 0.5)


def zoom(img, a0, a2):
    H, W = img.shape[:2]
    ys, xs = np.mgrid[0:H, 0:W].astype(np.int64)
    xi, yi = (a2 + a0 * xs) >> 16, (a2 + a0 * ys) >> 16
    ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
    out = np.zeros_like(img)
    out[ok] = img[yi[ok], xi[ok]]
    return out


def augment_tile(img, p, ztable):
    """One tile through a recipe.  p: dict with 'order' (colour op ids in application order), 'bf' 'cf' 'sf' 'hue',
    'blur' (kc, ks) or None, 'sigma' 'seed', 'geo' (list of 'v' / 'h' / 'r0'..'r3' in application order), 'zoom' (a0, a2) or
    None, 'cut' (y0, y1, x0, x1) or None (black box, Cutout on the [0, 1] tensor, transformations.py:10-45)."""
    for op in p.get("order", ()):
        img = color_op(img, op, p)
    if p.get("blur"):
        img = blur3(img, *p["blur"])
    if p.get("sigma", 0.0) > 0:
        img = add_noise(img, p["sigma"], p["seed"], ztable)
    for g in p.get("geo", ()):
        img = img[::-1] if g == "v" else img[:, ::-1] if g == "h" else np.rot90(img, int(g[1]))      # PIL rotates counter-clockwise
    if p.get("zoom"):
        img = zoom(img, *p["zoom"])
    img = np.ascontiguousarray(img).copy()
    if p.get("cut"):
        y0, y1, x0, x1 = p["cut"]
        img[y0:y1, x0:x1] = 0
    return img


# --------------------------------------------------------------------------- #
# DINO view augmentation (gv_crop_augment): per crop, after the random-resized crop + flip --
# ColorJitter ops in the drawn order -> grayscale (PIL "L", replicated) -> 3x3 Gaussian blur -> solarise (PIL ImageOps.solarize).
# Pinned against PIL in tests/test_oracle.py (convert("L"), ImageOps.solarize; the colour ops and the blur as above).
# --------------------------------------------------------------------------- #
def solarize(img, threshold=128):
    return np.where(img >= threshold, 255 - img.astype(np.int64), img).astype(np.uint8)


def to_gray(img):
    return np.repeat(_L(img)[..., None], 3, -1).astype(np.uint8)


def view_augment(crop, p):
    """crop u8 [S, S, 3]; p: dict(order, bf, cf, sf, hue, gray, blur=(kc, ks) | None, solar=threshold | -1)."""
    img = crop
    for op in p.get("order", ()):
        img = color_op(img, op, p)
    if p.get("gray"):
        img = to_gray(img)
    if p.get("blur"):
        img = blur3(img, *p["blur"])
    if p.get("solar", -1) >= 0:
        img = solarize(img, p["solar"])
    return img


def crop_views(tiles, boxes, out, params):
    crops = crop_resize(tiles, boxes, out)
    return np.stack([view_augment(c, p) for c, p in zip(crops, params)])
