"""Host side of the device augmentation (gv_augment): the reference's recipes as parameter samplers.

The reference builds a torchvision ``Compose`` per ``transform_type`` (transformations.py:131-197) and runs it on every
tile in DataLoader workers.  Here the RANDOM DRAWS of a recipe are made on the host -- a few numbers per tile -- and the
pixel work runs on the GPU over tiles that are already in HBM: ``TileAugmenter.apply(tiles)`` returns the augmented tiles
and, for the two operations whose result is not a uint8 pixel (Cutout after Normalize, transformations.py:206-207;
MyMeanPixelRegularization, 91-100), the per-tile fill boxes that ``gv_patchify`` applies in normalised space.

Recipes (names of transformations.py:131-197): flip, rvf, cbnfrsc / cbnfrs, pcbnfrsc / pcbnfrs, cbnfr, bnfrsc / bnfrs, frs,
aug_receptornet.  Draw semantics follow torchvision: ColorJitter = a random permutation of brightness / contrast /
saturation / hue with uniform factors; GaussianBlur(3) sigma ~ U(1e-7, 0.1); noise sigma ~ U(0, 0.05); vertical /
horizontal flip with p = 0.5; MyRotation one of 0 / 90 / 180 / 270 degrees (counter-clockwise); RandomAffine scale ~
U(1, 1.2); Cutout one 100-px hole centred uniformly (clipped at the border).  The streams are this build's own (numpy
Generator): same distributions as the reference, not the same numbers."""
from __future__ import annotations

import ctypes
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from . import ops

MEAN_PIXEL = (0.87316266, 0.79902739, 0.84941472)      # transformations.py:99
_D4_TABLES: Dict[tuple, np.ndarray] = {}

RECIPES: Dict[str, dict] = {
    "flip": dict(geo=("v?", "h?")),
    "rvf": dict(geo=("r", "v?")),
    "cbnfrsc": dict(color="fixed", blur=True, noise=True, geo=("v?", "r"), zoom=True, cutout="norm"),
    "cbnfrs": dict(color="fixed", blur=True, noise=True, geo=("v?", "r"), zoom=True),
    "pcbnfrsc": dict(color="param", blur=True, noise=True, geo=("v?", "r"), zoom=True, cutout="norm"),
    "pcbnfrs": dict(color="param", blur=True, noise=True, geo=("v?", "r"), zoom=True),
    "cbnfr": dict(color="fixed", blur=True, noise=True, geo=("v?", "r")),
    "bnfrsc": dict(blur=True, noise=True, geo=("v?", "r"), zoom=True, cutout="norm"),
    "bnfrs": dict(blur=True, noise=True, geo=("v?", "r"), zoom=True),
    "frs": dict(geo=("v?", "r"), zoom=True),
    "aug_receptornet": dict(color="receptornet", geo=("h?", "r"), cutout="black", mean_pixel=0.75),
}


def normal_table() -> torch.Tensor:
    """1024 quantile mid-points of N(0, 1): z_i = Phi^-1((i + 0.5) / 1024), float32."""
    p = (torch.arange(1024, dtype=torch.float64) + 0.5) / 1024.0
    return torch.special.ndtri(p).to(torch.float32)


def blur_weights(sigma: float) -> Tuple[float, float]:
    """torchvision _get_gaussian_kernel1d(3, sigma) in float32: (centre, side)."""
    x = np.linspace(-1.0, 1.0, 3).astype(np.float32)
    pdf = np.exp(-0.5 * np.square(x / np.float32(sigma)).astype(np.float32)).astype(np.float32)
    k = (pdf / pdf.sum(dtype=np.float32)).astype(np.float32)
    return float(k[1]), float(k[0])


def compose_d4(seq: Sequence[str]) -> int:
    """Any sequence of 'v' (flip rows), 'h' (flip columns), 'r<k>' (rotate k x 90 degrees counter-clockwise) on a square image
    is one of the 8 dihedral elements: out[y, x] = img[u, v] with (u, v) = (x, y) if bit 0 else (y, x), then u -> n-1-u if
    bit 1, v -> n-1-v if bit 2.  Found by pushing an index grid through the sequence."""
    n = 4
    idx = np.arange(n * n).reshape(n, n)
    g = idx
    for s in seq:
        g = g[::-1] if s == "v" else g[:, ::-1] if s == "h" else np.rot90(g, int(s[1]))
    for code in range(8):
        ys, xs = np.mgrid[0:n, 0:n]
        u, v = (xs, ys) if code & 1 else (ys, xs)
        u = n - 1 - u if code & 2 else u
        v = n - 1 - v if code & 4 else v
        if np.array_equal(idx[u, v], g):
            return code
    raise AssertionError("not a dihedral element")


def zoom_fixed(s: float, size: int) -> Tuple[int, int]:
    """16.16 fixed-point coefficients of PIL's NEAREST affine for a zoom by s about the image centre."""
    a = 1.0 / s
    c = a * (-size * 0.5) + size * 0.5
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return fix(a), fix(c + a * 0.5)


class TileAugmenter:
    def __init__(self, recipe: str, tile_size: int = 256, color_param: float = 0.1, norm_mean=(0.8998, 0.8253, 0.9357),
                 norm_std=(0.1125, 0.1751, 0.0787), seed: Optional[int] = None, device=None):
        if recipe not in RECIPES:
            raise ValueError(f"unknown augmentation recipe '{recipe}' (known: {sorted(RECIPES)})")
        self.recipe, self.cfg, self.size, self.cp = recipe, RECIPES[recipe], tile_size, color_param
        self.mean, self.std = tuple(norm_mean), tuple(norm_std)
        self.rng = np.random.default_rng(seed)
        self.dev = device
        self._z = self._stats = None

    # ---- draws -------------------------------------------------------------------------------
    def _color_ranges(self):
        kind = self.cfg.get("color")
        if kind == "fixed":          # transformations.py:143-144, 175-176
            return (0.85, 1.15), (0.75, 1.25), (0.9, 1.1), (-0.1, 0.1)
        if kind == "param":          # 153-154: ColorJitter(brightness=cp, contrast=2cp, saturation=cp, hue=cp)
            cp = self.cp
            return (max(0.0, 1 - cp), 1 + cp), (max(0.0, 1 - 2 * cp), 1 + 2 * cp), (max(0.0, 1 - cp), 1 + cp), (-cp, cp)
        if kind == "receptornet":    # 166
            return (max(0.0, 1 - 64.0 / 255), 1 + 64.0 / 255), (0.25, 1.75), (0.75, 1.25), (-0.04, 0.04)
        return None

    def sample_batch(self, n: int) -> Dict[str, np.ndarray]:
        """The draws of n tiles as COLUMNS (vectorised: this runs inside the launch loop, once per step)."""
        r, cfg, size = self.rng, self.cfg, self.size
        c: Dict[str, np.ndarray] = {"n": np.int64(n)}
        cr = self._color_ranges()
        if cr is not None:
            c["order"] = r.permuted(np.tile(np.arange(4, dtype=np.int32), (n, 1)), axis=1)
            c["bf"], c["cf"], c["sf"] = (r.uniform(cr[k][0], cr[k][1], n) for k in range(3))
            c["hue"] = (r.uniform(cr[3][0], cr[3][1], n) * 255).astype(np.int64) % 256          # int() truncates toward zero, like np.uint8()
        if cfg.get("blur"):
            c["blur_sigma"] = r.uniform(1e-7, 1e-1, n)
        if cfg.get("noise"):
            c["sigma"], c["seed"] = r.uniform(0.0, 0.05, n), r.integers(0, 2 ** 32, n, dtype=np.uint64)
        geo = cfg.get("geo", ())
        c["geo"] = np.stack([r.integers(0, 4, n) if g == "r" else (r.random(n) < 0.5).astype(np.int64) for g in geo], 1) if geo else np.zeros((n, 0), np.int64)
        if cfg.get("zoom"):
            c["zoom_s"] = r.uniform(1.0, 1.2, n)
        if cfg.get("cutout"):
            c["cut_y"], c["cut_x"] = r.integers(0, size, n), r.integers(0, size, n)
        if cfg.get("mean_pixel"):
            c["mean_pixel"] = r.random(n) < cfg["mean_pixel"]
        return c

    def _geo_seq(self, row) -> List[str]:
        out = []
        for g, v in zip(self.cfg.get("geo", ()), row):
            if g == "r":
                out.append(f"r{int(v)}")
            elif v:
                out.append(g[0])
        return out

    def to_dicts(self, c: Dict[str, np.ndarray]) -> List[dict]:
        """The same draws, one dict per tile in the oracle's format (oracle/augment_oracle.augment_tile) + 'fill'."""
        n, size, cfg = int(c["n"]), self.size, self.cfg
        ps = []
        for i in range(n):
            p: dict = {}
            if "order" in c:
                p["order"] = [int(o) for o in c["order"][i]]
                p["bf"], p["cf"], p["sf"], p["hue"] = float(c["bf"][i]), float(c["cf"][i]), float(c["sf"][i]), int(c["hue"][i])
            if "blur_sigma" in c:
                kc, ks = blur_weights(float(c["blur_sigma"][i]))
                p["blur"] = (kc, ks) if ks * 255.0 * 8.0 > 1e-3 else None       # below that the 3x3 sum rounds to the centre pixel exactly
            if "sigma" in c:
                p["sigma"], p["seed"] = float(c["sigma"][i]), int(c["seed"][i])
            p["geo"] = self._geo_seq(c["geo"][i])
            if "zoom_s" in c:
                p["zoom"] = zoom_fixed(float(c["zoom_s"][i]), size)
            fill = None
            if "cut_y" in c:
                y, x = int(c["cut_y"][i]), int(c["cut_x"][i])
                box = (max(0, y - 50), min(size, y + 50), max(0, x - 50), min(size, x + 50))      # Cutout(n_holes=1, length=100), 34-41
                if cfg["cutout"] == "black":
                    p["cut"] = box
                else:
                    fill = (*box, 0.0, 0.0, 0.0)
            if "mean_pixel" in c and bool(c["mean_pixel"][i]):
                fill = (0, size, 0, size) + tuple((MEAN_PIXEL[k] - self.mean[k]) / self.std[k] for k in range(3))
            p["fill"] = fill
            ps.append(p)
        return ps

    def sample_one(self) -> dict:
        return self.to_dicts(self.sample_batch(1))[0]

    _DT = np.dtype([("n_color", "<i4"), ("order", "<i4", (4,)), ("bf", "<f4"), ("cf", "<f4"), ("sf", "<f4"), ("hue", "<i4"), ("blur", "<i4"),
                    ("kc", "<f4"), ("ks", "<f4"), ("sigma", "<f4"), ("seed", "<u4"), ("d4", "<i4"), ("zoom", "<i4"), ("a0", "<i4"), ("a2", "<i4"),
                    ("cut", "<i4", (4,))])

    def pack_batch(self, c: Dict[str, np.ndarray]):
        """columns -> (packed gv_augment_params records as uint8 [n * 88], fill f32 [n, 8] or None), without a Python loop."""
        n, size, cfg = int(c["n"]), self.size, self.cfg
        a = np.zeros(n, self._DT)
        a["bf"] = a["cf"] = a["sf"] = 1.0
        if "order" in c:
            a["n_color"], a["order"] = 4, c["order"]
            a["bf"], a["cf"], a["sf"], a["hue"] = c["bf"], c["cf"], c["sf"], c["hue"]
        if "blur_sigma" in c:                       # torchvision _get_gaussian_kernel1d(3, sigma) in float32, all tiles at once
            sg = c["blur_sigma"].astype(np.float32)
            side = np.exp(np.float32(-0.5) * np.square(np.float32(1.0) / sg).astype(np.float32)).astype(np.float32)
            tot = (side + np.float32(1.0) + side).astype(np.float32)     # pdf.sum() over (-1, 0, 1) in that order
            ks, kc = (side / tot).astype(np.float32), (np.float32(1.0) / tot).astype(np.float32)
            on = ks * np.float32(255.0 * 8.0) > 1e-3
            a["blur"], a["kc"], a["ks"] = on, np.where(on, kc, 0), np.where(on, ks, 0)
        if "sigma" in c:
            a["sigma"], a["seed"] = c["sigma"], c["seed"].astype(np.uint32)
        if c["geo"].shape[1]:
            key = tuple(cfg["geo"])
            if key not in _D4_TABLES:               # every draw combination of this recipe's flip / rotation slots -> dihedral code
                dims = [4 if g == "r" else 2 for g in key]
                tab = np.zeros(dims, np.int32)
                for idx in np.ndindex(*dims):
                    tab[idx] = compose_d4(self._geo_seq(idx))
                _D4_TABLES[key] = tab
            a["d4"] = _D4_TABLES[key][tuple(c["geo"].T)]
        if "zoom_s" in c:
            inv = 1.0 / c["zoom_s"]
            cc = inv * (-size * 0.5) + size * 0.5
            a["zoom"], a["a0"], a["a2"] = 1, np.floor(inv * 65536.0 + 0.5).astype(np.int64), np.floor((cc + inv * 0.5) * 65536.0 + 0.5).astype(np.int64)
        fill = None
        if "cut_y" in c:
            box = np.stack([np.maximum(0, c["cut_y"] - 50), np.minimum(size, c["cut_y"] + 50), np.maximum(0, c["cut_x"] - 50), np.minimum(size, c["cut_x"] + 50)], 1)
            if cfg["cutout"] == "black":
                a["cut"] = box
            else:
                fill = np.zeros((n, 8), np.float32)
                fill[:, :4], fill[:, 7] = box, 1.0
        if "mean_pixel" in c and c["mean_pixel"].any():
            fill = np.zeros((n, 8), np.float32) if fill is None else fill
            m = c["mean_pixel"]
            fill[m, :4] = (0, size, 0, size)
            fill[m, 4:7] = [(MEAN_PIXEL[k] - self.mean[k]) / self.std[k] for k in range(3)]
            fill[m, 7] = 1.0
        return a.view(np.uint8).reshape(-1), fill

    @staticmethod
    def pack(ps: List[dict]) -> np.ndarray:
        arr = (L.gv_augment_params * len(ps))()
        for a, p in zip(arr, ps):
            order = p.get("order", [])
            a.n_color = len(order)
            for k, o in enumerate(order):
                a.order[k] = o
            a.bf, a.cf, a.sf, a.hue = p.get("bf", 1.0), p.get("cf", 1.0), p.get("sf", 1.0), p.get("hue", 0)
            if p.get("blur"):
                a.blur, a.kc, a.ks = 1, p["blur"][0], p["blur"][1]
            a.sigma, a.seed = p.get("sigma", 0.0), p.get("seed", 0)
            a.d4 = compose_d4(p.get("geo", ()))
            if p.get("zoom"):
                a.zoom, a.a0, a.a2 = 1, p["zoom"][0], p["zoom"][1]
            for k, v in enumerate(p.get("cut") or (0, 0, 0, 0)):
                a.cut[k] = v
        return np.frombuffer(bytes(arr), dtype=np.uint8).copy()

    # ---- device ------------------------------------------------------------------------------
    def run(self, tiles_u8: torch.Tensor, packed_dev: torch.Tensor) -> torch.Tensor:
        """Launch gv_augment with records that are already on the device (data.DevicePrefetcher carries them with the tiles)."""
        n, dev = tiles_u8.shape[0], tiles_u8.device
        if self._z is None or self._z.device != dev:
            self._z = normal_table().to(dev)
        if self._stats is None or self._stats.numel() < n or self._stats.device != dev:
            self._stats = torch.zeros(n, dtype=torch.int64, device=dev)
        return ops.augment(tiles_u8, packed_dev, self._stats, self._z)

    def apply(self, tiles_u8: torch.Tensor, params: Optional[List[dict]] = None):
        """tiles_u8 [n, H, W, 3] u8 on the GPU -> (augmented tiles, fill f32 [n, 8] or None for gv_patchify).  ``params``:
        explicit per-tile draws in the oracle's dict format (tests); default = fresh draws of the recipe."""
        n = tiles_u8.shape[0]
        dev = tiles_u8.device
        if self._z is None or self._z.device != dev:
            self._z = normal_table().to(dev)
        if self._stats is None or self._stats.numel() < n or self._stats.device != dev:
            self._stats = torch.zeros(n, dtype=torch.int64, device=dev)
        if params is None:
            packed, fill_np = self.pack_batch(self.sample_batch(n))
        else:
            packed = self.pack(params)
            fill_np = None
            if any(p.get("fill") for p in params):
                fill_np = np.zeros((n, 8), np.float32)
                for i, p in enumerate(params):
                    if p.get("fill"):
                        fill_np[i, :7], fill_np[i, 7] = p["fill"], 1.0
        out = ops.augment(tiles_u8, torch.from_numpy(packed).to(dev, non_blocking=True), self._stats, self._z)
        fill = None if fill_np is None else torch.from_numpy(fill_np).to(dev, non_blocking=True)
        return out, fill
