"""Augmentation / normalisation hook surface of the hot path's input contract.

Mirror of ``define_transformations(transform_type, train, tile_size, color_param=0.1,
norm_type='Ron')`` (reference transformations.py:103-209).  In the reference the returned
callable maps a PIL tile to a normalised float tensor [3,H,W] on the CPU; here the tile stays
uint8 NHWC -- ToTensor's /255 and Normalize are fused into the patchify kernel on the GPU
(gv_patchify mean/std) -- so the callable returned by this mirror maps a tile
(PIL.Image or uint8 HxWx3 array) to a uint8 HxWx3 numpy array and carries ``.mean`` / ``.std``.

As in the reference (transformations.py:199-200) a ``transform_type`` that is not a known
string is used AS the transform: that is the user's augmentation hook.  The named recipes
(flip, rvf, cbnfrsc, cbnfrs, pcbnfrsc, pcbnfrs, cbnfr, bnfrsc, bnfrs, frs, aug_receptornet) run ON THE DEVICE:
the returned object carries ``device_recipe`` and the driver hands the tiles, once they are in HBM, to
``gipvit.augment.TileAugmenter`` (gv_augment: colour jitter, blur, noise, flips / rotations, zoom, cutout --
byte-exact against oracle/augment_oracle.py, which is pinned against PIL).  Called directly on a host tile the
object applies only the flips / 90-degree rotations ('flip', 'rvf'); the other recipes have no CPU path.
The DINO random-resized crops + flips are cut on the device too (gipvit.multicrop.MultiCropSampler + gv_crop_resize).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Union

import numpy as np

# transformations.py:104-116 (data): per-channel statistics by cohort
MEAN = {
    "TCGA": [58.2069073 / 255, 96.22645279 / 255, 70.26442606 / 255],
    "HEROHE": [224.46091564 / 255, 190.67338568 / 255, 218.47883547 / 255],
    "Ron": [0.8998, 0.8253, 0.9357],
    "Imagenet": [0.485, 0.456, 0.406],
    "Amir": [0.9357, 0.8253, 0.8998],
}
STD = {
    "TCGA": [40.40400300279664 / 255, 58.90625962739444 / 255, 45.09334057330417 / 255],
    "HEROHE": [np.sqrt(1110.25292532) / 255, np.sqrt(2950.9804851) / 255, np.sqrt(1027.10911208) / 255],
    "Ron": [0.1125, 0.1751, 0.0787],
    "Imagenet": [0.229, 0.224, 0.225],
    "Amir": [0.0787, 0.1751, 0.1125],
}
GEOMETRIC = {"none", "flip", "rvf"}          # recipes made only of flips / 90-degree rotations: also callable on the host
DEVICE_RECIPES = {"flip", "rvf", "cbnfrsc", "cbnfrs", "pcbnfrsc", "pcbnfrs", "cbnfr", "bnfrsc", "bnfrs", "frs", "aug_receptornet"}


def _to_u8(img) -> np.ndarray:
    a = np.asarray(img)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise TypeError(f"tile must be uint8 HxWx3, got {a.dtype} {a.shape}")
    return a


class TileTransform:
    """Callable tile -> uint8 HxWx3 array; ``mean`` / ``std`` feed the fused GPU normalise."""

    def __init__(self, ops: Sequence[Callable[[np.ndarray, np.random.Generator], np.ndarray]], mean, std, tile_size: int, seed=None,
                 device_recipe: Optional[str] = None, color_param: float = 0.1):
        self.ops, self.mean, self.std, self.tile_size = list(ops), tuple(mean), tuple(std), tile_size
        self.rng = np.random.default_rng(seed)
        self.device_recipe, self.color_param = device_recipe, color_param      # for gipvit.augment.TileAugmenter

    def __call__(self, img) -> np.ndarray:
        if self.device_recipe is not None and self.device_recipe not in GEOMETRIC:
            raise NotImplementedError(f"recipe '{self.device_recipe}' runs on the device (gipvit.augment.TileAugmenter / gv_augment); it has no CPU path")
        a = _to_u8(img)
        for op in self.ops:
            a = op(a, self.rng)
        return np.ascontiguousarray(a)


def _hflip(a, rng):
    return a[:, ::-1] if rng.random() < 0.5 else a


def _vflip(a, rng):
    return a[::-1] if rng.random() < 0.5 else a


def _rot90(a, rng):          # MyRotation: one of 0 / 90 / 180 / 270 degrees (transformations.py:48-56)
    return np.rot90(a, int(rng.integers(0, 4)))


def define_transformations(transform_type: Union[str, Callable], train: bool, tile_size: int, color_param: float = 0.1,
                           norm_type: str = "Ron", seed: Optional[int] = None):
    if norm_type not in MEAN:
        raise KeyError(f"unknown norm_type '{norm_type}', known: {sorted(MEAN)}")
    mean, std = MEAN[norm_type], STD[norm_type]
    if not isinstance(transform_type, str):
        # user-supplied augmentation hook: used as-is; give it the stats if it has none
        if not hasattr(transform_type, "mean"):
            try:
                transform_type.mean, transform_type.std = tuple(mean), tuple(std)
            except AttributeError:
                pass
        return transform_type
    if transform_type not in DEVICE_RECIPES | {"none"}:
        raise ValueError(f"unknown transform_type '{transform_type}' (named recipes: {sorted(DEVICE_RECIPES)}; pass a callable for a custom one)")
    ops = []
    if train and transform_type == "flip":
        ops = [_hflip, _vflip]
    elif train and transform_type == "rvf":
        ops = [_rot90, _vflip]
    return TileTransform(ops, mean, std, tile_size, seed, device_recipe=transform_type if (train and transform_type != "none") else None,
                         color_param=color_param)
