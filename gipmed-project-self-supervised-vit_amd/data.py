"""Tile sources for the training driver.  The reference reads whole-slide images through
openslide (datasets.py:28-631, out of scope: SURVEY section 2 #15); what the hot path needs is
its OUTPUT contract -- batches ``{'Data': tiles, 'Target': [B,1] int64}`` -- which these
sources provide as uint8 NHWC tiles (normalisation is fused on the GPU):

  * ``SyntheticTiles``   seeded H&E-like tiles, for benchmarks / smoke runs;
  * ``TileFolder``       pre-extracted tiles in the reference's raw format
                         (datasets.py:452-466): ``tile_<i>.data`` = one ASCII header line
                         ``dtype w h c`` + raw bytes; the label comes from a ``labels.csv``
                         (``slide,label``) beside the slide folders.
"""
from __future__ import annotations

import csv
import os
from typing import Callable, Dict, Iterator, List, Optional

import numpy as np
import torch


def read_tile_file(path: str) -> np.ndarray:
    with open(path, "rb") as fh:
        header = fh.readline()
        raw = fh.read()
    dtype, w, h, c = header.decode("ascii").strip().split()
    return np.frombuffer(raw, dtype=dtype).reshape((int(w), int(h), int(c)))


def write_tile_file(path: str, tile: np.ndarray) -> None:
    w, h, c = tile.shape
    with open(path, "wb") as fh:
        fh.write(f"{tile.dtype.name} {w} {h} {c}\n".encode("ascii"))
        fh.write(np.ascontiguousarray(tile).tobytes())


class SyntheticTiles:
    """len(self) batches per epoch of [B, size, size, 3] uint8 tiles + [B,1] int64 targets."""

    def __init__(self, batch: int, size: int = 256, batches_per_epoch: int = 100, num_classes: int = 2, seed: int = 1234,
                 mean=(0.8998, 0.8253, 0.9357), std=(0.1125, 0.1751, 0.0787), fixed: bool = False):
        self.B, self.size, self.n, self.C, self.seed, self.fixed = batch, size, batches_per_epoch, num_classes, seed, fixed
        self.mean, self.std = torch.tensor(mean) * 255.0, torch.tensor(std) * 255.0
        self.epoch = 0

    def __len__(self):
        return self.n

    def _batch(self, g: torch.Generator) -> Dict[str, torch.Tensor]:
        x = torch.randn(self.B, self.size, self.size, 3, generator=g) * self.std + self.mean
        return {"Data": x.round().clamp(0, 255).to(torch.uint8), "Target": torch.randint(0, self.C, (self.B, 1), generator=g)}

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        g = torch.Generator().manual_seed(self.seed + (0 if self.fixed else 1000 * self.epoch))
        first = self._batch(g)
        for i in range(self.n):
            yield first if self.fixed else (first if i == 0 else self._batch(g))
        self.epoch += 1


class TileFolder:
    """root/<slide>/tile_<i>.data (+ root/labels.csv).  Tiles are sharded over ranks by index."""

    def __init__(self, root: str, batch: int, transform: Optional[Callable] = None, rank: int = 0, world: int = 1, seed: int = 0,
                 tile_size: int = 256):
        self.root, self.B, self.transform, self.tile_size = root, batch, transform, tile_size
        labels: Dict[str, int] = {}
        lp = os.path.join(root, "labels.csv")
        if os.path.exists(lp):
            with open(lp) as f:
                for row in csv.reader(f):
                    if len(row) >= 2 and row[1].strip().lstrip("-").isdigit():
                        labels[row[0].strip()] = int(row[1])
        items: List = []
        for slide in sorted(os.listdir(root)):
            d = os.path.join(root, slide)
            if os.path.isdir(d):
                for fn in sorted(os.listdir(d)):
                    if fn.startswith("tile_") and fn.endswith(".data"):
                        items.append((os.path.join(d, fn), labels.get(slide, 0)))
        if not items:
            raise FileNotFoundError(f"no tile_<i>.data files under {root}")
        self.items = items[rank::world]
        self.rng = np.random.default_rng(seed + rank)

    def __len__(self):
        return len(self.items) // self.B

    def __iter__(self):
        order = self.rng.permutation(len(self.items))
        for b in range(len(self)):
            tiles, tgt = [], []
            for j in order[b * self.B:(b + 1) * self.B]:
                path, y = self.items[j]
                t = read_tile_file(path)[: self.tile_size, : self.tile_size, :3]
                tiles.append(self.transform(t) if self.transform is not None else t)
                tgt.append(y)
            yield {"Data": torch.from_numpy(np.stack(tiles)), "Target": torch.tensor(tgt, dtype=torch.int64).view(-1, 1)}
