"""Tile sources for the training driver.  The reference reads whole-slide images through
openslide (datasets.py:28-631, out of scope: SURVEY section 2 #15); what the hot path needs is
its OUTPUT contract -- batches ``{'Data': tiles, 'Target': [B,1] int64}`` (train.py:1027-1028) and,
for slide-level inference, ``{'Data', 'Label', 'Is Last Batch', 'Slide Filename', 'Slide DataSet',
'Patch Loc'}`` (train.py:1186-1192) -- which these sources provide as uint8 NHWC tiles
(ToTensor / Normalize are fused on the GPU):

  * ``SyntheticTiles`` / ``SyntheticSlides``   seeded H&E-like tiles, for benchmarks / smoke runs;
  * ``TileFolder``       pre-extracted tiles in the reference's raw format (the ``presaved_tiles`` branch,
                         datasets.py:452-466): ``<root>/<slide>/tile_<i>.data`` = one ASCII header line
                         ``dtype w h c`` + raw bytes; labels / folds come from ``<root>/labels.csv``.
                         One epoch = n_slides x n_tiles draws, draw i = a random tile of slide
                         i % n_slides (datasets.py:428-429, 445-450 ``factor`` / ``real_length``);
  * ``InferTiles``       the per-slide chunked iterator of ``Infer_Dataset`` (datasets.py:634-817);
  * ``DevicePrefetcher`` hand-over to HBM: two pinned staging buffers filled by reader threads
                         (file reads release the GIL) and ``copy_(non_blocking=True)`` on a copy stream,
                         issued ONE batch ahead from the main loop (replaces ``pin_memory=True`` workers,
                         train.py:732).
"""
from __future__ import annotations

import csv
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch


def read_tile_file(path: str) -> np.ndarray:
    with open(path, "rb") as fh:
        header = fh.readline()
        raw = fh.read()
    dtype, w, h, c = header.decode("ascii").strip().split()
    return np.frombuffer(raw, dtype=dtype).reshape((int(w), int(h), int(c)))


def read_tile_into(path: str, dst: np.ndarray) -> None:
    """Read one tile file straight into ``dst`` (uint8 [H, W, 3], e.g. a row of a pinned staging buffer).
    The fast path (tile already H x W x 3 uint8) is a single ``readinto``: no intermediate copy."""
    with open(path, "rb") as fh:
        dtype, w, h, c = fh.readline().decode("ascii").strip().split()
        w, h, c = int(w), int(h), int(c)
        if dtype == "uint8" and (w, h, c) == dst.shape and dst.flags["C_CONTIGUOUS"]:
            n = fh.readinto(memoryview(dst).cast("B"))
            if n != dst.nbytes:
                raise IOError(f"{path}: short read ({n} of {dst.nbytes} bytes)")
            return
        t = np.frombuffer(fh.read(), dtype=dtype).reshape((w, h, c))
    H, W = dst.shape[:2]
    if t.shape[0] < H or t.shape[1] < W or t.shape[2] < 3:
        raise ValueError(f"{path}: tile {t.shape} smaller than the requested {dst.shape}")
    dst[...] = t[:H, :W, :3].astype(np.uint8, copy=False)


def write_tile_file(path: str, tile: np.ndarray) -> None:
    w, h, c = tile.shape
    with open(path, "wb") as fh:
        fh.write(f"{tile.dtype.name} {w} {h} {c}\n".encode("ascii"))
        fh.write(np.ascontiguousarray(tile).tobytes())


def _synth(shape, g: torch.Generator, mean, std) -> torch.Tensor:
    x = torch.randn(*shape, 3, generator=g) * std + mean
    return x.round().clamp(0, 255).to(torch.uint8)


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
        return {"Data": _synth((self.B, self.size, self.size), g, self.mean, self.std),
                "Target": torch.randint(0, self.C, (self.B, 1), generator=g)}

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        g = torch.Generator().manual_seed(self.seed + (0 if self.fixed else 1000 * self.epoch))
        first = self._batch(g)
        for i in range(self.n):
            yield first if self.fixed else (first if i == 0 else self._batch(g))
        self.epoch += 1

    # ---- device path (DevicePrefetcher): the same distribution drawn ON the GPU, straight into the device slot.  Drawing
    # 12.6 M normals per batch on the host took 0.18 s -- train.py --dataset synthetic ran at 320 tiles/s next to bench.py's 4 700.
    def start_epoch_device(self, device):
        self._gdev = torch.Generator(device=device).manual_seed(self.seed + (0 if self.fixed else 1000 * self.epoch))
        self._first_dev = None
        self.epoch += 1

    def fill_device(self, data: torch.Tensor, target: torch.Tensor):
        """Write the next batch into ``data`` u8 [B, size, size, 3] / ``target`` i64 [B, 1] (device tensors) on the current stream."""
        if self.fixed and self._first_dev is not None:
            data.copy_(self._first_dev[0]); target.copy_(self._first_dev[1])
            return
        x = torch.randn(data.shape, device=data.device, generator=self._gdev)
        x.mul_(self.std.to(data.device)).add_(self.mean.to(data.device)).round_().clamp_(0, 255)
        data.copy_(x)
        target.copy_(torch.randint(0, self.C, target.shape, device=data.device, generator=self._gdev))
        if self.fixed:
            self._first_dev = (data.clone(), target.clone())


# --------------------------------------------------------------------------- #
# slide folders
# --------------------------------------------------------------------------- #
def _read_labels(root: str, target: Optional[str]) -> Dict[str, Tuple[int, Optional[str]]]:
    """labels.csv: header ``slide,label[,fold][,<target name>...]``.  ``target`` (the reference's --target /
    ``target_kind``) selects a label column by name when the file has one, else the ``label`` column is used.
    Labels may be 0/1 or the reference's 'Positive' / 'Negative' strings (utils.py:770-785 get_label)."""
    out: Dict[str, Tuple[int, Optional[str]]] = {}
    lp = os.path.join(root, "labels.csv")
    if not os.path.exists(lp):
        return out
    with open(lp) as f:
        rows = list(csv.reader(f))
    if not rows:
        return out
    hdr = [h.strip() for h in rows[0]]
    has_hdr = not (len(hdr) >= 2 and hdr[1].lstrip("-").isdigit())
    col, fcol = 1, None
    if has_hdr:
        if target and target in hdr:
            col = hdr.index(target)
        elif "label" in hdr:
            col = hdr.index("label")
        fcol = hdr.index("fold") if "fold" in hdr else None
        rows = rows[1:]
    for r in rows:
        if len(r) <= col:
            continue
        v = r[col].strip()
        if v.lstrip("-").isdigit():
            y = int(v)
        elif v in ("Positive", "Negative"):
            y = int(v == "Positive")
        else:
            continue
        out[r[0].strip()] = (y, r[fcol].strip() if fcol is not None and len(r) > fcol else None)
    return out


def scan_slides(root: str, target: Optional[str] = None):
    """-> [(slide name, [tile paths sorted by index], label, fold)] for every ``<root>/<slide>/`` with tile files."""
    labels = _read_labels(root, target)
    slides = []
    for slide in sorted(os.listdir(root)):
        d = os.path.join(root, slide)
        if not os.path.isdir(d):
            continue
        files = [fn for fn in os.listdir(d) if fn.startswith("tile_") and fn.endswith(".data")]
        files.sort(key=lambda fn: int(fn[5:-5]) if fn[5:-5].isdigit() else 1 << 30)
        if files:
            y, fold = labels.get(slide, (0, None))
            slides.append((slide, [os.path.join(d, fn) for fn in files], y, fold))
    if not slides:
        raise FileNotFoundError(f"no <slide>/tile_<i>.data files under {root}")
    return slides


def select_fold(slides, test_fold, train: bool):
    """Reference fold rule (datasets.py:274-287): TRAINING uses every fold except ``test_fold`` and never the 'test' / 'val'
    folds (``test_fold == -1``: every fold but those two); EVALUATION uses ``[test_fold, 'val']`` (nothing for -1).  Fold 0 is
    spelled 'test' in the slide tables (datasets.py:290-291).  Slides of a root without a fold column (fold None) belong to
    both sets.  An empty selection raises: handing back all slides instead would let the evaluation AUC and the
    model_best choice be computed on training slides."""
    tf = None if test_fold in (-1, "-1") else ("test" if test_fold in (0, "0", "test") else str(test_fold))
    if train:
        keep = [s for s in slides if s[3] is None or s[3] not in (tf, "test", "val")]
    else:
        keep = [s for s in slides if s[3] is None or (tf is not None and s[3] in (tf, "val"))]
    if not keep:
        raise ValueError(f"test_fold={test_fold}: no {'training' if train else 'evaluation'} slides (folds present: "
                         f"{sorted({str(s[3]) for s in slides})})")
    return keep


class TileFolder:
    """Training source over ``root/<slide>/tile_<i>.data`` (+ ``root/labels.csv``).

    One epoch = ``len(slides) * n_tiles`` draws in shuffled order (the DataLoader's ``shuffle=True``), draw ``i`` =
    one uniformly random tile of slide ``i % len(slides)`` (datasets.py:445-466).  Draws are sharded over ranks
    after truncation to a common length, so every rank runs the same number of batches (the reference gives every
    rank the whole set, train.py:732 -- DESIGN.md section 6).  ``fill(out, targets)`` reads the next batch into
    caller buffers with ``workers`` reader threads; iteration yields the batch dict like the other sources."""

    def __init__(self, root: str, batch: int, transform: Optional[Callable] = None, rank: int = 0, world: int = 1, seed: int = 0,
                 tile_size: int = 256, n_tiles: int = 10, test_fold=None, train: bool = True, target: Optional[str] = None,
                 workers: int = 4, slides=None):
        self.root, self.B, self.transform, self.tile_size = root, batch, transform, tile_size
        slides = scan_slides(root, target) if slides is None else slides
        if test_fold is not None:
            slides = select_fold(slides, test_fold, train)
        self.slides = slides
        self.n_tiles, self.rank, self.world = max(1, int(n_tiles)), rank, world
        total = len(slides) * self.n_tiles
        self.per_rank = total // (world * batch) * batch              # common length: same #batches on every rank
        if self.per_rank == 0:
            raise ValueError(f"{root}: {len(slides)} slides x {self.n_tiles} tiles < one global batch of {world * batch}")
        self.rng = np.random.default_rng(seed)                          # the epoch order is SHARED by all ranks
        self.pick = np.random.default_rng(seed + 7919 * (rank + 1))     # which tile of a slide: per rank
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(workers)))
        self._order = None
        self._pos = 0

    def __len__(self):
        return self.per_rank // self.B

    def _start_epoch(self):
        order = self.rng.permutation(len(self.slides) * self.n_tiles)[: self.per_rank * self.world]
        self._order = order[self.rank::self.world] % len(self.slides)
        self._pos = 0

    def plan(self):
        """-> [(path, label)] of the next batch (advances the epoch cursor)."""
        if self._order is None or self._pos + self.B > len(self._order):
            self._start_epoch()
        idx = self._order[self._pos:self._pos + self.B]
        self._pos += self.B
        out = []
        for s in idx:
            _, files, y, _ = self.slides[int(s)]
            out.append((files[int(self.pick.integers(0, len(files)))], y))
        return out

    def fill(self, out: np.ndarray, targets: np.ndarray):
        """Read the next batch into ``out`` [B, H, W, 3] u8 / ``targets`` [B, 1] i64; returns futures to wait on."""
        items = self.plan()
        for j, (_, y) in enumerate(items):
            targets[j, 0] = y
        # one task per reader thread (a strided share of the batch), not one per tile: the launch loop that calls this has
        # ~16 ms per step for ~1000 kernel launches, and 64 submissions + 64 waits cost it more than 1 ms
        nw = min(self.pool._max_workers, len(items))
        return [self.pool.submit(self._read_share, items, out, w, nw) for w in range(nw)]

    def _read_share(self, items, out, w, nw):
        for j in range(w, len(items), nw):
            self._read_one(items[j][0], out[j])

    def _read_one(self, path, dst):
        if self.transform is None or not getattr(self.transform, "ops", True):
            read_tile_into(path, dst)
        else:
            t = read_tile_file(path)[: self.tile_size, : self.tile_size, :3]
            dst[...] = self.transform(t)

    def __iter__(self):
        self._start_epoch()
        for _ in range(len(self)):
            out = np.empty((self.B, self.tile_size, self.tile_size, 3), np.uint8)
            tgt = np.empty((self.B, 1), np.int64)
            for f in self.fill(out, tgt):
                f.result()
            yield {"Data": torch.from_numpy(out), "Target": torch.from_numpy(tgt)}


# --------------------------------------------------------------------------- #
# slide-level inference sources (Infer_Dataset, datasets.py:634-817)
# --------------------------------------------------------------------------- #
def _chunks(seq: Sequence, n: int):
    return [seq[i:i + n] for i in range(0, len(seq), n)]       # utils.py:32-34


class InferTiles:
    """Stateful per-slide chunked iterator: every slide contributes ``min(num_tiles, available)`` tiles, sampled
    without replacement (datasets.py:693), handed out in chunks of ``tiles_per_iter`` (datasets.py:699-700); the chunk
    that finishes a slide carries ``'Is Last Batch': True``.  Attributes ``num_tiles`` (per slide) and
    ``image_file_names`` mirror the ones ``validate`` reads (train.py:1203, 711)."""

    def __init__(self, root: str, tile_size: int = 256, tiles_per_iter: int = 500, num_tiles: int = 500, target: Optional[str] = None,
                 test_fold=None, seed: int = 0, dataset_name: str = "", workers: int = 4, slides=None):
        slides = scan_slides(root, target) if slides is None else slides
        if test_fold is not None:
            slides = select_fold(slides, test_fold, train=False)
        self.slides, self.tile_size, self.tiles_per_iter, self.dataset_name = slides, tile_size, max(1, tiles_per_iter), dataset_name or os.path.basename(root.rstrip("/"))
        rng = np.random.default_rng(seed)
        self.image_file_names = [s[0] for s in slides]
        self.num_tiles, self.slide_grids = [], []
        for _, files, _, _ in slides:
            n = min(num_tiles, len(files))
            self.num_tiles.append(n)
            self.slide_grids.append(_chunks([int(i) for i in rng.choice(len(files), size=n, replace=False)], self.tiles_per_iter))
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(workers)))
        self.slide_num = -1

    def __len__(self):
        return int(sum(len(g) for g in self.slide_grids))

    def reset_counter(self):
        self.slide_num = -1

    def _read(self, paths) -> torch.Tensor:
        out = np.empty((len(paths), self.tile_size, self.tile_size, 3), np.uint8)
        list(self.pool.map(lambda jp: read_tile_into(jp[1], out[jp[0]]), enumerate(paths)))
        return torch.from_numpy(out)

    def __iter__(self):
        for k, (name, files, y, _) in enumerate(self.slides):
            self.slide_num = k
            grid = self.slide_grids[k]
            for ci, idxs in enumerate(grid):
                yield {"Data": self._read([files[i] for i in idxs]), "Label": torch.tensor([y], dtype=torch.int64),
                       "Is Last Batch": ci == len(grid) - 1, "Initial Num Tiles": self.num_tiles[k], "Slide Filename": name,
                       "Slide DataSet": self.dataset_name, "Patch Loc": list(idxs)}


class SyntheticSlides:
    """``InferTiles`` interface over seeded synthetic tiles (slide k has label k % 2)."""

    def __init__(self, n_slides: int = 4, tiles_per_slide: int = 32, tile_size: int = 256, tiles_per_iter: int = 500, seed: int = 4321,
                 mean=(0.8998, 0.8253, 0.9357), std=(0.1125, 0.1751, 0.0787)):
        self.n, self.tile_size, self.tiles_per_iter, self.seed = n_slides, tile_size, max(1, tiles_per_iter), seed
        self.mean, self.std = torch.tensor(mean) * 255.0, torch.tensor(std) * 255.0
        self.image_file_names = [f"synthetic_{k}" for k in range(n_slides)]
        self.num_tiles = [tiles_per_slide] * n_slides
        self.slide_num = -1

    def __len__(self):
        return sum(len(_chunks(range(n), self.tiles_per_iter)) for n in self.num_tiles)

    def reset_counter(self):
        self.slide_num = -1

    def __iter__(self):
        for k in range(self.n):
            self.slide_num = k
            g = torch.Generator().manual_seed(self.seed + k)
            grid = _chunks(list(range(self.num_tiles[k])), self.tiles_per_iter)
            for ci, idxs in enumerate(grid):
                yield {"Data": _synth((len(idxs), self.tile_size, self.tile_size), g, self.mean, self.std),
                       "Label": torch.tensor([k % 2], dtype=torch.int64), "Is Last Batch": ci == len(grid) - 1,
                       "Initial Num Tiles": self.num_tiles[k], "Slide Filename": self.image_file_names[k], "Slide DataSet": "synthetic",
                       "Patch Loc": list(idxs)}


# --------------------------------------------------------------------------- #
# hand-over to HBM
# --------------------------------------------------------------------------- #
class DevicePrefetcher:
    """Iterate a source as device-resident batches, one batch ahead of the consumer.

    Two PINNED host staging buffers and two device buffers rotate.  While the step on batch i is being launched,
    batch i+1 is already being read (``source.fill`` reader threads write straight into pinned memory; sources
    without ``fill`` are copied into it) and its ``copy_(non_blocking=True)`` is queued on a dedicated copy stream,
    so the H2D transfer overlaps the step.  No Python thread touches the launch loop (a prefetch thread contends
    for the GIL with it: measured 26.6 vs 18.5 ms/step in round 1).  ``'Data'`` / ``'Target'`` come back as device
    tensors valid until the next-but-one ``next()``."""

    def __init__(self, source, device, tile_shape: Tuple[int, int, int, int], augmenter=None):
        """``augmenter`` (gipvit.augment.TileAugmenter): its per-tile draws for the batch are made while the batch is
        staged and travel in the same pinned slot / on the same copy stream (``'AugParams'``, ``'Fill'`` in the batch dict);
        a pageable copy issued from the launch loop would block it behind the tile transfer."""
        self.src, self.dev = source, torch.device(device)
        self.copy = torch.cuda.Stream(self.dev)
        B = tile_shape[0]
        self.aug = augmenter
        if augmenter is not None:
            rec = augmenter._DT.itemsize
            self.pin_a = [torch.empty(B * rec, dtype=torch.uint8).pin_memory() for _ in range(2)]
            self.pin_f = [torch.zeros(B, 8, dtype=torch.float32).pin_memory() for _ in range(2)]
            self.d_a = [torch.empty(B * rec, dtype=torch.uint8, device=self.dev) for _ in range(2)]
            self.d_f = [torch.zeros(B, 8, dtype=torch.float32, device=self.dev) for _ in range(2)]
            self.has_fill = [False, False]
        self.pin = [torch.empty(tile_shape, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.pin_t = [torch.empty((B, 1), dtype=torch.int64).pin_memory() for _ in range(2)]
        self.dbuf = [torch.empty(tile_shape, dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self.dtgt = [torch.empty((B, 1), dtype=torch.int64, device=self.dev) for _ in range(2)]
        self.ready = [torch.cuda.Event() for _ in range(2)]       # H2D of slot k done
        self.freed = [None, None]                                   # consumer's last use of slot k (recorded on its stream)

    def __len__(self):
        return len(self.src)

    def _submit(self, k: int, it) -> bool:
        """Start filling pinned slot k with the next batch (reader threads); False at the end of the epoch."""
        self._futs[k] = []
        if self._on_device:                       # the source draws its batch on the GPU (_h2d): nothing to stage on the host
            if self._left == 0:
                return False
            self._left -= 1
        elif hasattr(self.src, "fill"):
            if self._left == 0:
                return False
            self._left -= 1
            self._futs[k] = self.src.fill(self.pin[k].numpy(), self.pin_t[k].numpy())
        else:
            mb = next(it, None)
            if mb is None:
                return False
            self.pin[k].copy_(mb["Data"]); self.pin_t[k].copy_(mb["Target"].view(-1, 1))
        if self.aug is not None:
            packed, fill = self.aug.pack_batch(self.aug.sample_batch(self.pin[k].shape[0]))
            self.pin_a[k].numpy()[:] = packed
            self.has_fill[k] = fill is not None
            if fill is not None:
                self.pin_f[k].numpy()[:] = fill
        return True

    def _h2d(self, k: int):
        """Wait for slot k's reads, then queue its host-to-device copy on the copy stream."""
        for f in self._futs[k]:
            f.result()
        if self.freed[k] is not None:
            self.copy.wait_event(self.freed[k])                   # the step that read device slot k has finished
        with torch.cuda.stream(self.copy):
            if self._on_device:
                self.src.fill_device(self.dbuf[k], self.dtgt[k])
            else:
                self.dbuf[k].copy_(self.pin[k], non_blocking=True)
                self.dtgt[k].copy_(self.pin_t[k], non_blocking=True)
            if self.aug is not None:
                self.d_a[k].copy_(self.pin_a[k], non_blocking=True)
                if self.has_fill[k]:
                    self.d_f[k].copy_(self.pin_f[k], non_blocking=True)
            self.ready[k].record(self.copy)

    def __iter__(self):
        it = None
        self._futs = [[], []]
        self._on_device = hasattr(self.src, "fill_device") and self.dev.type == "cuda"
        if self._on_device:
            self.src.start_epoch_device(self.dev)
            self._left = len(self.src)
        elif hasattr(self.src, "fill"):
            self.src._start_epoch()
            self._left = len(self.src)
        else:
            it = iter(self.src)
        valid = [self._submit(0, it), False]
        if valid[0]:
            self._h2d(0)
            valid[1] = self._submit(1, it)
        k = 0
        while valid[k]:
            torch.cuda.current_stream().wait_event(self.ready[k])
            mb = {"Data": self.dbuf[k], "Target": self.dtgt[k]}
            if self.aug is not None:
                mb["AugParams"], mb["Fill"] = self.d_a[k], (self.d_f[k] if self.has_fill[k] else None)
            yield mb
            # the consumer has launched its step on batch i (slot k); batch i+1's reads ran meanwhile
            ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
            self.freed[k] = ev
            if valid[k ^ 1]:
                self._h2d(k ^ 1)                   # overlaps the step just launched
            self.ready[k].synchronize()            # pinned slot k has left the host (queued before the step: normally long done)
            valid[k] = self._submit(k, it)         # batch i+2: its reads overlap the launch of step i+1
            k ^= 1
