"""Checkpoint files in the layout the reference writes through timm's CheckpointSaver
(reference train.py:867-877, 970-973): ``checkpoint-<epoch>.pth.tar``,
``last.pth.tar``, ``model_best.pth.tar``, ``recovery-<epoch>-<batch>.pth.tar`` holding
``{'epoch', 'arch', 'state_dict', 'optimizer', 'version': 2, 'args', 'metric', 'state_dict_ema'}``
(``state_dict_ema`` = timm's key for the ``--model-ema`` copy; a DINO run stores its teacher there, plus ``dino_center``).

Deviation (documented): ``'args'`` is stored as a plain dict, not a pickled Namespace, and
the optimizer state as tensors, so every file loads with ``torch.load(weights_only=True)``.
"""
from __future__ import annotations

import os
import shutil
from typing import Dict, Optional

import torch


class CheckpointSaver:
    def __init__(self, checkpoint_dir: str, arch: str, args: Optional[dict] = None, decreasing: bool = False, max_history: int = 10):
        self.dir, self.arch, self.args, self.decreasing, self.max_history = checkpoint_dir, arch, dict(args or {}), decreasing, max_history
        os.makedirs(checkpoint_dir, exist_ok=True)
        self.history = []            # (path, metric)
        self.best_metric, self.best_epoch = None, None

    def _payload(self, epoch, state_dict, optimizer, metric, extra):
        p = {"epoch": epoch, "arch": self.arch, "state_dict": {k: v.detach().cpu() for k, v in state_dict.items()}, "version": 2,
             "args": {k: v for k, v in self.args.items() if isinstance(v, (int, float, str, bool, type(None), list, tuple))}}
        if optimizer is not None:
            p["optimizer"] = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in optimizer.items()}
        if metric is not None:
            p["metric"] = float(metric)
        for k, v in (extra or {}).items():       # e.g. 'state_dict_ema' (timm key: the EMA model / DINO teacher), 'dino_center'
            cpu = lambda t: t.detach().cpu() if torch.is_tensor(t) else t          # (plain str / int entries, e.g. a sampler's stream position, pass through)
            p[k] = {n: cpu(t) for n, t in v.items()} if isinstance(v, dict) else cpu(v)
        return p

    def save_checkpoint(self, epoch: int, state_dict: Dict[str, torch.Tensor], optimizer: Optional[dict] = None,
                        metric: Optional[float] = None, extra: Optional[dict] = None):
        last = os.path.join(self.dir, "last.pth.tar")
        torch.save(self._payload(epoch, state_dict, optimizer, metric, extra), last)
        path = os.path.join(self.dir, f"checkpoint-{epoch}.pth.tar")
        shutil.copyfile(last, path)
        self.history.append((path, metric))
        better = metric is not None and (self.best_metric is None or (metric < self.best_metric if self.decreasing else metric > self.best_metric))
        if better:
            self.best_metric, self.best_epoch = metric, epoch
            shutil.copyfile(last, os.path.join(self.dir, "model_best.pth.tar"))
        while len(self.history) > self.max_history:
            old, _ = self.history.pop(0)
            if os.path.exists(old):
                os.remove(old)
        return self.best_metric, self.best_epoch

    def save_recovery(self, epoch: int, batch_idx: int, state_dict, optimizer=None, extra: Optional[dict] = None):
        path = os.path.join(self.dir, f"recovery-{epoch}-{batch_idx}.pth.tar")
        torch.save(self._payload(epoch, state_dict, optimizer, None, dict(extra or {}, batch_idx=batch_idx)), path)
        return path


def load_checkpoint_file(path: str) -> dict:
    """Never executes anything from the file."""
    return torch.load(path, map_location="cpu", weights_only=True)
