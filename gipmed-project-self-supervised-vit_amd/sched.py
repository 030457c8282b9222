"""Learning-rate plumbing of the driver: batch-size scaling (reference train.py:569-581) and the
cosine / step schedules with linear warm-up that timm's create_scheduler_v2 gives the reference
(train.py:881-887), plus the DINO cosine schedules (paper: momentum, weight decay)."""
from __future__ import annotations

import math


def scaled_lr(lr, lr_base, batch_size, world_size, lr_base_size=256, lr_base_scale="", opt="sgd"):
    """train.py:569-581: lr = lr_base * (global_batch / lr_base_size) [sqrt for ada*/lamb]."""
    if lr:
        return lr
    ratio = batch_size * world_size / lr_base_size
    scale = lr_base_scale or ("sqrt" if any(o in opt.lower() for o in ("ada", "lamb")) else "linear")
    return lr_base * (math.sqrt(ratio) if scale == "sqrt" else ratio)


class LrSchedule:
    def __init__(self, lr, sched="cosine", epochs=300, warmup_epochs=5, warmup_lr=1e-5, min_lr=0.0, updates_per_epoch=1,
                 decay_epochs=90, decay_rate=0.1, on_updates=False):
        if sched not in ("cosine", "step"):
            raise ValueError(f"sched {sched!r}: 'cosine' or 'step' (nothing is substituted for timm's other schedulers)")
        self.lr, self.sched, self.E, self.WE, self.wlr, self.min_lr = lr, sched, epochs, warmup_epochs, warmup_lr, min_lr
        self.upe, self.decay_epochs, self.decay_rate, self.on_updates = max(1, updates_per_epoch), decay_epochs, decay_rate, on_updates

    def at(self, epoch: int, update_in_epoch: int = 0) -> float:
        t = epoch + (update_in_epoch / self.upe if self.on_updates else 0.0)
        if self.WE > 0 and t < self.WE:
            return self.wlr + (self.lr - self.wlr) * t / self.WE
        if self.sched == "cosine":
            return self.min_lr + 0.5 * (self.lr - self.min_lr) * (1 + math.cos(math.pi * min(t, self.E) / self.E))
        return self.lr * (self.decay_rate ** int(t // self.decay_epochs))          # "step"


def cosine_between(start: float, end: float, step: int, total: int) -> float:
    """DINO's cosine_scheduler without warm-up: value at `step` of `total`."""
    if total <= 1:
        return end
    return end + 0.5 * (start - end) * (1 + math.cos(math.pi * min(step, total - 1) / (total - 1)))
