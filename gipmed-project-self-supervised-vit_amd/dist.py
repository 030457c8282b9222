"""Data-parallel reduction over RCCL (torch.distributed backend "nccl" on ROCm).

The reference wraps the model in NativeDDP (train.py:624-635): bucketed all-reduce of
gradients inside loss.backward().  Here gradients live in ONE flat arena ordered by
backward completion, so the engine issues a few large all-reduces over contiguous ranges
as soon as a range is final (head first, then the backbone) -- each one runs on the
process group's communication stream while the main stream keeps computing, and
``finish()`` joins them.  Mean = SUM here, the 1/world factor is folded into the fused
optimizer kernel (gv_adamw_ema grad_scale).  Works unchanged on gloo/CPU tensors (tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class RcclReducer:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._pending = []

    def reduce_range(self, buf: torch.Tensor, lo: int, hi: int):
        if hi > lo:
            self._pending.append(dist.all_reduce(buf[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def reduce_tensor(self, t: torch.Tensor):
        self._pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for w in self._pending:
            w.wait()            # current stream waits for the communication stream
        self._pending.clear()


def shard_range(n_items: int, rank: int, world: int):
    """Tiles [lo, hi) owned by `rank` when a global batch is split evenly."""
    per = n_items // world
    return rank * per, (rank + 1) * per
