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


def comm_setup(world: int, backend: str = "nccl") -> dict:
    """Optionally make room on the chip for RCCL before the process group AND the kernel library exist.  Every heavy launch of this
    build is one workgroup per CU holding the CU's whole register file and LDS, so RCCL's channel workgroups cannot co-reside with
    them.  GIPVIT_COMM_CUS = C > 0 leaves C compute units to communication: the library sizes its launches for 256 - C
    (GIPVIT_CU_BUDGET, read once when it is loaded) and RCCL is capped to C channels (NCCL_MAX_NCHANNELS / NCCL_MIN_NCHANNELS;
    values already in the environment win).  The DEFAULT is C = 0 -- no budget, RCCL's own channel count: measured on one GPU with
    a communication stand-in (tools/comm_standin.py) the budget costs more than it saves at every C tried
    (C = 8: 14.03 ms per step without it, 14.16 with it; launches sized for 248 CUs alone: +0.48 ms), because a 251-workgroup
    launch that finds C CUs taken loses only its tail, while a launch sized for 256 - C loses panel height on every CU.
    Returns what was decided, for the log / JSON line.  No-op at world size 1 and for the gloo rehearsal transport."""
    import os
    import sys
    info = {"comm_cus": 0, "cu_budget": int(os.environ.get("GIPVIT_CU_BUDGET", "256"))}
    if world <= 1 or backend != "nccl":
        return info
    c = int(os.environ.get("GIPVIT_COMM_CUS", "0"))
    if c > 0:
        if "gipvit._lib" in sys.modules and "GIPVIT_CU_BUDGET" not in os.environ:
            raise RuntimeError("comm_setup() must run before the kernel library is loaded (the CU budget is read once at load)")
        os.environ.setdefault("GIPVIT_CU_BUDGET", str(256 - c))
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(c))
        os.environ.setdefault("NCCL_MIN_NCHANNELS", str(min(c, 4)))
    info.update(comm_cus=c, cu_budget=int(os.environ.get("GIPVIT_CU_BUDGET", "256")), nccl_max_nchannels=os.environ.get("NCCL_MAX_NCHANNELS"))
    return info


def quiet_init_process_group(backend: str, **kw):
    """init_process_group with file descriptor 1 pointed at stderr meanwhile: gloo's transport prints "[Gloo] Rank 0 is connected ..."
    on stdout, where bench.py owes the driver exactly one JSON line."""
    import os
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        dist.init_process_group(backend, **kw)
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


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


def reduction_plan(head_span, block_spans, n_total: int, bucket_bytes: int = 25 << 20, elt: int = 4):
    """Contiguous arena ranges of one step's gradient all-reduce, in the order they become final in backward.

    ``head_span``: (lo, hi) of the head's weight-decayed matrices (final when the head backward ends; None when there is no
    head), ``block_spans``: {block index: (lo, hi)} of every transformer block's matrices -- the arena lays them out in
    backward-completion order (head, block depth-1 .. 0, patch embed, then the no-decay tensors), ``n_total``: arena length.
    Returns ``[(trigger, lo, hi)]``: trigger "head" (released when the head backward ends), a block index (released when that
    block's backward ends) or "end" (released when backward ends).  Blocks are coalesced until a range holds at least
    ``bucket_bytes`` (the reference's DDP reduces 25-MB buckets, train.py:634: a 7-MB ViT-S block alone is a latency-bound
    message).  The LAST range is block 0 together with everything behind it in the arena -- patch embed and the no-decay
    tensors, final only when backward ends -- so one message, not a block range plus a small tail, is issued there; to keep
    that un-overlapped message short, the bucket in front of it is closed at block 1 whatever its size.
    The ranges tile [0, n_total) exactly once, in release order."""
    plan, lo = [], 0
    if head_span is not None:
        plan.append(("head", 0, head_span[1]))
        lo = head_span[1]
    order = sorted(block_spans, reverse=True)
    for i in order:
        b_lo, b_hi = block_spans[i]
        assert b_lo >= lo, "block spans must follow the arena's backward-completion order"
        if i == order[-1]:
            break
        if (b_hi - lo) * elt >= bucket_bytes or (len(order) > 1 and i == order[-2]):
            plan.append((i, lo, b_hi))
            lo = b_hi
    if lo < n_total:
        plan.append(("end", lo, n_total))
    return plan


def shard_range(n_items: int, rank: int, world: int):
    """Tiles [lo, hi) owned by `rank` when a global batch is split evenly."""
    per = n_items // world
    return rank * per, (rank + 1) * per
