"""Stochastic depth draws for ``--drop-path`` (reference train.py:287-288 -> timm ``create_model(drop_path_rate=...)``:
``DropPath`` modules around both residual branches of every block, vit.pyc@L66-74; block i drops with probability
rate * i / (depth - 1), vit.pyc@L186).

The draws are made on the host, one Bernoulli per (block, branch, image), and travel to the device as the factors
keep_mask / keep_prob; the engine expands them to one factor per token row (``gv_expand_rows``) and the kernels apply them
in the residual-add epilogues and to the gradient that enters each branch (include/gipvit.h: ``row_scale`` / ``gb_scale``).
Same distribution as timm's ``drop_path`` (per sample, scale_by_keep=True), not the same random stream."""
from __future__ import annotations

import numpy as np
import torch


class DropPathSampler:
    def __init__(self, depth: int, n_img: int, rate: float, seed: int, device):
        if not 0.0 <= rate < 1.0:
            raise ValueError(f"drop-path rate {rate}: need 0 <= rate < 1")
        self.depth, self.n_img, self.rate = depth, n_img, float(rate)
        self.keep = np.array([1.0 - rate * i / max(depth - 1, 1) for i in range(depth)], dtype=np.float32)
        self.rng = np.random.default_rng(seed)
        self.pin = [torch.empty(depth, 2, n_img, dtype=torch.float32).pin_memory() for _ in range(2)] if torch.device(device).type == "cuda" else None
        self.dev = [torch.empty(depth, 2, n_img, dtype=torch.float32, device=device) for _ in range(2)]
        self.copied = [None, None]      # per pinned slot: the event behind its last H2D copy (a slot is rewritten only after it)
        self.k = 0

    def sample_host(self) -> np.ndarray:
        """f32 [depth, 2, n_img]: 0 where the branch is dropped, 1 / keep_prob where it is kept."""
        u = self.rng.random((self.depth, 2, self.n_img), dtype=np.float32)
        keep = self.keep[:, None, None]
        return np.where(u < keep, 1.0 / keep, 0.0).astype(np.float32)

    def sample(self) -> torch.Tensor:
        """The next step's factors on the device (two slots alternate: the previous step may still be reading its own).
        A host loop may run any number of steps ahead of the GPU: before a pinned slot is rewritten, the event recorded behind
        its previous copy is waited for (that copy is stream-ordered behind the step that last read the device slot, so the
        device slot is free as well)."""
        self.k ^= 1
        f = self.sample_host()
        if self.pin is None:
            self.dev[self.k].copy_(torch.from_numpy(f))
        else:
            if self.copied[self.k] is not None:
                self.copied[self.k].synchronize()
            self.pin[self.k].numpy()[:] = f
            self.dev[self.k].copy_(self.pin[self.k], non_blocking=True)
            if self.copied[self.k] is None:
                self.copied[self.k] = torch.cuda.Event()
            self.copied[self.k].record()
        return self.dev[self.k]

    def state_dict(self):
        """The draw stream's position (checkpoint 'extra' dict): a resumed --drop-path run continues the stream instead of
        replaying epoch 0's masks."""
        import json
        return {"rng": json.dumps(self.rng.bit_generator.state), "k": self.k}      # plain str / int: loads with weights_only=True

    def load_state_dict(self, st):
        import json
        self.rng.bit_generator.state = json.loads(st["rng"])
        self.k = int(st.get("k", 0))
