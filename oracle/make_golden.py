"""Generate tests/golden/*.npz from the oracle (run on CPU: python -m oracle.make_golden).

The reference ships no fixtures and cannot be imported here, so these vectors
are produced by the CPU restatement itself, after tests/test_oracle.py has pinned it
against independent implementations.  They freeze the oracle against drift and let the
GPU tests check the HIP path without re-running the oracle at the larger sizes."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import step_oracle as so, vit_oracle as vo

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def supervised_c1(steps=100):
    """BASELINE config 1: ViT-T/16, 64x64, supervised head, B=8, AdamW lr 1e-4 (100-step loss curve: the length north_star gates at 1e-3)."""
    orc = so.SupervisedOracle(arch="vit_tiny", img_size=64, num_classes=2, seed=0, lr=1e-4, wd=0.05)
    tiles = vo.synth_tiles(8, 64, seed=1234)
    tgt = torch.randint(0, 2, (8, 1), generator=torch.Generator().manual_seed(5))
    loss0, grads0, logits0 = orc.forward_backward(tiles, tgt)
    curve = [orc.step(tiles, tgt)["loss"] for _ in range(steps)]
    np.savez_compressed(os.path.join(OUT, "supervised_c1.npz"), loss0=float(loss0), logits0=logits0.numpy(),
                        grad_norm0=so.grad_norm(grads0), curve=np.array(curve, dtype=np.float64),
                        g_qkv0=grads0["blocks.0.attn.qkv.weight"].numpy()[:8, :16], g_head=grads0["head.weight"].numpy())


def dino_tiny():
    """ViT-T, 2x224 + 8x96 crops of two 256-px tiles, K = 4096: one forward/backward."""
    orc = so.DinoOracle(arch="vit_tiny", img_size=224, out_dim=4096, seed=0)
    tiles = vo.synth_tiles(2, 256, seed=1234)
    loss, grads, s_out, t_out, bsum = orc.forward_backward(tiles)
    np.savez_compressed(os.path.join(OUT, "dino_tiny.npz"), loss=float(loss), grad_norm=so.grad_norm(grads),
                        student=s_out.numpy()[:, :64], teacher=t_out.numpy()[:, :64], center_sum=bsum.numpy()[0, :256],
                        g_pos=grads["backbone.pos_embed"].numpy()[0, :4, :32],
                        g_last=grads["head.last_layer.weight_v"].numpy()[:4, :32])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    supervised_c1()
    dino_tiny()
    print("wrote", sorted(os.listdir(OUT)))
