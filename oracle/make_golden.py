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


def dino_curve_schedule(step: int, steps: int = 100, per_epoch: int = 25, B: int = 8):
    """The DINO recipe's per-step values (paper / SURVEY row D5), restated independently of the product's sched.py:
    lr = 5e-4 * B / 256 with a one-epoch linear warm-up from 1e-6 then cosine to 1e-6; weight decay cosine 0.04 -> 0.4;
    teacher momentum cosine 0.996 -> 1; teacher temperature 0.04 -> 0.07 linearly over the first 2 epochs; the head's
    last layer frozen during epoch 0."""
    import math
    cos = lambda a, b, t, n: b + 0.5 * (a - b) * (1 + math.cos(math.pi * min(t, n - 1) / (n - 1)))
    epoch, peak, floor = step // per_epoch, 5e-4 * B / 256.0, 1e-6
    t = step / per_epoch                                            # fractional epoch (schedule stepped per update)
    epochs = steps // per_epoch
    lr = floor + (peak - floor) * t / 1.0 if t < 1.0 else floor + 0.5 * (peak - floor) * (1 + math.cos(math.pi * min(t, epochs) / epochs))
    tt = [0.04, 0.07][min(epoch, 1)] if epoch < 2 else 0.07        # np.linspace(0.04, 0.07, 2) then 0.07
    return dict(lr=lr, wd=cos(0.04, 0.4, step, steps), momentum_teacher=cos(0.996, 1.0, step, steps), teacher_temp=tt,
                train_last_layer=epoch >= 1)


def _dino_curve(arch, fname, steps, B=8):
    orc = so.DinoOracle(arch=arch, img_size=224, out_dim=4096, seed=0, clip_grad=3.0)
    curve, gnorm = [], []
    for t in range(steps):
        r = orc.step(vo.synth_tiles(B, 256, seed=5000 + t), **dino_curve_schedule(t, steps, B=B))
        curve.append(r["loss"]); gnorm.append(r["grad_norm"])
    np.savez_compressed(os.path.join(OUT, fname), curve=np.array(curve), grad_norm=np.array(gnorm),
                        center=orc.center.numpy()[0, :512],
                        s_qkv0=orc.p["blocks.0.attn.qkv.weight"].numpy()[:8, :32], t_qkv0=orc.tp["blocks.0.attn.qkv.weight"].numpy()[:8, :32],
                        s_fc2_11=orc.p["blocks.11.mlp.fc2.weight"].numpy()[:8, :32], t_fc2_11=orc.tp["blocks.11.mlp.fc2.weight"].numpy()[:8, :32],
                        s_last=orc.hp["last_layer.weight_v"].numpy()[:8, :32], t_last=orc.thp["last_layer.weight_v"].numpy()[:8, :32],
                        s_pos=orc.p["pos_embed"].numpy()[0, :4, :32], s_mlp4=orc.hp["mlp.4.weight"].numpy()[:8, :32],
                        init_qkv0=vo.init_vit(arch, 224, 0, 0)["blocks.0.attn.qkv.weight"].numpy()[:8, :32])


def dino_tiny_curve(steps=100):
    """100 DINO steps (the length north_star gates at 1e-3): ViT-T, 2x224 + 8x96 crops of eight fresh 256-px tiles per
    step, K = 4096, clip 3.0, the recipe's schedules above.  Keeps the loss curve, the centre and slices of the
    student / teacher weights at the end."""
    _dino_curve("vit_tiny", "dino_tiny_curve.npz", steps)


def dino_small_curve(steps=100):
    """The same 100 recipe steps on the HEADLINE architecture (ViT-S/16: the width whose Linear + LayerNorm products run
    fused in csrc/panel.hip), eight fresh tiles per step.  About ten minutes of CPU, run once; the fixture is committed."""
    _dino_curve("vit_small", "dino_small_curve.npz", steps)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    import sys
    which = sys.argv[1:] or ["supervised_c1", "dino_tiny", "dino_tiny_curve"]
    for name in which:
        globals()[name]()
    print("wrote", sorted(os.listdir(OUT)))
