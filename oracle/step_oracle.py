"""CPU restatement of one training step (TEST INFRASTRUCTURE, see oracle/__init__.py).

Supervised step: ordering of /root/reference/train.py:1044-1078 (forward ->
softmax -> loss -> zero_grad -> backward -> clip -> optimizer.step).
DINO step: absent from the reference; DINO paper Alg. 1 -- teacher
forward on the global crops, student forward on all crops, loss, backward,
AdamW, teacher EMA, center update.
"""
from __future__ import annotations

import copy
import math
from typing import Dict, Optional

import torch

from . import vit_oracle as vo


def _leafify(d):
    return {k: v.detach().clone().requires_grad_(True) for k, v in d.items()}


def grad_norm(grads) -> float:
    return math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values() if g is not None))


class DinoOracle:
    """Student + teacher (backbone, head), center, AdamW.  fp32 by default."""

    def __init__(self, arch="vit_small", img_size=224, out_dim=65536, seed=0, lr=5e-4, wd=0.04,
                 momentum_teacher=0.996, student_temp=0.1, teacher_temp=0.04, center_momentum=0.9,
                 n_global=2, n_local=8, gsize=224, lsize=96, clip_grad: Optional[float] = None,
                 hidden_dim=2048, bottleneck_dim=256, dtype=torch.float32):
        self.arch, self.dtype = arch, dtype
        D = vo.ARCHS[arch]["embed_dim"]
        self.p = vo.init_vit(arch, img_size, 0, seed, dtype)
        self.hp = vo.init_dino_head(D, out_dim, seed + 1, hidden_dim, bottleneck_dim, dtype)
        # teacher starts as a copy of the student (DINO main.py: load_state_dict(student))
        self.tp = {k: v.clone() for k, v in self.p.items()}
        self.thp = {k: v.clone() for k, v in self.hp.items()}
        self.center = torch.zeros(1, out_dim, dtype=dtype)
        self.m, self.ts, self.tt, self.cm = momentum_teacher, student_temp, teacher_temp, center_momentum
        self.n_global, self.n_local = n_global, n_local
        self.wins = vo.crop_windows(n_global, n_local, gsize, lsize)
        self.clip = clip_grad
        self.all = {**{"backbone." + k: v for k, v in self.p.items()},
                    **{"head." + k: v for k, v in self.hp.items()}}
        # norm_last_layer=True: weight_g frozen (vit.pyc@L315-318)
        self.opt = vo.AdamW(self.all, lr, wd, frozen=("head.last_layer.weight_g",))

    def crops(self, tiles_u8):
        return [vo.normalize_window(tiles_u8, w, dtype=self.dtype) for w in self.wins]

    def forward_backward(self, tiles_u8, teacher_temp=None, drop=None, dropout=None):
        """``drop``: stochastic-depth factors of the STUDENT pass, [depth, 2, V * B] in crop order (vo.drop_path_factors); the
        teacher runs without (DINO: teacher in eval mode)."""
        crops = self.crops(tiles_u8)
        V, G = len(crops), self.n_global
        with torch.no_grad():
            t_out = vo.multicrop_forward(self.tp, self.thp, crops[:G], self.arch)
        sp, shp = _leafify(self.p), _leafify(self.hp)
        s_out = vo.multicrop_forward(sp, shp, crops, self.arch, drop=drop, dropout=dropout)     # dropout = (p, step seed): student only
        loss, bsum = vo.dino_loss(s_out, t_out, self.center, V, G, self.ts, self.tt if teacher_temp is None else teacher_temp)
        loss.backward()
        grads = {**{"backbone." + k: v.grad for k, v in sp.items()},
                 **{"head." + k: v.grad for k, v in shp.items()}}
        return loss.detach(), grads, s_out.detach(), t_out, bsum

    def step(self, tiles_u8, lr=None, wd=None, momentum_teacher=None, teacher_temp=None, train_last_layer=True, drop=None):
        """One DINO step with this step's schedule values (cosine wd / teacher momentum, teacher-temperature
        warm-up, last layer frozen during the first epochs).  A frozen last layer has NO gradient (DINO
        cancel_gradients_last_layer: p.grad = None), so it is left out of the clip norm and skipped by AdamW."""
        loss, grads, s_out, t_out, bsum = self.forward_backward(tiles_u8, teacher_temp, drop)
        if not train_last_layer:
            grads["head.last_layer.weight_v"] = None
        gn = grad_norm(grads)
        if self.clip is not None and gn > self.clip:
            c = self.clip / (gn + 1e-6)
            grads = {k: (g * c if g is not None else None) for k, g in grads.items()}
        self.opt.step(grads, lr, wd)
        m = self.m if momentum_teacher is None else momentum_teacher
        vo.ema_update(self.tp, self.p, m)
        vo.ema_update(self.thp, self.hp, m)
        self.center = vo.update_center(self.center, bsum, t_out.shape[0], self.cm)
        return dict(loss=float(loss), grad_norm=gn, student_out=s_out, teacher_out=t_out)


class SupervisedOracle:
    """BASELINE config 1: single crop, timm head, softmax -> LabelSmoothingCE."""

    def __init__(self, arch="vit_tiny", img_size=64, num_classes=2, seed=0, lr=1e-3, wd=0.0,
                 smoothing=0.1, dtype=torch.float32, opt="adamw"):
        self.arch, self.dtype, self.img = arch, dtype, img_size
        self.p = vo.init_vit(arch, img_size, num_classes, seed, dtype)
        self.smoothing = smoothing
        self.opt = vo.Lamb(self.p, lr, wd) if opt == "lamb" else vo.AdamW(self.p, lr, wd)

    def forward_backward(self, tiles_u8, target, drop=None, dropout=None):
        x = vo.normalize_window(tiles_u8, (0, 0, self.img), dtype=self.dtype)
        sp = _leafify(self.p)
        logits = vo.vit_logits(sp, x, self.arch, drop=drop, dropout=None if dropout is None else (dropout[0], dropout[1], 0))
        loss = vo.softmax_lsce(logits, target, self.smoothing)
        loss.backward()
        return loss.detach(), {k: v.grad for k, v in sp.items()}, logits.detach()

    def step(self, tiles_u8, target, lr=None, drop=None):
        loss, grads, logits = self.forward_backward(tiles_u8, target, drop)
        gn = grad_norm(grads)
        self.opt.step(grads, lr)
        return dict(loss=float(loss), grad_norm=gn, logits=logits)
