"""Model registry + parameter initialisation + encoder-checkpoint I/O for the hot path.

Counterpart of ``create_model(args.model, ...)`` (reference train.py:482-495) for the ViT
names the reference's documented commands use (train_instruct.txt:16-34) and of the
DINO ViT factory functions (vit.pyc@L275-293).  State-dict keys are timm's / DINO's
(SURVEY.md section 5), so encoder checkpoints interchange with the reference's.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import torch

from .engine import ARCHS, dino_head_specs, vit_param_specs

# timm names used by the reference -> (arch, default img size after the documented cfg patch)
MODEL_REGISTRY = {
    "vit_tiny_patch16_224": "vit_tiny", "vit_small_patch16_224": "vit_small", "vit_base_patch16_224": "vit_base",
    "vit_small_patch16_224_dino": "vit_small", "vit_base_patch16_224_dino": "vit_base",
    "vit_tiny": "vit_tiny", "vit_small": "vit_small", "vit_base": "vit_base",
}


def resolve_arch(name: str) -> str:
    if name not in MODEL_REGISTRY:
        raise ValueError(f"unknown model '{name}'; this build covers {sorted(MODEL_REGISTRY)}")
    return MODEL_REGISTRY[name]


def _trunc_normal_(t: torch.Tensor, std: float, gen: torch.Generator):
    # vit.pyc@L25-63: truncated normal on the absolute interval [-2, 2] via uniform -> erfinv
    lo = (1.0 + math.erf(-2.0 / std / math.sqrt(2.0))) / 2.0
    hi = (1.0 + math.erf(2.0 / std / math.sqrt(2.0))) / 2.0
    t.uniform_(2 * lo - 1, 2 * hi - 1, generator=gen).erfinv_().mul_(std * math.sqrt(2.0)).clamp_(-2.0, 2.0)
    return t


def init_vit_state(arch: str, img_size: int, num_classes: int = 0, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """vit.pyc@L173-211: trunc_normal(.02) on pos/cls/Linear weights, zero biases, LayerNorm (1, 0);
    the patch-embedding conv keeps torch's default (kaiming-uniform, bound 1/sqrt(fan_in))."""
    gen = torch.Generator().manual_seed(seed)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    bound = 1.0 / math.sqrt(3 * 16 * 16)
    for name, shape in vit_param_specs(arch, img_size, num_classes).items():
        t = torch.empty(shape)
        if name.startswith("patch_embed"):
            t.uniform_(-bound, bound, generator=gen)
        elif name in ("cls_token", "pos_embed") or (name.endswith(".weight") and len(shape) == 2):
            _trunc_normal_(t, 0.02, gen)
        elif name.endswith("weight"):       # LayerNorm weight
            t.fill_(1.0)
        else:
            t.zero_()
        out[name] = t
    return out


def init_dino_head_state(in_dim: int, out_dim: int, hidden: int = 2048, bottleneck: int = 256, seed: int = 1):
    """vit.pyc@L296-324."""
    gen = torch.Generator().manual_seed(seed)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in dino_head_specs(in_dim, out_dim, hidden, bottleneck).items():
        t = torch.empty(shape)
        if name == "last_layer.weight_g":
            t.fill_(1.0)
        elif name == "last_layer.weight_v":
            b = 1.0 / math.sqrt(bottleneck)
            t.uniform_(-b, b, generator=gen)
        elif name.endswith(".weight"):
            _trunc_normal_(t, 0.02, gen)
        else:
            t.zero_()
        out[name] = t
    return out


def resize_pos_embed(pos: torch.Tensor, new_tokens: int) -> torch.Tensor:
    """Bicubic resize of a [1, 1+P, D] pos-embed to a new grid (what timm's checkpoint filter
    does when the reference loads 224-px DINO weights into the 256-px cfg)."""
    if pos.shape[1] == new_tokens:
        return pos
    import torch.nn.functional as F
    cls, grid = pos[:, :1], pos[:, 1:]
    s0, s1 = int(math.sqrt(grid.shape[1])), int(math.sqrt(new_tokens - 1))
    grid = F.interpolate(grid.reshape(1, s0, s0, -1).permute(0, 3, 1, 2), size=(s1, s1), mode="bicubic", align_corners=False)
    return torch.cat([cls, grid.permute(0, 2, 3, 1).reshape(1, s1 * s1, -1)], dim=1)


def load_encoder_checkpoint(path: str, arch: str, img_size: int, num_classes: int = 0) -> Dict[str, torch.Tensor]:
    """Read an encoder ``state_dict`` from a timm-style ``*.pth.tar`` / DINO checkpoint with a
    loader that executes nothing from the file, strip 'module.' / 'backbone.' prefixes, resize
    the pos-embed if the grid differs and fill a missing / mismatched classifier head."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("state_dict", "model", "teacher", "student"):
        if isinstance(ck, dict) and key in ck and isinstance(ck[key], dict):
            ck = ck[key]
            break
    sd = {}
    for k, v in ck.items():
        for pre in ("module.", "backbone."):
            if k.startswith(pre):
                k = k[len(pre):]
        sd[k] = v
    want = vit_param_specs(arch, img_size, num_classes)
    fresh = init_vit_state(arch, img_size, num_classes)
    out = OrderedDict()
    for name, shape in want.items():
        t = sd.get(name)
        if name == "pos_embed" and t is not None:
            t = resize_pos_embed(t.float(), shape[1])
        if t is None or tuple(t.shape) != tuple(shape):
            t = fresh[name]
        out[name] = t.float()
    return out
