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


# --------------------------------------------------------------------------- #
# the model-object seam of the reference driver (train.py:482-510, 615-622, 1045)
# --------------------------------------------------------------------------- #
class Param:
    """What the reference touches on a parameter: ``.data`` (an f32 view into the engine's arena), ``.requires_grad``
    (train.py:497-503 sets it), ``.shape`` / ``.numel()`` (the parameter-count log line, train.py:512-514)."""

    def __init__(self, name: str, data: torch.Tensor):
        self.name, self.data, self.requires_grad = name, data, True
        self.shape = data.shape

    def numel(self) -> int:
        return self.data.numel()


class _Head:
    """``model.head`` (timm's classifier ``nn.Linear``): ``.weight`` / ``.bias`` / ``.parameters()`` (train.py:500-502)."""

    def __init__(self, weight: Param, bias: Param):
        self.weight, self.bias = weight, bias
        self.in_features, self.out_features = weight.shape[1], weight.shape[0]

    def parameters(self):
        return iter((self.weight, self.bias))


class VitModel:
    """The object ``create_model`` returns: the attributes and methods the reference driver uses on timm's
    ``VisionTransformer`` (``num_classes``, ``head``, ``no_weight_decay()``, ``set_grad_checkpointing()``, ``parameters()``,
    ``state_dict()`` / ``load_state_dict()``, ``train()`` / ``eval()``, ``model(input)``), backed by a ``SupervisedEngine``
    (the hot path itself: HIP kernels behind the C ABI, explicit backward -- there is no ``nn.Module`` and no autograd)."""

    def __init__(self, engine, arch: str):
        self.engine, self.arch = engine, arch
        self.num_classes, self.embed_dim, self.num_features = engine.C, engine.D, engine.D
        a = engine.arena
        self._params = OrderedDict((n, Param(n, a.view(a.p, n))) for n in a.specs)
        self.head = _Head(self._params["head.weight"], self._params["head.bias"])
        self.training = True

    def parameters(self):
        return iter(self._params.values())

    def named_parameters(self):
        return iter(self._params.items())

    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}               # vit.pyc@L208-211

    def set_grad_checkpointing(self, enable: bool = True):
        # the explicit backward keeps every block's activations resident (4 GB at B = 64 on a 288-GB part): nothing to do
        self.grad_checkpointing = bool(enable)

    @property
    def backbone_trainable(self) -> bool:
        """False after the reference's ``--no-grad`` pattern (every parameter frozen, then the head's re-enabled)."""
        return any(p.requires_grad for n, p in self._params.items() if not n.startswith("head."))

    def apply_requires_grad(self):
        """Hand the ``requires_grad`` flags to the engine (it steps either everything or the classifier only)."""
        head_on = self.head.weight.requires_grad and self.head.bias.requires_grad
        if not head_on:
            raise ValueError("a frozen classifier head is not a configuration of the reference driver")
        self.engine.train_backbone = self.backbone_trainable
        return self

    def state_dict(self):
        return self.engine.state_dict()

    def load_state_dict(self, state: Dict[str, torch.Tensor], strict: bool = True):
        want = set(self._params)
        missing, extra = sorted(want - set(state)), sorted(set(state) - want)
        if strict and (missing or extra):
            raise KeyError(f"load_state_dict: missing {missing[:4]}, unexpected {extra[:4]}")
        cur = self.engine.state_dict()
        cur.update({k: v for k, v in state.items() if k in want})
        self.engine.load_state(cur)
        return missing, extra

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def cuda(self, *a, **k):
        return self

    to = cuda

    def forward_features(self, tiles_u8: torch.Tensor) -> torch.Tensor:
        return self.engine.forward(tiles_u8)[1]

    def __call__(self, tiles_u8: torch.Tensor) -> torch.Tensor:
        """NHWC uint8 tiles [batch, H, W, 3] already on the device -> logits f32 [batch, num_classes] (the engine's buffer)."""
        return self.engine.forward(tiles_u8)[0]


def create_model(model_name: str, pretrained: bool = False, in_chans: int = 3, num_classes: Optional[int] = None,
                 drop_rate: float = 0.0, drop_path_rate: Optional[float] = None, checkpoint_path: str = "",
                 img_size: int = 256, batch: int = 8, device: str = "cuda:0", seed: int = 0, **engine_kwargs) -> VitModel:
    """Counterpart of ``timm.create_model`` as the reference calls it (train.py:482-495) for the ViT names it documents.
    Unsupported requests fail here instead of being ignored: other channel counts, dropout (not built; stochastic depth is),
    ``pretrained`` without a checkpoint file (there is no network).  ``num_classes=None`` keeps the checkpoint-less default
    of the reference's runs (2 classes, train_instruct.txt:16-34)."""
    from .engine import SupervisedEngine
    arch = resolve_arch(model_name)
    if in_chans != 3:
        raise ValueError(f"create_model: in_chans={in_chans}; the hot path is built for RGB tiles")
    if drop_rate and not 0.0 <= drop_rate < 1.0:
        raise ValueError(f"create_model: drop_rate {drop_rate}: need 0 <= rate < 1")
    if pretrained and not checkpoint_path:
        raise ValueError("create_model: pretrained=True needs checkpoint_path (no network on this system)")
    C = 2 if num_classes is None else int(num_classes)
    eng = SupervisedEngine(arch=arch, img_size=img_size, num_classes=C, batch=batch, device=device, **engine_kwargs)
    state = load_encoder_checkpoint(checkpoint_path, arch, img_size, C) if checkpoint_path else init_vit_state(arch, img_size, C, seed=seed)
    eng.load_state(state)
    model = VitModel(eng, arch)
    # stochastic depth: the training loop hands the next step's draws to the engine -- engine.set_drop_path(model.drop_path.sample())
    # dropout: like the stochastic-depth draws, the training loop hands each step its seed -- engine.set_dropout(model.drop_rate, seed)
    model.drop_rate = float(drop_rate or 0.0)
    model.drop_path_rate = float(drop_path_rate or 0.0)
    model.drop_path = None
    if model.drop_path_rate:
        from .droppath import DropPathSampler
        model.drop_path = DropPathSampler(ARCHS[arch]["depth"], batch, model.drop_path_rate, seed, device)
    return model
