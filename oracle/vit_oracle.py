"""Pure-PyTorch CPU restatement of the reference's ViT encoder + DINOHead.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Written from the behavioural
spec recovered from ``/root/reference/nn_encoder_arch/__pycache__/
vision_transformer.cpython-37.pyc`` (cited below as
``vit.pyc@L<n>`` = original source line n recorded in the bytecode) and from
``/root/reference/train.py``.  Functional style: parameters live in an ordered
dict keyed by the timm / DINO ``state_dict`` names so the same dict is what the
HIP engine loads.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

# vit.pyc@L275-293: vit_tiny / vit_small / vit_base(patch_size=16)
ARCHS = {
    "vit_tiny": dict(embed_dim=192, depth=12, num_heads=3),
    "vit_small": dict(embed_dim=384, depth=12, num_heads=6),
    "vit_base": dict(embed_dim=768, depth=12, num_heads=12),
}
PATCH = 16
LN_EPS = 1e-6          # vit.pyc@L275-293 norm_layer=partial(LayerNorm, eps=1e-6)
MLP_RATIO = 4

# transformations.py:104-116 -- 'Ron' normalisation (the reference default)
MEAN_RON = (0.8998, 0.8253, 0.9357)
STD_RON = (0.1125, 0.1751, 0.0787)


def trunc_normal_(t: torch.Tensor, std: float, gen: torch.Generator) -> torch.Tensor:
    """vit.pyc@L25-63: uniform -> erfinv truncated normal on [-2, 2] (absolute)."""
    mean, a, b = 0.0, -2.0, 2.0

    def cdf(x):
        return (1.0 + math.erf(x / math.sqrt(2.0))) / 2.0

    lo, hi = cdf((a - mean) / std), cdf((b - mean) / std)
    with torch.no_grad():
        t.uniform_(2 * lo - 1, 2 * hi - 1, generator=gen)
        t.erfinv_()
        t.mul_(std * math.sqrt(2.0))
        t.add_(mean)
        t.clamp_(min=a, max=b)
    return t


def init_vit(arch: str, img_size: int, num_classes: int = 0, seed: int = 0,
             dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """vit.pyc@L173-211 VisionTransformer.__init__ + _init_weights.

    Key names / shapes follow SURVEY.md section 5 (identical in timm)."""
    a = ARCHS[arch]
    D, depth = a["embed_dim"], a["depth"]
    P = (img_size // PATCH) ** 2
    g = torch.Generator().manual_seed(seed)
    p: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def lin(name, out_f, in_f):
        p[name + ".weight"] = trunc_normal_(torch.empty(out_f, in_f, dtype=dtype), 0.02, g)
        p[name + ".bias"] = torch.zeros(out_f, dtype=dtype)

    def ln(name):
        p[name + ".weight"] = torch.ones(D, dtype=dtype)
        p[name + ".bias"] = torch.zeros(D, dtype=dtype)

    p["cls_token"] = trunc_normal_(torch.empty(1, 1, D, dtype=dtype), 0.02, g)
    p["pos_embed"] = trunc_normal_(torch.empty(1, P + 1, D, dtype=dtype), 0.02, g)
    # vit.pyc@L158-165: Conv2d(3, D, 16, 16).  _init_weights touches only
    # Linear / LayerNorm, so the conv keeps torch's default init; the oracle
    # uses the same kaiming-uniform bound drawn from the same generator.
    fan_in = 3 * PATCH * PATCH
    bound = 1.0 / math.sqrt(fan_in)
    p["patch_embed.proj.weight"] = torch.empty(D, 3, PATCH, PATCH, dtype=dtype).uniform_(-bound, bound, generator=g)
    p["patch_embed.proj.bias"] = torch.empty(D, dtype=dtype).uniform_(-bound, bound, generator=g)
    for i in range(depth):
        b = f"blocks.{i}."
        ln(b + "norm1")
        lin(b + "attn.qkv", 3 * D, D)
        lin(b + "attn.proj", D, D)
        ln(b + "norm2")
        lin(b + "mlp.fc1", MLP_RATIO * D, D)
        lin(b + "mlp.fc2", D, MLP_RATIO * D)
    ln("norm")
    if num_classes > 0:
        lin("head", num_classes, D)        # vit.pyc@L198 / timm head
    return p


def init_dino_head(in_dim: int, out_dim: int, seed: int = 1, hidden_dim: int = 2048,
                   bottleneck_dim: int = 256, dtype=torch.float32):
    """vit.pyc@L296-324 DINOHead.__init__ (use_bn=False, nlayers=3, norm_last_layer=True)."""
    g = torch.Generator().manual_seed(seed)
    p: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    dims = [(hidden_dim, in_dim), (hidden_dim, hidden_dim), (bottleneck_dim, hidden_dim)]
    for idx, (o, i) in zip((0, 2, 4), dims):
        p[f"mlp.{idx}.weight"] = trunc_normal_(torch.empty(o, i, dtype=dtype), 0.02, g)
        p[f"mlp.{idx}.bias"] = torch.zeros(o, dtype=dtype)
    # weight_norm(Linear(bottleneck, out_dim, bias=False)); weight_g.fill_(1)
    bound = 1.0 / math.sqrt(bottleneck_dim)
    p["last_layer.weight_g"] = torch.ones(out_dim, 1, dtype=dtype)
    p["last_layer.weight_v"] = torch.empty(out_dim, bottleneck_dim, dtype=dtype).uniform_(-bound, bound, generator=g)
    return p


# --------------------------------------------------------------------------- #
# forward pieces
# --------------------------------------------------------------------------- #
def interpolate_pos_encoding(pos_embed: torch.Tensor, npatch: int, w: int, h: int) -> torch.Tensor:
    """vit.pyc@L213-233."""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    cls_pe = pos_embed[:, 0]
    patch_pe = pos_embed[:, 1:]
    dim = pos_embed.shape[-1]
    w0, h0 = w // PATCH + 0.1, h // PATCH + 0.1
    s = int(math.sqrt(N))
    patch_pe = F.interpolate(patch_pe.reshape(1, s, s, dim).permute(0, 3, 1, 2),
                             scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)), mode="bicubic")
    assert int(w0) == patch_pe.shape[-2] and int(h0) == patch_pe.shape[-1]
    patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((cls_pe.unsqueeze(0), patch_pe), dim=1)


def layer_norm(x, w, b):
    """nn.LayerNorm(D, eps=1e-6), biased variance (spec row A3)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * w + b


def gelu(x):
    """nn.GELU exact-erf (vit.pyc@L88-104 act_layer=nn.GELU)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def dropout_site_seed(seed: int, layer: int, site: int) -> int:
    """Restatement of the product's per-site seed (gipvit.ops.dropout_site_seed): murmur3 finaliser of (step seed, layer, site)."""
    h = (seed ^ (0x85EBCA6B * (layer * 8 + site + 1))) & 0xFFFFFFFF
    h ^= h >> 16; h = (h * 0x85EBCA6B) & 0xFFFFFFFF; h ^= h >> 13; h = (h * 0xC2B2AE35) & 0xFFFFFFFF; h ^= h >> 16
    return h


def dropout_mask(seed: int, first: int, n: int, p: float) -> torch.Tensor:
    """keep / (1 - p) factors of elements [first, first + n) of a dropout site (include/gipvit.h gv_dropout): element i is kept iff
    fmix32(seed + 0x9E3779B9 (i + 1)) >= floor(p 2^32).  nn.Dropout semantics with the product's counter-based stream."""
    import numpy as np
    i = (np.arange(first, first + n, dtype=np.uint64) + 1) & np.uint64(0xFFFFFFFF)
    h = (np.uint64(seed) + np.uint64(0x9E3779B9) * i) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    thr = min(int(p * 4294967296.0), 4294967295)
    return torch.from_numpy((h >= np.uint64(thr)).astype(np.float32) * np.float32(1.0 / (1.0 - p)))      # the product's scale: float32(1 / (1 - p))


def _drop(x, dropout, layer, site):
    """Apply the dropout of (layer, site) to x [B, N, C]; dropout = (p, step seed, first row of this group in the site's row space)."""
    if dropout is None:
        return x
    p, seed, row0 = dropout
    B, N, C = x.shape
    m = dropout_mask(dropout_site_seed(seed, layer, site), row0 * C, B * N * C, p).reshape(B, N, C)
    return x * m.to(x.dtype)


def attention(x, p, pre, num_heads):
    """vit.pyc@L119-131 Attention.forward."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = x @ p[pre + "qkv.weight"].t() + p[pre + "qkv.bias"]
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return y @ p[pre + "proj.weight"].t() + p[pre + "proj.bias"]


def drop_path_factors(depth: int, n_img: int, rate: float, gen: torch.Generator) -> torch.Tensor:
    """Stochastic depth draws (timm DropPath behind --drop-path, vit.pyc@L66-74; train.py:283-288): block i drops a sample's
    residual branch with probability rate * i / (depth - 1) (vit.pyc@L186 linspace rule) and scales the kept ones by
    1 / keep.  Returns f32 [depth, 2, n_img]: the factor keep_mask / keep_prob per block, branch (attention, MLP), image."""
    out = torch.ones(depth, 2, n_img)
    for i in range(depth):
        p_drop = rate * i / max(depth - 1, 1)
        if p_drop > 0:
            keep = 1.0 - p_drop
            out[i] = (torch.rand(2, n_img, generator=gen) < keep).float() / keep
    return out


def block(x, p, i, num_heads, drop=None, dropout=None):
    """vit.pyc@L146-152 Block.forward.  ``drop``: f32 [2, B] stochastic-depth factors of this block's two branches
    (drop_path_factors), None = Identity (rate 0 / evaluation).  ``dropout``: (p, step seed, row0) -- nn.Dropout after attn.proj
    (site 1), after the MLP activation (site 2) and after mlp.fc2 (site 3), with the product's counter-based masks."""
    b = f"blocks.{i}."
    a = attention(layer_norm(x, p[b + "norm1.weight"], p[b + "norm1.bias"]), p, b + "attn.", num_heads)
    a = _drop(a, dropout, i, 1)
    x = x + (a if drop is None else a * drop[0].to(a.dtype)[:, None, None])
    h = layer_norm(x, p[b + "norm2.weight"], p[b + "norm2.bias"])
    h = gelu(h @ p[b + "mlp.fc1.weight"].t() + p[b + "mlp.fc1.bias"])
    h = _drop(h, dropout, i, 2)
    h = h @ p[b + "mlp.fc2.weight"].t() + p[b + "mlp.fc2.bias"]
    h = _drop(h, dropout, i, 3)
    return x + (h if drop is None else h * drop[1].to(h.dtype)[:, None, None])


def prepare_tokens(x, p):
    """vit.pyc@L235-246 (PatchEmbed L167-170 = conv k16 s16 -> flatten -> transpose)."""
    B, _, w, h = x.shape
    t = F.conv2d(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=PATCH)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((p["cls_token"].expand(B, -1, -1), t), dim=1)
    return t + interpolate_pos_encoding(p["pos_embed"], t.shape[1] - 1, w, h)


def vit_features(p, x, arch: str, return_tokens: bool = False, drop=None, dropout=None):
    """vit.pyc@L248-253 VisionTransformer.forward -> x[:, 0] after the final norm.  ``drop``: [depth, 2, B] stochastic-depth
    factors (training with --drop-path) or None.  ``dropout``: (p, step seed, row0) for --drop (pos_drop is site 0 of layer 0)."""
    a = ARCHS[arch]
    t = _drop(prepare_tokens(x, p), dropout, 0, 0)
    for i in range(a["depth"]):
        t = block(t, p, i, a["num_heads"], None if drop is None else drop[i], dropout)
    t = layer_norm(t, p["norm.weight"], p["norm.bias"])
    return t if return_tokens else t[:, 0]


def vit_logits(p, x, arch: str, drop=None, dropout=None):
    """timm variant used at runtime (train.py:482-495): CLS -> head Linear."""
    f = vit_features(p, x, arch, drop=drop, dropout=dropout)
    return f @ p["head.weight"].t() + p["head.bias"]


def dino_head(hp, x):
    """vit.pyc@L326-330 DINOHead.forward; weight_norm: w = g * v / ||v||_row."""
    x = gelu(x @ hp["mlp.0.weight"].t() + hp["mlp.0.bias"])
    x = gelu(x @ hp["mlp.2.weight"].t() + hp["mlp.2.bias"])
    x = x @ hp["mlp.4.weight"].t() + hp["mlp.4.bias"]
    x = F.normalize(x, dim=-1, p=2)
    v = hp["last_layer.weight_v"]
    w = hp["last_layer.weight_g"] * v / v.norm(dim=1, keepdim=True)
    return x @ w.t()


# --------------------------------------------------------------------------- #
# input contract (SURVEY 8a row I0 / 8d synthetic inputs)
# --------------------------------------------------------------------------- #
def synth_tiles(B: int, size: int = 256, seed: int = 1234) -> torch.Tensor:
    """H&E-like NHWC uint8 tiles: per-channel normal around the 'Ron' stats."""
    g = torch.Generator().manual_seed(seed)
    mean = torch.tensor(MEAN_RON) * 255.0
    std = torch.tensor(STD_RON) * 255.0
    x = torch.randn(B, size, size, 3, generator=g) * std + mean
    return x.round().clamp(0, 255).to(torch.uint8)


def crop_windows(n_global: int = 2, n_local: int = 8, gsize: int = 224, lsize: int = 96):
    """Deterministic crop windows (y0, x0, size) -- SURVEY 8(d)."""
    wins = [(16 * g, 16 * g, gsize) for g in range(n_global)]
    wins += [(20 * l, 160 - 20 * l, lsize) for l in range(n_local)]
    return wins


def normalize_window(tiles_u8: torch.Tensor, win, mean=MEAN_RON, std=STD_RON, dtype=torch.float32):
    """ToTensor (/255) + Normalize (transformations.py:124-128) of one crop window -> NCHW."""
    y0, x0, s = win
    x = tiles_u8[:, y0:y0 + s, x0:x0 + s, :].to(dtype) / 255.0
    x = (x - torch.tensor(mean, dtype=dtype)) / torch.tensor(std, dtype=dtype)
    return x.permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------- #
# losses
# --------------------------------------------------------------------------- #
def softmax_lsce(logits, target, smoothing: float = 0.1):
    """train.py:1046 softmax, then train.py:1053 timm LabelSmoothingCrossEntropy
    with the gather index patched to ``target`` of shape [B,1]
    (train_instruct.txt:3-7).  Yes, that is a double softmax; it is the
    reference's behaviour."""
    prob = torch.softmax(logits, dim=1)
    logp = torch.log_softmax(prob, dim=-1)
    nll = -logp.gather(dim=-1, index=target).squeeze(1)
    smooth = -logp.mean(dim=-1)
    return ((1.0 - smoothing) * nll + smoothing * smooth).mean()


def dino_loss(student_out, teacher_out, center, n_crops: int, n_global: int = 2,
              student_temp: float = 0.1, teacher_temp: float = 0.04):
    """DINO paper Alg. 1 (absent from the reference).

    student_out [V*B, K] crop-major, teacher_out [G*B, K] crop-major,
    center [1, K].  Returns (loss, batch_center_sum) where the second value is
    ``teacher_out.sum(0, keepdim=True)`` (row D3, before the all-reduce)."""
    s = (student_out / student_temp).chunk(n_crops)
    t = torch.softmax((teacher_out - center) / teacher_temp, dim=-1).detach().chunk(n_global)
    total, n = 0.0, 0
    for iq, q in enumerate(t):
        for v in range(n_crops):
            if v == iq:
                continue
            total = total + torch.sum(-q * torch.log_softmax(s[v], dim=-1), dim=-1).mean()
            n += 1
    return total / n, teacher_out.sum(dim=0, keepdim=True).detach()


def update_center(center, batch_sum, n_rows: int, momentum: float = 0.9):
    """Row D3: c <- m c + (1-m) * sum / (rows * world)."""
    return center * momentum + (batch_sum / n_rows) * (1.0 - momentum)


def ema_update(teacher: Dict[str, torch.Tensor], student: Dict[str, torch.Tensor], m: float):
    """Row D4: theta_t <- m theta_t + (1-m) theta_s."""
    with torch.no_grad():
        for k in teacher:
            teacher[k].mul_(m).add_(student[k].detach(), alpha=1.0 - m)


# --------------------------------------------------------------------------- #
# multi-crop forward (row D1)
# --------------------------------------------------------------------------- #
def multicrop_forward(p, hp, crops: Sequence[torch.Tensor], arch: str, drop=None, dropout=None):
    """Group consecutive crops of equal resolution, run the backbone once per
    group on the concatenated batch, concat CLS features, head once.  ``drop``: [depth, 2, sum of crop batches]
    stochastic-depth factors in crop order (every crop of every image is its own sample), or None."""
    feats, i, img0, row0 = [], 0, 0, 0          # row0: first token row of the group in the token-concatenated row space (dropout indices)
    while i < len(crops):
        j = i
        while j < len(crops) and crops[j].shape[-1] == crops[i].shape[-1]:
            j += 1
        x = torch.cat(list(crops[i:j]))
        ntok = (x.shape[-1] // PATCH) ** 2 + 1
        feats.append(vit_features(p, x, arch, drop=None if drop is None else drop[:, :, img0:img0 + x.shape[0]],
                                  dropout=None if dropout is None else (dropout[0], dropout[1], row0)))
        img0 += x.shape[0]
        row0 += x.shape[0] * ntok
        i = j
    return dino_head(hp, torch.cat(feats))


# --------------------------------------------------------------------------- #
# AdamW (torch semantics) on flat dicts
# --------------------------------------------------------------------------- #
def no_weight_decay(name: str, t: torch.Tensor) -> bool:
    """timm create_optimizer_v2 filter: 1-D params, *.bias and
    pos_embed / cls_token get weight decay 0."""
    base = name.split(".", 1)[1] if name.startswith(("backbone.", "head.")) else name   # DinoOracle prefixes its two modules
    return t.ndim <= 1 or name.endswith(".bias") or base in ("pos_embed", "cls_token") \
        or name.endswith("weight_g")


class AdamW:
    def __init__(self, params: Dict[str, torch.Tensor], lr, wd, betas=(0.9, 0.999), eps=1e-8,
                 frozen: Sequence[str] = ()):
        self.p, self.lr, self.wd, self.b1, self.b2, self.eps = params, lr, wd, betas[0], betas[1], eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0
        self.frozen = set(frozen)

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor], lr: Optional[float] = None, wd: Optional[float] = None):
        lr = self.lr if lr is None else lr
        wd = self.wd if wd is None else wd
        self.t += 1
        bc1, bc2 = 1 - self.b1 ** self.t, 1 - self.b2 ** self.t
        for k, p in self.p.items():
            if k in self.frozen or grads.get(k) is None:
                continue
            g = grads[k]
            if not no_weight_decay(k, p):
                p.mul_(1 - lr * wd)
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-lr / bc1)


class Lamb:
    """timm.optim.Lamb restated (reference train.py:161 `--opt lamb` -> create_optimizer_v2; timm is not importable here, so this
    follows the published algorithm -- parity unpinned beyond this restatement): global-norm pre-clip to max_grad_norm, Adam
    moments with bias correction, update + wd * p, per-TENSOR trust ratio ||p|| / ||update|| where the tensor is decayed
    (always_adapt=False), p -= lr * ratio * update.  Defaults: betas (0.9, 0.999), eps 1e-6, max_grad_norm 1."""

    def __init__(self, params: Dict[str, torch.Tensor], lr, wd, betas=(0.9, 0.999), eps=1e-6, max_grad_norm=1.0):
        self.p, self.lr, self.wd, self.b1, self.b2, self.eps, self.mgn = params, lr, wd, betas[0], betas[1], eps, max_grad_norm
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor], lr: Optional[float] = None):
        lr = self.lr if lr is None else lr
        self.t += 1
        bc1, bc2 = 1 - self.b1 ** self.t, 1 - self.b2 ** self.t
        gn = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values() if g is not None))
        clip = max(gn / self.mgn, 1.0) if self.mgn else 1.0
        for k, p in self.p.items():
            if grads.get(k) is None:
                continue
            g = grads[k] / clip
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            upd = (self.m[k] / bc1) / ((self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps))
            wd = 0.0 if no_weight_decay(k, p) else self.wd
            if wd != 0:
                upd = upd + wd * p
                wn, un = float(p.norm()), float(upd.norm())
                upd = upd * (wn / un if wn > 0 and un > 0 else 1.0)
            p.add_(upd, alpha=-lr)


def adaptive_clip_grad(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], clip_factor: float, eps: float = 1e-3, skip=()):
    """timm.utils.adaptive_clip_grad restated (reference train.py:1072-1077 `--clip-mode agc`; timm is not importable here: follows
    the published algorithm, unpinned beyond this restatement): per unit (index of dim 0; the whole tensor for 1-D parameters)
    g <- g * max_norm / max(||g||, 1e-6) where ||g|| >= max_norm = clip_factor * max(||p||, eps).  Returns new gradients."""
    def unit_norm(x):
        return x.norm(2) if x.ndim <= 1 else x.norm(2, dim=tuple(range(1, x.ndim)), keepdim=True)
    out = {}
    for k, g in grads.items():
        if g is None or k in skip:
            out[k] = g
            continue
        max_norm = unit_norm(params[k]).clamp(min=eps) * clip_factor
        gn = unit_norm(g)
        out[k] = torch.where(gn < max_norm, g, g * (max_norm / gn.clamp(min=1e-6)))
    return out
