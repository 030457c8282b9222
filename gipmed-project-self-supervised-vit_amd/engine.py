"""Host-side sequencing of the hot path over the C ABI (no autograd, no torch math).

Mirrors, for this path only, what the reference reaches through timm's
``VisionTransformer`` + ``loss.backward()`` + ``optimizer.step()``
(reference train.py:1044-1078) and what the orphaned DINO ViT / DINOHead
bytecode specifies (``vit.pyc@L..``):

  * ``Arena``     -- every parameter of (backbone [+ head]) in ONE flat f32 buffer with
                     matching flat gradient / Adam-moment / bf16 / teacher buffers, so the
                     optimizer + EMA + bf16 refresh is one fused pass and the data-parallel
                     reduction is a few large RCCL calls over contiguous ranges.  Weight-decayed
                     matrices come first, in the order their gradients complete in backward.
  * ``VitGroup``  -- activations of one network pass: the crop groups are ``Segment``s of one
                     token-concatenated row space (multi-crop wrapper, row D1).
  * ``VitRunner`` -- forward / backward of the encoder as explicit launch sequences.
  * ``DinoEngine`` / ``SupervisedEngine`` -- one training step (rows S1, D1-D5, L1, O1).

The critical path is launched on the caller's current stream with static buffers; work that
feeds nothing downstream (teacher forward, weight-gradient GEMMs, LayerNorm-gradient finalize)
goes to one side stream ordered by events.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import ops

bf16, f32 = ops.bf16, torch.float32      # ops.bf16: the 16-bit dtype of the loaded library build (float16 under GIPVIT_ACT_FORMAT=f16)

ARCHS = {   # vit.pyc@L275-293
    "vit_tiny": dict(embed_dim=192, depth=12, num_heads=3),
    "vit_small": dict(embed_dim=384, depth=12, num_heads=6),
    "vit_base": dict(embed_dim=768, depth=12, num_heads=12),
}
MEAN_RON = (0.8998, 0.8253, 0.9357)   # transformations.py:106
STD_RON = (0.1125, 0.1751, 0.0787)    # transformations.py:113
PAD = 64                              # arena offsets are multiples of 64 elements


def _empty(shape, dt, device):
    """torch.empty, or NaN / 0xFF-poisoned memory when GIPVIT_POISON=1 (debug: any read of a
    buffer before it was written then shows up as NaN in the loss / gradients)."""
    import os
    t = torch.empty(shape, dtype=dt, device=device)
    if os.environ.get("GIPVIT_POISON"):
        if dt.is_floating_point:
            t.fill_(float("nan"))
        else:
            t.fill_(-1)
    return t


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


# --------------------------------------------------------------------------- #
# parameter inventory (timm / DINO state_dict names)
# --------------------------------------------------------------------------- #
def vit_param_specs(arch: str, img_size: int, num_classes: int = 0) -> "OrderedDict[str, Tuple[int, ...]]":
    a = ARCHS[arch]
    D, depth = a["embed_dim"], a["depth"]
    P = (img_size // 16) ** 2
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["cls_token"] = (1, 1, D)
    s["pos_embed"] = (1, P + 1, D)
    s["patch_embed.proj.weight"] = (D, 3, 16, 16)
    s["patch_embed.proj.bias"] = (D,)
    for i in range(depth):
        b = f"blocks.{i}."
        s[b + "norm1.weight"] = (D,); s[b + "norm1.bias"] = (D,)
        s[b + "attn.qkv.weight"] = (3 * D, D); s[b + "attn.qkv.bias"] = (3 * D,)
        s[b + "attn.proj.weight"] = (D, D); s[b + "attn.proj.bias"] = (D,)
        s[b + "norm2.weight"] = (D,); s[b + "norm2.bias"] = (D,)
        s[b + "mlp.fc1.weight"] = (4 * D, D); s[b + "mlp.fc1.bias"] = (4 * D,)
        s[b + "mlp.fc2.weight"] = (D, 4 * D); s[b + "mlp.fc2.bias"] = (D,)
    s["norm.weight"] = (D,); s["norm.bias"] = (D,)
    if num_classes > 0:
        s["head.weight"] = (num_classes, D); s["head.bias"] = (num_classes,)
    return s


def dino_head_specs(in_dim: int, out_dim: int, hidden: int = 2048, bottleneck: int = 256):
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()   # vit.pyc@L296-318
    s["mlp.0.weight"] = (hidden, in_dim); s["mlp.0.bias"] = (hidden,)
    s["mlp.2.weight"] = (hidden, hidden); s["mlp.2.bias"] = (hidden,)
    s["mlp.4.weight"] = (bottleneck, hidden); s["mlp.4.bias"] = (bottleneck,)
    s["last_layer.weight_g"] = (out_dim, 1)
    s["last_layer.weight_v"] = (out_dim, bottleneck)
    return s


def no_weight_decay(name: str, shape) -> bool:
    """timm create_optimizer_v2 filter: 1-D, *.bias, pos_embed, cls_token."""
    base = name.split(".", 1)[1] if name.startswith(("backbone.", "head.")) else name
    return len(shape) <= 1 or name.endswith(".bias") or base in ("pos_embed", "cls_token") or name.endswith("weight_g")


class Arena:
    """Flat parameter storage.  ``order`` lists names in arena order."""

    def __init__(self, specs: "OrderedDict[str, Tuple[int, ...]]", device, teacher: bool):
        decay = [n for n, s in specs.items() if not no_weight_decay(n, s)]
        nodecay = [n for n, s in specs.items() if no_weight_decay(n, s)]
        decay.reverse()                      # gradients complete head-first, block 11 .. 0, patch embed last
        self.specs, self.order = specs, decay + nodecay
        self.off: Dict[str, int] = {}
        o = 0
        for n in decay:
            self.off[n] = o
            o += _round_up(math.prod(specs[n]), PAD)
        self.n_decay = o
        for n in nodecay:
            self.off[n] = o
            o += _round_up(math.prod(specs[n]), PAD)
        self.n = o
        z = lambda dt: torch.zeros(self.n, dtype=dt, device=device)
        self.p, self.g, self.m, self.v, self.pb = z(f32), z(f32), z(f32), z(f32), z(bf16)
        self.t, self.tb = (z(f32), z(bf16)) if teacher else (None, None)

    def view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        shape = self.specs[name]
        o = self.off[name]
        return buf[o:o + math.prod(shape)].view(shape)

    def span(self, name: str) -> Tuple[int, int]:
        """[lo, hi) of one parameter in the flat buffers, padding included."""
        return self.off[name], self.off[name] + _round_up(math.prod(self.specs[name]), PAD)

    def load(self, state: Dict[str, torch.Tensor], buf: Optional[torch.Tensor] = None):
        buf = self.p if buf is None else buf
        for n in self.specs:
            self.view(buf, n).copy_(state[n].to(f32))

    def state_dict(self, buf=None, prefix: str = "") -> "OrderedDict[str, torch.Tensor]":
        buf = self.p if buf is None else buf
        return OrderedDict((n[len(prefix):], self.view(buf, n).detach().clone()) for n in self.specs if n.startswith(prefix))


class Weights:
    """Accessor over one weight set of an arena (student or teacher) with a name prefix."""

    def __init__(self, arena: Arena, prefix: str, teacher: bool = False, fp32: bool = False):
        self.a, self.prefix, self.teacher, self.fp32 = arena, prefix, teacher, fp32

    def w(self, name):      # matrix used as a GEMM operand: the bf16 copy (the f32 master itself in the fp32 operand mode)
        if self.fp32:
            return self.f(name)
        return self.a.view(self.a.tb if self.teacher else self.a.pb, self.prefix + name)

    def f(self, name):      # f32 parameter (bias, LN, pos, cls)
        return self.a.view(self.a.t if self.teacher else self.a.p, self.prefix + name)

    def g(self, name):      # f32 gradient
        return self.a.view(self.a.g, self.prefix + name)


# --------------------------------------------------------------------------- #
# bicubic pos-embed resampling as a fixed linear map (vit.pyc@L213-233)
# --------------------------------------------------------------------------- #
def pos_interp_matrix(n_src_side: int, crop: int) -> torch.Tensor:
    """[P_dst, P_src] matrix M with interpolate(pos) = M @ pos, built by pushing the
    identity basis through the exact torch call the reference makes."""
    import torch.nn.functional as F
    Ns = n_src_side * n_src_side
    w0 = crop // 16 + 0.1
    eye = torch.eye(Ns, dtype=torch.float64).reshape(1, n_src_side, n_src_side, Ns).permute(0, 3, 1, 2)
    out = F.interpolate(eye, scale_factor=(w0 / n_src_side, w0 / n_src_side), mode="bicubic")
    side = crop // 16
    assert out.shape[-1] == side and out.shape[-2] == side
    return out.permute(0, 2, 3, 1).reshape(side * side, Ns).to(torch.float32).contiguous()


# --------------------------------------------------------------------------- #
# activations of a crop group: one or more SEGMENTS (crop resolutions) whose tokens are
# concatenated -- every token-wise op (LayerNorm, all GEMMs, dW) then runs ONCE over all rows;
# only attention, patch embedding and the CLS handling run per segment (multi-crop wrapper, row D1)
# --------------------------------------------------------------------------- #
class Segment:
    def __init__(self, arch, n_img, crop, img_size, row0, img0, device, save, depth, H, act=bf16):
        D = ARCHS[arch]["embed_dim"]
        self.n_img, self.crop, self.row0, self.img0 = n_img, crop, row0, img0
        self.P = (crop // 16) ** 2
        self.N = self.P + 1
        self.T = n_img * self.N
        e = lambda shape, dt: _empty(shape, dt, device)
        nb = depth if save else 1
        self.patches = e((n_img * self.P, 768), act)
        self.lse = [e((n_img, H, self.N), f32) for _ in range(nb)]
        self.fstats = [e((n_img,), f32) for _ in range(2)]
        self.pos = None if crop == img_size else e((self.N, D), f32)
        self.interp = None if crop == img_size else pos_interp_matrix(img_size // 16, crop).to(device)
        if save:
            self.gpatch = e((n_img * self.P, D), act)
            self.dpos = e((self.N, D), f32)

    def rows(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.row0:self.row0 + self.T]


class VitGroup:
    def __init__(self, arch: str, segments, img_size: int, device, save: bool, act=bf16):
        """segments: [(n_img, crop), ...] in feature-row order.  ``act``: element type of the activation / gradient buffers
        that feed GEMMs and attention -- bf16 on the training path, f32 in the fp32 operand mode (csrc/f32path.hip)."""
        assert act in (bf16, f32)
        self.act = act
        a = ARCHS[arch]
        D, depth, H = a["embed_dim"], a["depth"], a["num_heads"]
        self.save = save
        self.segs: List[Segment] = []
        row0 = img0 = 0
        for n_img, crop in segments:
            sg = Segment(arch, n_img, crop, img_size, row0, img0, device, save, depth, H, act)
            self.segs.append(sg)
            row0 += sg.T
            img0 += n_img
        self.T, self.n_img = row0, img0
        T, nb = self.T, (depth if save else 1)
        e = lambda shape, dt: _empty(shape, dt, device)
        self.x = [e((T, D), f32) for _ in range(2 * depth + 1 if save else 3)]
        self.xn1 = [e((T, D), act) for _ in range(nb)]
        self.xn2 = [e((T, D), act) for _ in range(nb)]
        self.qkv = [e((T, 3 * D), act) for _ in range(nb)]
        self.o = [e((T, D), act) for _ in range(nb)]
        self.hp = [e((T, 4 * D), act) for _ in range(nb)]
        self.h = [e((T, 4 * D), act) for _ in range(nb)]
        self.stats = [[e((T,), f32) for _ in range(4)] for _ in range(nb)]   # mean1, rstd1, mean2, rstd2
        if save:   # backward scratch
            self.g, self.gb, self.gb2 = e((T, D), f32), e((T, D), act), e((T, D), act)
            self.dh = e((T, 4 * D), act)
            self.dxn = e((T, D), act)
            self.dqkv = e((T, 3 * D), act)
            self.do = e((T, D), act)
            # grouped weight gradients (one launch per block, queued on the side stream while the NEXT block runs): the dY
            # buffers alternate between two sets by block parity, so a block's set stays untouched until its group has run
            self.gb3, self.gb4 = e((T, D), act), e((T, D), act)
            self.dh_b, self.dqkv_b = e((T, 4 * D), act), e((T, 3 * D), act)

        self.depth, self.device = depth, device
        # CLS-only last block (VitRunner.cls_last): buffers over the n_img CLS rows, allocated on first use (alloc_cls)
        self.c_o = None
        self.D_ = D
        self.prepared = False     # VitRunner.prepare_backward has zeroed this group's backward scratch for the coming backward
        self.dropout, self.tmp = None, None   # --drop: (p, step seed) and the f32 branch-output buffer of the unfused residual adds (set_dropout)
        self.rs = None            # stochastic depth: f32 [depth, 2, T] row factors of the attention / MLP branch (set_drop)
        self.drop_img = None      # ... and the per-image factors they were expanded from
        self._row_img = None

    def set_drop(self, per_img: Optional[torch.Tensor]):
        """Stochastic depth (timm DropPath, vit.pyc@L66-74) for the next forward / backward of this group: ``per_img`` f32
        [depth, 2, n_img] on the device = keep / (1 - p) per image for each block's attention and MLP branch (images in
        segment order), or None to switch it off.  Expanded to one factor per token row in one launch."""
        self.drop_img = per_img
        if per_img is None:
            self.rs = None
            return
        assert per_img.dtype == f32 and tuple(per_img.shape) == (self.depth, 2, self.n_img), (per_img.shape, (self.depth, 2, self.n_img))
        if self._row_img is None:
            m = torch.cat([sg.img0 + torch.arange(sg.T, dtype=torch.int32) // sg.N for sg in self.segs])
            self._row_img = m.to(self.device)
            self._rs_buf = _empty((self.depth, 2, self.T), f32, self.device)
        ops.expand_rows(per_img.contiguous(), self._row_img, self._rs_buf, self.depth * 2, self.n_img, self.T)
        self.rs = self._rs_buf

    def set_dropout(self, p: float, seed: int):
        """Dropout (--drop, nn.Dropout after the pos-embed add, attn.proj, the MLP activation and mlp.fc2; vit.pyc@L98-131, L235-246) for
        the next forward / backward of this group: probability ``p`` and this step's 32-bit ``seed`` (the masks are counter-based:
        the backward regenerates them), or p = 0 to switch it off.  The step then runs the unfused Linear / LayerNorm kernels."""
        if not p:
            self.dropout = None
            return
        if not 0.0 < p < 1.0:
            raise ValueError(f"dropout probability {p}: need 0 <= p < 1")
        assert self.save, "dropout belongs to a training pass"
        if self.T * 4 * self.x[0].shape[1] >= 1 << 32:
            raise ValueError("dropout: a site has more than 2^32 elements")
        self.dropout = (float(p), int(seed) & 0xFFFFFFFF)
        if self.tmp is None:
            self.tmp = _empty(tuple(self.x[0].shape), f32, self.device)

    def alloc_cls(self):
        """Buffers of the last block's CLS-only tail: n_img rows instead of T (one per crop image, images in segment order)."""
        if self.c_o is not None:
            return
        n, D, act = self.n_img, self.D_, self.act
        e = lambda shape, dt: _empty(shape, dt, self.device)
        self.c_o, self.c_xn2 = e((n, D), act), e((n, D), act)
        self.c_xa, self.c_xb, self.c_xc = e((n, D), f32), e((n, D), f32), e((n, D), f32)
        self.c_h, self.c_hp = e((n, 4 * D), act), e((n, 4 * D), act)
        self.c_stats = [e((n,), f32), e((n,), f32)]                      # mean2, rstd2
        if self.save:
            self.c_g, self.c_gb, self.c_gb_att = e((n, D), f32), e((n, D), act), e((n, D), act)
            self.c_dh, self.c_dxn, self.c_do = e((n, 4 * D), act), e((n, D), act), e((n, D), act)

    def gather_cls(self, src: torch.Tensor, dst: torch.Tensor):
        """dst[image] = src[the image's CLS row] (strided copies, one per segment)."""
        for sg in self.segs:
            dst[sg.img0:sg.img0 + sg.n_img].copy_(sg.rows(src).view(sg.n_img, sg.N, -1)[:, 0])

    def scatter_cls(self, src: torch.Tensor, dst: torch.Tensor):
        """dst[the image's CLS row] = src[image]; the other rows of dst are left as they are."""
        for sg in self.segs:
            sg.rows(dst).view(sg.n_img, sg.N, -1)[:, 0].copy_(src[sg.img0:sg.img0 + sg.n_img])

    def xbuf(self, j):
        return self.x[j] if self.save else self.x[j % 3]

    def slot(self, i):
        return i if self.save else 0


class VitRunner:
    def __init__(self, arch: str, img_size: int, device, fp32: bool = False):
        a = ARCHS[arch]
        self.arch, self.D, self.depth, self.H, self.img_size = arch, a["embed_dim"], a["depth"], a["num_heads"], img_size
        self.fp32 = fp32    # fp32 operand mode: one f32 kernel per op (no fused Linear + LayerNorm, no grouped dW)
        self.scale = 64 ** -0.5
        # full-row Linear + LayerNorm kernels (csrc/panel.hip) exist for the ViT-S width; GIPVIT_FUSED_LN=0 keeps the
        # round-1 pair (128x128-tile GEMM + stand-alone LayerNorm pass) for A/B runs
        self.fused = self.D == 384 and not fp32 and os.environ.get("GIPVIT_FUSED_LN", "1") != "0"
        # the four weight-gradient products of a block as ONE split-K launch (gv_linear_dw_group): a quarter of the slab
        # traffic and four times longer k-loops than four launches (ViT-S: 165 us per block at 950 TFLOP/s against
        # 4 x (51 + 7) us).  GIPVIT_GROUP_DW=0 keeps one launch per product for A/B runs.
        self.group_dw = not fp32 and os.environ.get("GIPVIT_GROUP_DW", "1") != "0"
        # Only the CLS row of the last block's output feeds the head (vit.pyc@L248-253: forward returns x[:, 0]), so that block's
        # attention projection, MLP and residual adds for the other tokens -- and their whole backward except the K / V path --
        # are dead values: the last block runs them on the n_img CLS rows only (same loss and gradients; ~5.7 % fewer FLOPs of a
        # ViT-S DINO step).  GIPVIT_CLS_ONLY_LAST=0 computes every token of every block, as the reference's module + autograd do.
        self.cls_last = os.environ.get("GIPVIT_CLS_ONLY_LAST", "1") != "0"
        # forward-only passes (teacher, inference) run the MLP as one launch; GIPVIT_FUSED_MLP=0 keeps fc1 / fc2 apart (A/B runs)
        self.fused_mlp = os.environ.get("GIPVIT_FUSED_MLP", "1") != "0"
        # the CLS-only tail of the last block (n_img rows): fused Linear + LayerNorm kernels, or the tiled GEMM + a LayerNorm pass
        self.cls_tail_fused = os.environ.get("GIPVIT_CLS_TAIL_FUSED", "0") != "0"
        self.partials = _empty((L.LN_PARTIAL_BLOCKS, 3, self.D), f32, device)
        self.partials_ring = [self.partials] + [_empty((L.LN_PARTIAL_BLOCKS, 3, self.D), f32, device) for _ in range(2)]
        self.cs_ws = _empty((64 * 4 * self.D,), f32, device)
        self.one = torch.ones(1, dtype=f32, device=device)
        self.ws = _empty((L.lib.gv_linear_workspace_bytes() // 4,), f32, device)   # split-K slabs of the stream the weight gradients run on (the side stream when there is one)
        self.ws_main = _empty((16 << 20) // 4, f32, device)                              # ... of the few-row products of the CLS-only tail queued on the other one
        cuda = torch.device(device).type == "cuda"
        self.side = None
        if cuda and os.environ.get("GIPVIT_DW_STREAM", "1") != "0":
            prio = os.environ.get("GIPVIT_SIDE_PRIORITY")
            self.side = torch.cuda.Stream(device, priority=int(prio)) if prio is not None else torch.cuda.Stream(device)
        self._events: List[torch.cuda.Event] = []

    # ---- forward: tiles -> CLS features written into feats[row_off + seg.img0 ...]
    def forward(self, W: Weights, G: VitGroup, tiles_u8, windows, mean, std, feats, row_off: int = 0, fill=None):
        """windows: one list of (y0, x0) crop origins per segment; tiles_u8: one NHWC u8 tensor for all
        segments, or one per segment (pre-cut crops: each with the single window (0, 0)).  ``fill``: per-tile normalised
        fill boxes of the augmentation (gipvit.augment: Cutout after Normalize, MeanPixelRegularization), f32 [n_tiles, 8]."""
        D, T, H = self.D, G.T, self.H
        E = L
        pos_full = W.f("pos_embed").view(-1, D)
        x0 = G.xbuf(0)
        for k, (sg, wins) in enumerate(zip(G.segs, windows)):
            src = tiles_u8[k] if isinstance(tiles_u8, (list, tuple)) else tiles_u8
            ops.patchify(src, wins, sg.crop, mean, std, out=sg.patches, fill=fill)
            if sg.pos is None:
                pos = pos_full
            else:   # interpolate_pos_encoding: row 0 = cls pos, rows 1.. = M @ pos[1:]
                pos = sg.pos
                Ps = pos_full.shape[0] - 1
                ops.small_matmul(self.one, pos_full, pos, 1, D, 1, sam=0, sak=0, sbk=0, sbn=1)
                ops.small_matmul(sg.interp, pos_full[1:], pos[1:], sg.P, D, Ps, sam=Ps, sak=1, sbk=D, sbn=1)
            xs = sg.rows(x0)
            ops.cls_rows(xs, W.f("cls_token").view(-1), pos, sg.n_img, sg.N, D)
            ops.linear(sg.patches, W.w("patch_embed.proj.weight").view(D, 768), xs, sg.n_img * sg.P, D, 768,
                       epilogue=E.EPI_BIAS | E.EPI_POS, bias=W.f("patch_embed.proj.bias"), pos=pos, P=sg.P)
        # --drop: counter-based masks at the four nn.Dropout sites (gv_dropout / gv_dropout_add); the residual adds then run unfused
        dp = G.dropout
        dseed = (lambda layer, site: ops.dropout_site_seed(dp[1], layer, site)) if dp else None
        if dp:
            ops.dropout(x0, dseed(0, 0), dp[0], n=T * D)                                  # pos_drop
        # D = 384 (ViT-S): proj / fc2 run as full-row products with the residual add AND the LayerNorm that reads the new
        # row next fused into the epilogue (gv_linear_ln_fwd) -- only block 0's norm1 is a stand-alone pass
        fused = self.fused and dp is None
        for i in range(self.depth):
            b, s = f"blocks.{i}.", G.slot(i)
            xa, xb, xc = G.xbuf(2 * i), G.xbuf(2 * i + 1), G.xbuf(2 * i + 2)
            st = G.stats[s]
            rs_a, rs_m = (G.rs[i, 0], G.rs[i, 1]) if G.rs is not None else (None, None)      # stochastic depth of the two branches
            if not fused or i == 0:
                ops.layernorm_fwd(xa, W.f(b + "norm1.weight"), W.f(b + "norm1.bias"), T, D, y=G.xn1[s], mean=st[0], rstd=st[1])
            ops.linear(G.xn1[s], W.w(b + "attn.qkv.weight"), G.qkv[s], T, 3 * D, D, epilogue=E.EPI_BIAS, bias=W.f(b + "attn.qkv.bias"))
            # the last block under the CLS-only tail: only the CLS query's attention output is used (row 0 of every image)
            ql = 1 if (self._cls_tail(G) and i == self.depth - 1 and os.environ.get("GIPVIT_CLS_QLIMIT", "1") != "0") else 0
            if len(G.segs) > 1 and os.environ.get("GIPVIT_VARLEN_ATTN", "1") != "0":
                # the crop lengths of a multi-crop pass in ONE call (gv_attention_fwd_varlen: the 37-token pairs fill the 197-token
                # launch's half-empty last round); GIPVIT_VARLEN_ATTN=0 keeps one launch per segment for A/B runs
                ops.attention_fwd_varlen(G.qkv[s], G.o[s], [(sg.n_img, sg.N, sg.lse[s]) for sg in G.segs], H, self.scale, q_limit=ql)
            else:
                for sg in G.segs:
                    ops.attention_fwd(sg.rows(G.qkv[s]), sg.n_img, sg.N, H, self.scale, o=sg.rows(G.o[s]), lse=sg.lse[s], q_limit=ql)
            if self._cls_tail(G) and i == self.depth - 1:
                self._last_block_tail_fwd(W, G, i, xa, fused and self.cls_tail_fused)
                break
            if fused:
                ops.linear_ln_fwd(G.o[s], W.w(b + "attn.proj.weight"), xb, T, D, bias=W.f(b + "attn.proj.bias"), resid=xa,
                                  gamma=W.f(b + "norm2.weight"), beta=W.f(b + "norm2.bias"), y=G.xn2[s], mean=st[2], rstd=st[3], row_scale=rs_a)
            else:
                if dp:
                    ops.linear(G.o[s], W.w(b + "attn.proj.weight"), G.tmp, T, D, D, epilogue=E.EPI_BIAS, bias=W.f(b + "attn.proj.bias"))
                    ops.dropout_add(G.tmp, xa, xb, T, D, dseed(i, 1), dp[0], row_scale=rs_a)
                else:
                    ops.linear(G.o[s], W.w(b + "attn.proj.weight"), xb, T, D, D, epilogue=E.EPI_BIAS | E.EPI_RESID,
                               bias=W.f(b + "attn.proj.bias"), resid=xa, row_scale=rs_a)
                ops.layernorm_fwd(xb, W.f(b + "norm2.weight"), W.f(b + "norm2.bias"), T, D, y=G.xn2[s], mean=st[2], rstd=st[3])
            if fused and self.fused_mlp and not G.save:
                # a pass that keeps no activations (the teacher, inference): fc1 -> GELU -> fc2 -> + residual -> the next norm1 in ONE
                # launch (gv_mlp_ln_fwd); the [T, 4 D] activation never reaches HBM
                nxt = i + 1 < self.depth
                nb, ns = f"blocks.{i + 1}.", G.slot(i + 1)
                ops.mlp_ln_fwd(G.xn2[s], W.w(b + "mlp.fc1.weight"), W.f(b + "mlp.fc1.bias"), W.w(b + "mlp.fc2.weight"), xc, T, D, 4 * D,
                               bias2=W.f(b + "mlp.fc2.bias"), resid=xb, gamma=W.f(nb + "norm1.weight") if nxt else None,
                               beta=W.f(nb + "norm1.bias") if nxt else None, y=G.xn1[ns] if nxt else None,
                               mean=G.stats[ns][0] if nxt else None, rstd=G.stats[ns][1] if nxt else None, row_scale=rs_m)
                continue
            ops.linear(G.xn2[s], W.w(b + "mlp.fc1.weight"), G.h[s], T, 4 * D, D,
                       epilogue=E.EPI_BIAS | E.EPI_GELU | (E.EPI_SAVE_PRE if G.save else 0),   # a forward-only group keeps no pre-activation
                       bias=W.f(b + "mlp.fc1.bias"), aux_out=G.hp[s] if G.save else None)
            if dp:
                ops.dropout(G.h[s], dseed(i, 2), dp[0], n=T * 4 * D)
            if fused:
                nxt = i + 1 < self.depth
                nb, ns = f"blocks.{i + 1}.", G.slot(i + 1)
                ops.linear_ln_fwd(G.h[s], W.w(b + "mlp.fc2.weight"), xc, T, 4 * D, bias=W.f(b + "mlp.fc2.bias"), resid=xb,
                                  gamma=W.f(nb + "norm1.weight") if nxt else None, beta=W.f(nb + "norm1.bias") if nxt else None,
                                  y=G.xn1[ns] if nxt else None, mean=G.stats[ns][0] if nxt else None, rstd=G.stats[ns][1] if nxt else None,
                                  row_scale=rs_m)
            elif dp:
                ops.linear(G.h[s], W.w(b + "mlp.fc2.weight"), G.tmp, T, D, 4 * D, epilogue=E.EPI_BIAS, bias=W.f(b + "mlp.fc2.bias"))
                ops.dropout_add(G.tmp, xb, xc, T, D, dseed(i, 3), dp[0], row_scale=rs_m)
            else:
                ops.linear(G.h[s], W.w(b + "mlp.fc2.weight"), xc, T, D, 4 * D, epilogue=E.EPI_BIAS | E.EPI_RESID,
                           bias=W.f(b + "mlp.fc2.bias"), resid=xb, row_scale=rs_m)
        xl = G.xbuf(2 * self.depth)
        for sg in G.segs:
            r0 = row_off + sg.img0
            if self._cls_tail(G):         # the last block left its output for the CLS rows only, contiguous
                ops.layernorm_fwd(G.c_xc[sg.img0:sg.img0 + sg.n_img], W.f("norm.weight"), W.f("norm.bias"), sg.n_img, D,
                                  y=feats[r0:r0 + sg.n_img], mean=sg.fstats[0], rstd=sg.fstats[1])
            else:
                ops.layernorm_fwd(sg.rows(xl), W.f("norm.weight"), W.f("norm.bias"), sg.n_img, D, x_stride=sg.N * D,
                                  y=feats[r0:r0 + sg.n_img], mean=sg.fstats[0], rstd=sg.fstats[1])

    def _cls_tail(self, G: VitGroup) -> bool:
        """Does this group run the last block's projection / MLP on the CLS rows only?  (Training groups: on the grouped
        weight-gradient path and without --drop, whose backward this build restates for that case.)"""
        return self.cls_last and G.dropout is None and (not G.save or self.group_dw)

    def _last_block_tail_fwd(self, W: Weights, G: VitGroup, i: int, xa: torch.Tensor, fused: bool):
        """x + proj(attention) -> norm2 -> MLP -> residual of block i for the CLS rows only (vit.pyc@L146-152 on the rows that
        VisionTransformer.forward returns, L248-253): n_img rows through the same kernels the full blocks use."""
        G.alloc_cls()
        D, n, s, E, b = self.D, G.n_img, G.slot(i), L, f"blocks.{i}."
        rs_a, rs_m = (G.drop_img[i, 0], G.drop_img[i, 1]) if G.rs is not None else (None, None)       # per image = per CLS row
        G.gather_cls(G.o[s], G.c_o)
        G.gather_cls(xa, G.c_xa)
        if fused:
            ops.linear_ln_fwd(G.c_o, W.w(b + "attn.proj.weight"), G.c_xb, n, D, bias=W.f(b + "attn.proj.bias"), resid=G.c_xa,
                              gamma=W.f(b + "norm2.weight"), beta=W.f(b + "norm2.bias"), y=G.c_xn2, mean=G.c_stats[0], rstd=G.c_stats[1], row_scale=rs_a)
        else:
            ops.linear(G.c_o, W.w(b + "attn.proj.weight"), G.c_xb, n, D, D, epilogue=E.EPI_BIAS | E.EPI_RESID,
                       bias=W.f(b + "attn.proj.bias"), resid=G.c_xa, row_scale=rs_a)
            ops.layernorm_fwd(G.c_xb, W.f(b + "norm2.weight"), W.f(b + "norm2.bias"), n, D, y=G.c_xn2, mean=G.c_stats[0], rstd=G.c_stats[1])
        ops.linear(G.c_xn2, W.w(b + "mlp.fc1.weight"), G.c_h, n, 4 * D, D, epilogue=E.EPI_BIAS | E.EPI_GELU | (E.EPI_SAVE_PRE if G.save else 0),
                   bias=W.f(b + "mlp.fc1.bias"), aux_out=G.c_hp if G.save else None)
        if fused:
            ops.linear_ln_fwd(G.c_h, W.w(b + "mlp.fc2.weight"), G.c_xc, n, 4 * D, bias=W.f(b + "mlp.fc2.bias"), resid=G.c_xb, row_scale=rs_m)
        else:
            # (a few hundred rows, a 4 D-deep reduction: gv_linear splits K when handed a scratch -- the one of the stream this pass is queued on)
            on_side = self.side is not None and xa.is_cuda and torch.cuda.current_stream() == self.side
            ops.linear(G.c_h, W.w(b + "mlp.fc2.weight"), G.c_xc, n, D, 4 * D, epilogue=E.EPI_BIAS | E.EPI_RESID,
                       bias=W.f(b + "mlp.fc2.bias"), resid=G.c_xb, row_scale=rs_m, workspace=self.ws if on_side else self.ws_main)

    def prepare_backward(self, G: VitGroup):
        """Zero the backward scratch that is accumulated into or only partly written: the residual gradient (the final norm's backward
        writes the CLS rows), and the attention-output gradient of the CLS-only last block / the first dY buffer otherwise.  Stream-
        ordered fills with no input: an engine may issue them early on its side stream (DinoEngine does, beside the forward pass)."""
        G.g.zero_()
        if self._cls_tail(G):
            G.alloc_cls()
            G.do.zero_()
        else:
            (G.gb3 if (self.group_dw and ((self.depth - 1) & 1)) else G.gb).zero_()
        G.prepared = True

    def _fin3(self, dgamma, dbeta, dbias):
        ops.ln_finalize(self.partials, L.LN_PARTIAL_BLOCKS, self.D, dgamma, dbeta, dbias)

    # ---- backward from d(CLS features) bf16 [G.n_img, D]; gradients ACCUMULATE into the arena
    def backward(self, W: Weights, G: VitGroup, dfeat: torch.Tensor, on_block_done=None):
        """``on_block_done(i)`` is called once block i's parameter gradients are complete
        (data-parallel engines start that block's all-reduce there)."""
        D, T, H = self.D, G.T, self.H
        E = L
        ACC = E.EPI_ACCUM
        grouped = self.group_dw
        # --drop: the gradient entering a dropout site carries that site's mask (regenerated from the step seed); the bias gradients
        # of attn.proj / mlp.fc2 are then column sums of the MASKED gradient, not LayerNorm backward's third sum
        dp = G.dropout
        dseed = (lambda layer, site: ops.dropout_site_seed(dp[1], layer, site)) if dp else None
        fused_b = self.fused and dp is None

        def drop_branch_grad(gb, layer, site, bias_grad):
            ops.dropout(gb, dseed(layer, site), dp[0], n=T * D)
            ops.colsum(gb, T, D, self.cs_ws, bias_grad, accumulate=True)
        sets = ((G.gb, G.gb2, G.dh, G.dqkv), (G.gb3, G.gb4, G.dh_b, G.dqkv_b))      # per block parity: dY of the MLP / attention half, dh, dqkv
        gb_first = sets[(self.depth - 1) & 1][0] if grouped else G.gb
        cls_tail = self._cls_tail(G)          # the last block ran its projection / MLP on the CLS rows only (forward): so does its backward
        if not G.prepared:
            self.prepare_backward(G)
        G.prepared = False
        xl = G.x[2 * self.depth]
        # stochastic depth: the bf16 gradient handed to a branch carries that branch's row factor (rs[i, 0] attention, rs[i, 1] MLP)
        rs = G.rs
        for sg in G.segs:
            # (the final norm touches the CLS rows only, one per image: the per-image factors are its row factors)
            im = slice(sg.img0, sg.img0 + sg.n_img)
            if cls_tail:
                ops.layernorm_bwd(dfeat[im], G.c_xc[im], sg.fstats[0], sg.fstats[1], W.f("norm.weight"), G.c_g[im], G.c_gb[im], self.partials,
                                  sg.n_img, D, g_init=True, gb_scale=None if rs is None else G.drop_img[self.depth - 1, 1, im])
            else:
                ops.layernorm_bwd(dfeat[im], sg.rows(xl), sg.fstats[0], sg.fstats[1], W.f("norm.weight"), sg.rows(G.g),
                                  sg.rows(gb_first), self.partials, sg.n_img, D, x_stride=sg.N * D, g_stride=sg.N * D, gb_stride=sg.N * D, g_init=True,
                                  gb_scale=None if rs is None else G.drop_img[self.depth - 1, 1, im])
            self._fin3(W.g("norm.weight"), W.g("norm.bias"), None if dp else W.g(f"blocks.{self.depth - 1}.mlp.fc2.bias"))
        # The weight-gradient GEMMs are off the critical path (nothing in backward consumes dW):
        # they run on a side stream beside the dX chain, so their tiles fill the tail of every
        # main-stream kernel (a 345 x 3-tile GEMM occupies 2.02 rounds of the 512 workgroup slots).
        # fork(): side waits for what main has enqueued; join(ev): main waits for a side event
        # before it overwrites a buffer a dW product still reads (gb / dh / dqkv).
        main = torch.cuda.current_stream() if xl.is_cuda else None
        side = self.side if (main is not None and self.side is not None) else None
        evs = self._events
        n_ev = [0]

        def new_event():
            if n_ev[0] == len(evs):
                evs.append(torch.cuda.Event())
            n_ev[0] += 1
            return evs[n_ev[0] - 1]

        def dw(A, Bm, Cg, M, N, colsum_a=None):
            """dW (+)= A^T Bm on the side stream once main's work so far is done; returns the event that marks its end."""
            if side is None:
                ops.linear(A, Bm, Cg, M, N, T, trans_a=True, trans_b=True, epilogue=ACC, colsum_a=colsum_a, workspace=self.ws)
                return None
            e0 = new_event(); e0.record(main); side.wait_event(e0)
            with torch.cuda.stream(side):
                ops.linear(A, Bm, Cg, M, N, T, trans_a=True, trans_b=True, epilogue=ACC, colsum_a=colsum_a, workspace=self.ws)
                e1 = new_event(); e1.record(side)
            return e1

        def join(ev):
            if ev is not None:
                main.wait_event(ev)

        # LayerNorm backward leaves per-block column sums in a partials buffer; folding them into the gamma / beta /
        # bias gradients (ln_finalize) is parameter-gradient work too, so it also goes to the side stream.  Three
        # partials buffers rotate; one is rewritten only after the finalize that read it (three calls ago) has run.
        ring, fin_ev, ring_i, last_side = self.partials_ring, [None, None, None], [0], [None]

        def ln_bwd(dy, x, mean, rstd, gamma, gb, d0, d1, d2, dx_of=None, gb_scale=None, rows=None, g=None):
            """LayerNorm backward into the residual gradient.  ``dx_of = (dY, W, K)``: the dX product that produces ``dy``
            runs fused with it (gv_linear_ln_bwd) and ``dy`` never exists in HBM.  ``rows`` / ``g``: a row count and residual
            gradient other than the group's T rows / G.g (the last block's CLS-only tail)."""
            k = ring_i[0]
            ring_i[0] = (k + 1) % 3
            join(fin_ev[k])
            M_, g_ = (T if rows is None else rows), (G.g if g is None else g)
            if dx_of is not None and fused_b and (rows is None or self.cls_tail_fused):
                nblk = ops.linear_ln_bwd(dx_of[0], dx_of[1], x, mean, rstd, gamma, g_, gb, ring[k], M_, dx_of[2], gb_scale=gb_scale)
            else:
                if dx_of is not None:
                    ops.linear(dx_of[0], dx_of[1], dy, M_, D, dx_of[2], trans_b=True, workspace=self.ws_main if rows is not None else None)
                ops.layernorm_bwd(dy, x, mean, rstd, gamma, g_, gb, ring[k], M_, D, gb_scale=gb_scale)
                nblk = L.LN_PARTIAL_BLOCKS
            if side is None:
                ops.ln_finalize(ring[k], nblk, D, d0, d1, d2)
                return
            e0 = new_event(); e0.record(main); side.wait_event(e0)
            with torch.cuda.stream(side):
                ops.ln_finalize(ring[k], nblk, D, d0, d1, d2)
                e1 = new_event(); e1.record(side)
            fin_ev[k] = last_side[0] = e1

        def dw_group(problems):
            """all weight gradients of one block in one launch on the side stream; returns the event that marks its end."""
            if side is None:
                ops.linear_dw_group(problems, T, self.ws)
                return None
            e0 = new_event(); e0.record(main); side.wait_event(e0)
            with torch.cuda.stream(side):
                ops.linear_dw_group(problems, T, self.ws)
                e1 = new_event(); e1.record(side)
            return e1

        def on_side(fn):
            """``fn()`` on the side stream once main's work so far is done; returns the event that marks its end."""
            if side is None:
                fn()
                return None
            e0 = new_event(); e0.record(main); side.wait_event(e0)
            with torch.cuda.stream(side):
                fn()
                e1 = new_event(); e1.record(side)
            return e1

        def report(i):
            if on_block_done is not None:
                # the block's weight gradients are produced by the side stream: report the block from
                # there, so a data-parallel all-reduce queues behind the dW products and main never waits
                if side is None:
                    on_block_done(i)
                else:
                    with torch.cuda.stream(side):
                        on_block_done(i)

        done_grp = [None, None]
        for i in reversed(range(self.depth) if grouped else ()):
            b, st = f"blocks.{i}.", G.stats[i]
            par = i & 1
            gb_mlp, gb_att, dh, dqkv = sets[par]
            gb_next = sets[par ^ 1][0]                    # dY of block i - 1's MLP half
            join(done_grp[par])                           # block i + 2's group read this parity's buffers
            if cls_tail and i == self.depth - 1:
                # ---- last block, CLS rows only (forward: _last_block_tail_fwd): MLP and projection backward over n_img rows; the
                # attention backward and the qkv product's dX + norm1 backward then run over all tokens as in every block, with
                # dO and the residual gradient zero outside the CLS rows (only K and V of the other tokens reached the output)
                n = G.n_img
                ops.linear(G.c_gb, W.w(b + "mlp.fc2.weight"), G.c_dh, n, 4 * D, D, trans_b=True, epilogue=E.EPI_DGELU, aux_in=G.c_hp)
                ln_bwd(G.c_dxn, G.c_xb, G.c_stats[0], G.c_stats[1], W.f(b + "norm2.weight"), G.c_gb_att,
                       W.g(b + "norm2.weight"), W.g(b + "norm2.bias"), W.g(b + "attn.proj.bias"),
                       dx_of=(G.c_dh, W.w(b + "mlp.fc1.weight"), 4 * D), gb_scale=None if rs is None else G.drop_img[i, 0], rows=n, g=G.c_g)
                ops.linear(G.c_gb_att, W.w(b + "attn.proj.weight"), G.c_do, n, D, D, trans_b=True)
                G.scatter_cls(G.c_do, G.do)           # (G.do was zeroed by prepare_backward)
                G.scatter_cls(G.c_g, G.g)                 # (G.g was zeroed above)
                # dO is zero behind every image's CLS row: the attention backward skips the other queries (their dQ rows come out zero)
                ops.attention_bwd_varlen(G.qkv[i], G.o[i], G.do, dqkv, [(sg.n_img, sg.N, sg.lse[i]) for sg in G.segs], H, self.scale,
                                         q_limit=1 if os.environ.get("GIPVIT_CLS_QLIMIT", "1") != "0" else 0)
                ln_bwd(G.dxn, G.x[2 * i], st[0], st[1], W.f(b + "norm1.weight"), gb_next,
                       W.g(b + "norm1.weight"), W.g(b + "norm1.bias"), W.g(f"blocks.{i - 1}.mlp.fc2.bias") if i > 0 else None,
                       dx_of=(dqkv, W.w(b + "attn.qkv.weight"), 3 * D), gb_scale=None if (rs is None or i == 0) else rs[i - 1, 1])

                def last_block_dw():
                    # three weight gradients reduce over the n CLS rows, the qkv one over all tokens
                    ops.linear(G.c_gb, G.c_h, W.g(b + "mlp.fc2.weight"), D, 4 * D, n, trans_a=True, trans_b=True, epilogue=ACC, workspace=self.ws)
                    ops.linear(G.c_dh, G.c_xn2, W.g(b + "mlp.fc1.weight"), 4 * D, D, n, trans_a=True, trans_b=True, epilogue=ACC,
                               colsum_a=W.g(b + "mlp.fc1.bias"), workspace=self.ws)
                    ops.linear(G.c_gb_att, G.c_o, W.g(b + "attn.proj.weight"), D, D, n, trans_a=True, trans_b=True, epilogue=ACC, workspace=self.ws)
                    ops.linear_dw_group([(dqkv, G.xn1[i], W.g(b + "attn.qkv.weight"), W.g(b + "attn.qkv.bias"))], T, self.ws)
                done_grp[par] = on_side(last_block_dw)
                report(i)
                continue
            if dp:
                drop_branch_grad(gb_mlp, i, 3, W.g(b + "mlp.fc2.bias"))
            ops.linear(gb_mlp, W.w(b + "mlp.fc2.weight"), dh, T, 4 * D, D, trans_b=True, epilogue=E.EPI_DGELU, aux_in=G.hp[i])
            if dp:
                ops.dropout(dh, dseed(i, 2), dp[0], n=T * 4 * D)
            ln_bwd(G.dxn, G.x[2 * i + 1], st[2], st[3], W.f(b + "norm2.weight"), gb_att,
                   W.g(b + "norm2.weight"), W.g(b + "norm2.bias"), None if dp else W.g(b + "attn.proj.bias"),
                   dx_of=(dh, W.w(b + "mlp.fc1.weight"), 4 * D), gb_scale=None if rs is None else rs[i, 0])
            if dp:
                drop_branch_grad(gb_att, i, 1, W.g(b + "attn.proj.bias"))
            ops.linear(gb_att, W.w(b + "attn.proj.weight"), G.do, T, D, D, trans_b=True)
            ops.attention_bwd_varlen(G.qkv[i], G.o[i], G.do, dqkv, [(sg.n_img, sg.N, sg.lse[i]) for sg in G.segs], H, self.scale)
            # the block's four dY operands are final once the attention backward has run: its weight gradients go to the side stream
            # BEFORE the last kernel of the block's dX chain (which only reads dqkv), one kernel earlier than the chain's end
            probs = [(gb_mlp, G.h[i], W.g(b + "mlp.fc2.weight"), None),
                     (dh, G.xn2[i], W.g(b + "mlp.fc1.weight"), W.g(b + "mlp.fc1.bias")),
                     (gb_att, G.o[i], W.g(b + "attn.proj.weight"), None),
                     (dqkv, G.xn1[i], W.g(b + "attn.qkv.weight"), W.g(b + "attn.qkv.bias"))]
            done_grp[par] = dw_group(probs)
            join(done_grp[par ^ 1])                       # block i + 1's group read gb_next (its MLP-half dY)
            ln_bwd(G.dxn, G.x[2 * i], st[0], st[1], W.f(b + "norm1.weight"), gb_next,
                   W.g(b + "norm1.weight"), W.g(b + "norm1.bias"), W.g(f"blocks.{i - 1}.mlp.fc2.bias") if (i > 0 and not dp) else None,
                   dx_of=(dqkv, W.w(b + "attn.qkv.weight"), 3 * D), gb_scale=None if (rs is None or i == 0) else rs[i - 1, 1])
            report(i)
        join(done_grp[0]); join(done_grp[1])

        gbs = (G.gb, G.gb2)              # gb: dY of the MLP half, gb2: dY of the attention half
        done_fc1 = done_qkv = done_fc2 = done_proj = None
        for i in reversed(range(self.depth) if not grouped else ()):
            b, st = f"blocks.{i}.", G.stats[i]
            # MLP
            join(done_fc1)               # last block's dW_fc1 read dh
            if dp:
                drop_branch_grad(gbs[0], i, 3, W.g(b + "mlp.fc2.bias"))
            ops.linear(gbs[0], W.w(b + "mlp.fc2.weight"), G.dh, T, 4 * D, D, trans_b=True, epilogue=E.EPI_DGELU, aux_in=G.hp[i])
            if dp:
                ops.dropout(G.dh, dseed(i, 2), dp[0], n=T * 4 * D)
            done_fc2 = dw(gbs[0], G.h[i], W.g(b + "mlp.fc2.weight"), D, 4 * D)
            done_fc1 = dw(G.dh, G.xn2[i], W.g(b + "mlp.fc1.weight"), 4 * D, D, colsum_a=W.g(b + "mlp.fc1.bias"))
            join(done_proj)              # last block's dW_proj read gb2
            ln_bwd(G.dxn, G.x[2 * i + 1], st[2], st[3], W.f(b + "norm2.weight"), gbs[1],
                   W.g(b + "norm2.weight"), W.g(b + "norm2.bias"), None if dp else W.g(b + "attn.proj.bias"),
                   dx_of=(G.dh, W.w(b + "mlp.fc1.weight"), 4 * D), gb_scale=None if rs is None else rs[i, 0])
            if dp:
                drop_branch_grad(gbs[1], i, 1, W.g(b + "attn.proj.bias"))
            # attention
            ops.linear(gbs[1], W.w(b + "attn.proj.weight"), G.do, T, D, D, trans_b=True)
            done_proj = dw(gbs[1], G.o[i], W.g(b + "attn.proj.weight"), D, D)
            join(done_qkv)               # last block's dW_qkv read dqkv
            for sg in G.segs:
                ops.attention_bwd(sg.rows(G.qkv[i]), sg.rows(G.o[i]), sg.rows(G.do), sg.lse[i], sg.n_img, sg.N, H, self.scale, dqkv=sg.rows(G.dqkv))
            done_qkv = dw(G.dqkv, G.xn1[i], W.g(b + "attn.qkv.weight"), 3 * D, D, colsum_a=W.g(b + "attn.qkv.bias"))
            join(done_fc2)               # this block's dW_fc2 read gb
            ln_bwd(G.dxn, G.x[2 * i], st[0], st[1], W.f(b + "norm1.weight"), gbs[0],
                   W.g(b + "norm1.weight"), W.g(b + "norm1.bias"), W.g(f"blocks.{i - 1}.mlp.fc2.bias") if (i > 0 and not dp) else None,
                   dx_of=(G.dqkv, W.w(b + "attn.qkv.weight"), 3 * D), gb_scale=None if (rs is None or i == 0) else rs[i - 1, 1])
            report(i)
        join(done_qkv)          # every dW product is in (the side stream runs them in order); ws is free again
        join(last_side[0])      # ... and the last finalize
        if dp:
            ops.dropout(G.g, dseed(0, 0), dp[0], n=T * D)                                 # pos_drop's mask on the token gradient
        # token assembly + patch embedding, per segment
        gpos = W.g("pos_embed").view(-1, D)
        for sg in G.segs:
            ops.tokens_bwd(sg.rows(G.g), sg.gpatch, sg.dpos, None, sg.n_img, sg.N, D, accumulate=False)
            # d cls_token = sum over images of the CLS-row gradient = dpos row 0
            ops.small_matmul(self.one, sg.dpos, W.g("cls_token").view(1, D), 1, D, 1, sam=0, sak=0, sbk=0, sbn=1, accumulate=True)
            ops.linear(sg.gpatch, sg.patches, W.g("patch_embed.proj.weight").view(D, 768), D, 768, sg.n_img * sg.P,
                       trans_a=True, trans_b=True, epilogue=ACC, workspace=self.ws)
            ops.colsum(sg.dpos[1:], sg.P, D, self.cs_ws, W.g("patch_embed.proj.bias"), accumulate=True)
            if sg.pos is None:
                ops.small_matmul(self.one, sg.dpos, gpos, 1, sg.N * D, 1, sam=0, sak=0, sbk=0, sbn=1, accumulate=True)
            else:
                Ps = gpos.shape[0] - 1
                ops.small_matmul(self.one, sg.dpos, gpos, 1, D, 1, sam=0, sak=0, sbk=0, sbn=1, accumulate=True)
                ops.small_matmul(sg.interp, sg.dpos[1:], gpos[1:], Ps, D, sg.P, sam=1, sak=Ps, sbk=D, sbn=1, accumulate=True)


# --------------------------------------------------------------------------- #
# DINO head (vit.pyc@L296-330) forward / backward over R rows
# --------------------------------------------------------------------------- #
class HeadBuffers:
    def __init__(self, R: int, D: int, K: int, hidden: int, bott: int, device, save: bool, act=bf16):
        e = lambda shape, dt: _empty(shape, dt, device)
        self.R = R
        self.feats = e((R, D), act)
        self.h1p, self.h1 = e((R, hidden), act), e((R, hidden), act)
        self.h2p, self.h2 = e((R, hidden), act), e((R, hidden), act)
        self.z, self.zn, self.inv = e((R, bott), f32), e((R, bott), act), e((R,), f32)
        self.logits = e((R, K), f32)
        if save:
            self.dlogits = e((R, K), act)
            self.dzn, self.dz = e((R, bott), f32), e((R, bott), act)
            self.dh2, self.dh1 = e((R, hidden), act), e((R, hidden), act)
            self.dfeats = e((R, D), act)
            self.dwn = e((K, bott), f32)


class HeadRunner:
    def __init__(self, D: int, K: int, hidden: int, bott: int, device):
        self.D, self.K, self.hidden, self.bott = D, K, hidden, bott
        self.cs_ws = _empty((64 * max(hidden, bott),), f32, device)
        self.ws = _empty((L.lib.gv_linear_workspace_bytes() // 4,), f32, device)   # split-K slabs
        self.ws_side = None                                                           # second one, for products queued on a side stream

    def forward(self, W: Weights, wn: torch.Tensor, hb: HeadBuffers, on_side: bool = False):
        """``on_side``: the call is queued on the engine's side stream (the teacher's head beside the student's forward): its
        split-K products then take the side stream's scratch.  (A few hundred rows against 2048-deep reductions: gv_linear
        splits K when it is handed a scratch buffer.)"""
        R, D, Hd, Bt, K, E = hb.R, self.D, self.hidden, self.bott, self.K, L
        if on_side and self.ws_side is None:
            self.ws_side = _empty((L.lib.gv_linear_workspace_bytes() // 4,), f32, hb.feats.device)
        ws = self.ws_side if on_side else self.ws
        ops.linear(hb.feats, W.w("mlp.0.weight"), hb.h1, R, Hd, D, epilogue=E.EPI_BIAS | E.EPI_GELU | E.EPI_SAVE_PRE,
                   bias=W.f("mlp.0.bias"), aux_out=hb.h1p)
        ops.linear(hb.h1, W.w("mlp.2.weight"), hb.h2, R, Hd, Hd, epilogue=E.EPI_BIAS | E.EPI_GELU | E.EPI_SAVE_PRE,
                   bias=W.f("mlp.2.bias"), aux_out=hb.h2p, workspace=ws)
        ops.linear(hb.h2, W.w("mlp.4.weight"), hb.z, R, Bt, Hd, epilogue=E.EPI_BIAS, bias=W.f("mlp.4.bias"), workspace=ws)
        ops.l2norm_fwd(hb.z, hb.zn, hb.inv, R, Bt)
        ops.linear(hb.zn, wn, hb.logits, R, K, Bt)

    def backward(self, W: Weights, wn: torch.Tensor, hb: HeadBuffers, train_last_layer: bool = True, side=None):
        """``side``: a stream for the weight-gradient products (they feed nothing in backward); each is queued
        there behind an event that marks its operands ready, with a split-K workspace of its own."""
        R, D, Hd, Bt, K, E = hb.R, self.D, self.hidden, self.bott, self.K, L
        ACC = E.EPI_ACCUM
        if side is not None and self.ws_side is None:
            self.ws_side = _empty((L.lib.gv_linear_workspace_bytes() // 4,), f32, hb.feats.device)
        main = torch.cuda.current_stream() if side is not None else None

        def on_side(fn):
            if side is None:
                return fn(self.ws)
            ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
            with torch.cuda.stream(side):
                fn(self.ws_side)

        def dw_last(ws):   # dW_n = dlogits^T zn, then through weight_norm (g frozen: norm_last_layer)
            ops.linear(hb.dlogits, hb.zn, hb.dwn, K, Bt, R, trans_a=True, trans_b=True)
            ops.weightnorm_bwd(hb.dwn, W.f("last_layer.weight_v"), W.f("last_layer.weight_g").view(-1),
                               W.g("last_layer.weight_v"), None, K, Bt, accumulate=True)
        if train_last_layer:
            on_side(dw_last)
        hb.dzn.zero_()         # split-K over the K=65536 classes accumulates with atomics
        ops.linear(hb.dlogits, wn, hb.dzn, R, Bt, K, trans_b=True, epilogue=ACC, workspace=self.ws)
        ops.l2norm_bwd(hb.dzn, hb.zn, hb.inv, hb.dz, R, Bt)
        on_side(lambda ws: ops.linear(hb.dz, hb.h2, W.g("mlp.4.weight"), Bt, Hd, R, trans_a=True, trans_b=True, epilogue=ACC,
                                      colsum_a=W.g("mlp.4.bias"), workspace=ws))
        ops.linear(hb.dz, W.w("mlp.4.weight"), hb.dh2, R, Hd, Bt, trans_b=True, epilogue=E.EPI_DGELU, aux_in=hb.h2p)
        on_side(lambda ws: ops.linear(hb.dh2, hb.h1, W.g("mlp.2.weight"), Hd, Hd, R, trans_a=True, trans_b=True, epilogue=ACC,
                                      colsum_a=W.g("mlp.2.bias"), workspace=ws))
        ops.linear(hb.dh2, W.w("mlp.2.weight"), hb.dh1, R, Hd, Hd, trans_b=True, epilogue=E.EPI_DGELU, aux_in=hb.h1p, workspace=self.ws)
        on_side(lambda ws: ops.linear(hb.dh1, hb.feats, W.g("mlp.0.weight"), Hd, D, R, trans_a=True, trans_b=True, epilogue=ACC,
                                      colsum_a=W.g("mlp.0.bias"), workspace=ws))
        ops.linear(hb.dh1, W.w("mlp.0.weight"), hb.dfeats, R, D, Hd, trans_b=True, workspace=self.ws)


# --------------------------------------------------------------------------- #
# data-parallel reduction hooks (engine stays importable without torch.distributed)
# --------------------------------------------------------------------------- #
class NoReducer:
    world = 1

    def reduce_range(self, buf, lo, hi):
        pass

    def reduce_tensor(self, t):
        pass

    def finish(self):
        pass


# --------------------------------------------------------------------------- #
# one DINO multi-crop training step (paper Alg. 1)
# --------------------------------------------------------------------------- #
class DinoEngine:
    def __init__(self, arch="vit_small", img_size=224, out_dim=65536, batch=64, tile=256, n_global=2, n_local=8,
                 gsize=224, lsize=96, hidden=2048, bottleneck=256, lr=5e-4, weight_decay=0.04, betas=(0.9, 0.999), eps=1e-8,
                 momentum_teacher=0.996, student_temp=0.1, teacher_temp=0.04, center_momentum=0.9, clip_grad: float = 0.0,
                 mean=MEAN_RON, std=STD_RON, windows=None, device="cuda:0", reducer=None, precision: str = "bf16", clip_mode: str = "norm"):
        """``precision``: "bf16" (the training path) or "fp32" (every GEMM / attention operand, the head activations, the
        weight-normalised prototype matrix and the logit gradient in f32 -- the verification mode of SURVEY 8d's fp32 column).
        ``clip_mode``: "norm" (global norm, the reference default) or "value" (element-wise clamp), train.py:1072-1077."""
        if clip_mode not in ("norm", "value"):
            raise ValueError(f"clip_mode {clip_mode!r}: 'norm' or 'value'")
        self.clip_mode = clip_mode
        if precision not in ("bf16", "fp32"):
            raise ValueError(f"precision {precision!r}: 'bf16' or 'fp32'")
        fp32 = precision == "fp32"
        act = f32 if fp32 else bf16
        self.precision = precision
        dev = torch.device(device)
        # the float16 library build (--amp --amp-dtype float16) trains under dynamic loss scaling: torch's GradScaler on the device
        self.scaler = ops.LossScaler(dev) if (L.ACT_FORMAT == "f16" and not fp32) else None
        self.dev, self.arch, self.B, self.tile = dev, arch, batch, tile
        D = ARCHS[arch]["embed_dim"]
        self.D, self.K, self.G, self.V = D, out_dim, n_global, n_global + n_local
        self.mean, self.std = tuple(mean), tuple(std)
        self.gsize, self.lsize = gsize, lsize
        self._gcrops = self._lcrops = self._vstats = None       # crop buffers of the random-resized-crop path (allocated on first use)
        if windows is None:   # deterministic crop windows of SURVEY 8(d): (y0, x0)
            windows = [(16 * g, 16 * g) for g in range(n_global)] + [(20 * l, 160 - 20 * l) for l in range(n_local)]
        self.gwins, self.lwins = list(windows[:n_global]), list(windows[n_global:])
        specs = OrderedDict(("backbone." + k, v) for k, v in vit_param_specs(arch, img_size, 0).items())
        specs.update(("head." + k, v) for k, v in dino_head_specs(D, out_dim, hidden, bottleneck).items())
        self.arena = Arena(specs, dev, teacher=True)
        self.sW, self.tW = Weights(self.arena, "backbone.", fp32=fp32), Weights(self.arena, "backbone.", teacher=True, fp32=fp32)
        self.sH, self.tH = Weights(self.arena, "head.", fp32=fp32), Weights(self.arena, "head.", teacher=True, fp32=fp32)
        self.vit = VitRunner(arch, img_size, dev, fp32=fp32)
        self.head = HeadRunner(D, out_dim, hidden, bottleneck, dev)
        B = batch
        segs = [(n_global * B, gsize)] + ([(n_local * B, lsize)] if n_local else [])
        self.g_stu = VitGroup(arch, segs, img_size, dev, save=True, act=act)   # all student crops, tokens concatenated
        self.g_teach = VitGroup(arch, [(n_global * B, gsize)], img_size, dev, save=False, act=act)
        self.s_wins = [self.gwins] + ([self.lwins] if n_local else [])
        self.hb_s = HeadBuffers(self.V * B, D, out_dim, hidden, bottleneck, dev, save=True, act=act)
        self.hb_t = HeadBuffers(n_global * B, D, out_dim, hidden, bottleneck, dev, save=False, act=act)
        self.wn_s = _empty((out_dim, bottleneck), act, dev)
        self.wn_t = _empty((out_dim, bottleneck), act, dev)
        self.center = torch.zeros(out_dim, dtype=f32, device=dev)
        self.center_sum = torch.zeros(out_dim, dtype=f32, device=dev)
        self.loss = torch.zeros(1, dtype=f32, device=dev)
        self.loss_ws = _empty((2 * (self.V + self.G) * B,), f32, dev)
        self.gnorm_sq = torch.zeros(1, dtype=f32, device=dev)
        self.red_ws = _empty((1024,), f32, dev)
        self.hyper = torch.zeros(L.HYP_COUNT, dtype=f32, device=dev)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.m_teacher, self.ts, self.tt, self.cm, self.clip = momentum_teacher, student_temp, teacher_temp, center_momentum, clip_grad
        self.train_last_layer = True
        self.t = 0
        self.reducer = reducer if reducer is not None else NoReducer()
        self._n_micro = 1
        self._plan = None             # gradient all-reduce ranges (dist.reduction_plan), built on first use
        # the teacher's forward runs on the side stream beside the student's; GIPVIT_TEACHER_SIDE=0 queues it in front instead (A/B runs)
        self._teacher_on_side = os.environ.get("GIPVIT_TEACHER_SIDE", "1") != "0"
        if torch.device(device).type == "cuda":
            self._ev_fork, self._ev_join, self._ev_zero = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        # contiguous arena range of every block's weight-decayed matrices (arena order = backward order)
        self._block_range = {}
        for i in range(ARCHS[arch]["depth"]):
            names = [n for n in self.arena.order if n.startswith(f"backbone.blocks.{i}.") and self.arena.off[n] < self.arena.n_decay]
            self._block_range[i] = (min(self.arena.off[n] for n in names),
                                    max(self.arena.off[n] + _round_up(math.prod(self.arena.specs[n]), PAD) for n in names))

    # ---- parameters ----------------------------------------------------------------
    def load_state(self, backbone: Dict[str, torch.Tensor], head: Dict[str, torch.Tensor]):
        st = {"backbone." + k: v for k, v in backbone.items()}
        st.update({"head." + k: v for k, v in head.items()})
        self.arena.load(st)
        self.arena.t.copy_(self.arena.p)            # teacher starts as a copy of the student
        self.refresh_bf16()

    def load_teacher_state(self, backbone: Dict[str, torch.Tensor], head: Dict[str, torch.Tensor], center: Optional[torch.Tensor] = None):
        """Restore the EMA teacher (and the centre) of a saved run: without it a resumed run would restart the
        teacher from the student while the momentum schedule is already near 1."""
        st = {"backbone." + k: v for k, v in backbone.items()}
        st.update({"head." + k: v for k, v in head.items()})
        self.arena.load(st, self.arena.t)
        if center is not None:
            self.center.copy_(center.to(self.center.device, f32).view(-1))
        self.refresh_bf16()

    def refresh_bf16(self):
        a = self.arena
        ops.cast_bf16(a.p, a.pb); ops.cast_bf16(a.t, a.tb)
        self._refresh_wn()

    def _refresh_wn(self):
        bt = self.head.bott
        ops.weightnorm_fwd(self.sH.f("last_layer.weight_v"), self.sH.f("last_layer.weight_g").view(-1), self.wn_s, self.K, bt)
        ops.weightnorm_fwd(self.tH.f("last_layer.weight_v"), self.tH.f("last_layer.weight_g").view(-1), self.wn_t, self.K, bt)

    def backbone_state_dict(self, teacher: bool = False):
        return self.arena.state_dict(self.arena.t if teacher else self.arena.p, "backbone.")

    def head_state_dict(self, teacher: bool = False):
        return self.arena.state_dict(self.arena.t if teacher else self.arena.p, "head.")

    def grads(self) -> Dict[str, torch.Tensor]:
        """Copies of the parameter gradients of the last forward_backward.  Under float16 loss scaling the arena holds S x gradient:
        the copies are divided by the scale in force (call before optimizer_step, whose scaler update may change S)."""
        inv = None if self.scaler is None else 1.0 / self.scaler.state[0]
        return {n: self.arena.view(self.arena.g, n).detach().clone() if inv is None else self.arena.view(self.arena.g, n).detach() * inv
                for n in self.arena.specs}

    # ---- the step --------------------------------------------------------------------
    def set_drop_path(self, per_img: Optional[torch.Tensor]):
        """Stochastic depth for the STUDENT's next steps (the teacher runs without, as in DINO): f32 [depth, 2, V * B] on the
        device, keep / (1 - p) per crop image (global crops first, crop-major like the feature rows), or None."""
        self.g_stu.set_drop(per_img)

    def set_dropout(self, p: float, seed: int = 0):
        """--drop for the STUDENT's next step (the teacher runs in eval mode): probability and this step's 32-bit seed; p = 0 switches it off.
        Under gradient accumulation every micro-batch of the step derives its own seed from this one (forward_backward)."""
        self._drop = (float(p), int(seed) & 0xFFFFFFFF) if p else None
        self.g_stu.set_dropout(p, seed)

    def set_hyper(self, lr=None, wd=None, momentum_teacher=None, teacher_temp=None, n_micro: int = 1):
        """Put the per-step schedule values into the device hyper vector with a tiny stream-ordered
        kernel whose ARGUMENTS carry them.  Not a memcpy: the host runs several steps ahead of the
        GPU, and a pinned staging buffer would be rewritten before the queued steps consumed it."""
        self.t += 1
        vals = [0.0] * L.HYP_COUNT
        vals[L.HYP_LR] = self.lr if lr is None else lr
        vals[L.HYP_WD] = self.wd if wd is None else wd
        vals[L.HYP_BC1] = 1.0 - self.betas[0] ** self.t
        vals[L.HYP_BC2] = 1.0 - self.betas[1] ** self.t
        vals[L.HYP_TEACHER_MOM] = self._cur_mom = self.m_teacher if momentum_teacher is None else momentum_teacher
        vals[L.HYP_GRAD_SCALE] = 1.0 / (self.reducer.world * n_micro)
        vals[L.HYP_TEACHER_TEMP] = self.tt if teacher_temp is None else teacher_temp
        vals[L.HYP_STUDENT_TEMP] = self.ts
        ops.store_f32(self.hyper, vals)

    def forward_backward(self, tiles_u8: torch.Tensor, micro: Tuple[int, int] = (0, 1), boxes=None, fill=None, views=None):
        """teacher fwd (global crops) -> student fwd (all crops) -> loss -> backward.
        Leaves gradients in arena.g, loss in self.loss, center_sum.  ``micro = (j, n)``: this is
        micro-batch j of n (gradient accumulation): gradients, loss and centre sums add up over
        the n calls and the data-parallel reduction is issued from the last one only.
        ``views``: (global, local) packed per-crop records of multicrop.ViewAugmentSampler.sample -- with ``boxes``, every crop
        gets its own colour jitter / grayscale / blur / solarisation in the pass that cuts it (gv_crop_augment)."""
        B, G, V = self.B, self.G, self.V
        a = self.arena
        mj, mn = micro
        first, last = mj == 0, mj == mn - 1
        if mn > 1 and getattr(self, "_drop", None):      # micro-batches of one step must not share their dropout masks
            self.g_stu.set_dropout(self._drop[0], ops.dropout_site_seed(self._drop[1], 15, mj))
        t_src, t_win, s_src, s_win = tiles_u8, [self.gwins], tiles_u8, self.s_wins
        if boxes is not None and fill is not None:
            raise ValueError("fill boxes are tile coordinates: they cannot be combined with re-cut random crops (boxes)")
        if views is not None and boxes is None:
            raise ValueError("views (per-crop augmentation draws) need boxes: the views are produced by the pass that cuts the crops")
        if boxes is not None:
            # random-resized crops (multicrop.MultiCropSampler): cut on the device, then every crop is its own
            # "tile" with the single window (0, 0); rows stay crop-major like the fixed windows
            bg, bl = boxes
            assert bg.shape[0] == self.G * B and (self.V == self.G or bl.shape[0] == (self.V - self.G) * B)
            if self._gcrops is None:
                self._gcrops = _empty((self.G * B, self.gsize, self.gsize, 3), torch.uint8, self.dev)
                self._lcrops = _empty(((self.V - self.G) * B, self.lsize, self.lsize, 3), torch.uint8, self.dev) if self.V > self.G else None
            if views is not None and self._vstats is None:
                self._vstats = torch.zeros(max(self.G, self.V - self.G) * B, dtype=torch.int64, device=self.dev)
            if views is None:
                ops.crop_resize(tiles_u8, bg, self.gsize, out=self._gcrops)
            else:
                ops.crop_augment(tiles_u8, bg, views[0], self._vstats, self.gsize, out=self._gcrops)
            t_src, t_win = self._gcrops, [[(0, 0)]]
            s_src, s_win = [self._gcrops], [[(0, 0)]]
            if self.V > self.G:
                if views is None:
                    ops.crop_resize(tiles_u8, bl, self.lsize, out=self._lcrops)
                else:
                    ops.crop_augment(tiles_u8, bl, views[1], self._vstats, self.lsize, out=self._lcrops)
                s_src.append(self._lcrops); s_win.append([(0, 0)])
        # the teacher's forward shares nothing with the student's until the loss: it runs on the
        # side stream beside the student forward (fills the tail of each other's kernels)
        side = self.vit.side if tiles_u8.is_cuda else None
        t_side = side if self._teacher_on_side else None
        # fills nothing in the forward depends on (the gradient arena, the backward's zero-initialised scratch) go to the side
        # stream, beside the forward pass; the main stream picks their event up before the head backward
        if side is not None:
            main = torch.cuda.current_stream()
            self._ev_fork.record(main); side.wait_event(self._ev_fork)
            with torch.cuda.stream(side):
                if first:
                    a.g.zero_()
                self.vit.prepare_backward(self.g_stu)
                self._ev_zero.record(side)
        else:
            if first:
                a.g.zero_()
        if t_side is not None:
            with torch.cuda.stream(side):
                self.vit.forward(self.tW, self.g_teach, t_src, t_win, self.mean, self.std, self.hb_t.feats, fill=fill)
                self.head.forward(self.tH, self.wn_t, self.hb_t, on_side=True)
                self._ev_join.record(side)
        else:
            self.vit.forward(self.tW, self.g_teach, t_src, t_win, self.mean, self.std, self.hb_t.feats, fill=fill)
            self.head.forward(self.tH, self.wn_t, self.hb_t)
        self.vit.forward(self.sW, self.g_stu, s_src, s_win, self.mean, self.std, self.hb_s.feats, fill=fill)
        self.head.forward(self.sH, self.wn_s, self.hb_s)
        if t_side is not None:
            main.wait_event(self._ev_join)
        ops.dino_loss(self.hb_s.logits, self.hb_t.logits, self.center, self.hb_s.dlogits, self.loss, self.center_sum,
                      self.loss_ws, B, V, G, self.K, self.ts, self.tt, hyper=self.hyper,
                      loss_scale=self.scaler.scale if self.scaler is not None else None)
        if mn > 1:
            if first:
                self._loss_acc, self._center_acc = self.loss.clone(), self.center_sum.clone()
            else:
                self._loss_acc += self.loss; self._center_acc += self.center_sum
            if last:
                self.loss.copy_(self._loss_acc / mn); self.center_sum.copy_(self._center_acc)
        if last:
            self.reducer.reduce_tensor(self.center_sum)
        if side is not None:
            main.wait_event(self._ev_zero)
        self.head.backward(self.sH, self.wn_s, self.hb_s, self.train_last_layer, side=side)
        if not last:        # more micro-batches follow: gradients keep accumulating locally
            self.vit.backward(self.sW, self.g_stu, self.hb_s.dfeats)
            return
        # The gradient all-reduce goes out as a few contiguous arena ranges, each as soon as it is final (dist.reduction_plan):
        # the head's matrices when the head backward ends, then coalesced block ranges of >= 25 MB from the side stream (the
        # block's weight gradients are produced there), then block 0 + patch embed + the no-decay tail as ONE message when
        # backward ends -- RCCL's stream runs beside the remaining backward (reference: NativeDDP's buckets, train.py:634, 1071)
        plan = self._reduction_plan()
        active = self.reducer.world > 1 or getattr(self.reducer, "always", False)

        def release(trigger):
            if active:
                for trg, lo, hi in plan:
                    if trg == trigger:
                        self.reducer.reduce_range(a.g, lo, hi)
        if side is not None:              # the head's weight gradients were queued on the side stream
            with torch.cuda.stream(side):
                release("head")
        else:
            release("head")
        self.vit.backward(self.sW, self.g_stu, self.hb_s.dfeats, on_block_done=release)
        release("end")
        self.reducer.finish()

    def _reduction_plan(self):
        if self._plan is None:
            from .dist import reduction_plan
            a = self.arena
            head_names = [n for n in a.order if n.startswith("head.") and a.off[n] < a.n_decay]
            head_span = (min(a.off[n] for n in head_names), max(a.span(n)[1] for n in head_names)) if head_names else None
            self._plan = reduction_plan(head_span, self._block_range, a.n, bucket_bytes=int(os.environ.get("GIPVIT_BUCKET_MB", "25")) << 20)
        return self._plan

    def optimizer_step(self):
        a = self.arena
        by_norm = self.clip > 0 and self.clip_mode == "norm"
        scaled = self.scaler is not None
        if by_norm or scaled:           # loss scaling: the sum of squares is also the finite check of GradScaler.step()
            ops.sumsq(a.g, self.red_ws, self.gnorm_sq)
        kw = dict(lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, step=max(self.t, 1),
                  clip_norm=self.clip if by_norm else 0.0, gnorm_sq=self.gnorm_sq if (by_norm or scaled) else None, hyper=self.hyper,
                  clip_value=self.clip if (self.clip > 0 and self.clip_mode == "value") else 0.0,
                  loss_scale=self.scaler.scale if scaled else None)
        lo = 0
        if not self.train_last_layer:
            # the head's last layer is frozen for the first epochs (DINO cancel_gradients_last_layer sets its grad to
            # None, so the optimizer SKIPS it: no decay, no moment update) -- only the teacher EMA runs over its range
            lo, hi = a.span("head.last_layer.weight_v")
            assert lo == 0, "the last layer opens the arena (backward-completion order)"
            sl = slice(lo, hi)
            ops.adamw_ema(a.p[sl], a.g[sl], a.m[sl], a.v[sl], a.pb[sl], a.t[sl], a.tb[sl], hi - lo, lr=0.0, beta1=self.betas[0],
                          beta2=self.betas[1], eps=self.eps, weight_decay=0.0, step=1, hyper=self.hyper, mode=3)
            lo = hi
        sl = slice(lo, a.n_decay)
        ops.adamw_ema(a.p[sl], a.g[sl], a.m[sl], a.v[sl], a.pb[sl], a.t[sl], a.tb[sl], a.n_decay - lo, weight_decay=1.0, **kw)
        if a.n > a.n_decay:   # biases / LN / pos / cls: same schedules, weight-decay multiplier 0
            sl = slice(a.n_decay, a.n)
            ops.adamw_ema(a.p[sl], a.g[sl], a.m[sl], a.v[sl], a.pb[sl], a.t[sl], a.tb[sl], a.n - a.n_decay, weight_decay=0.0, **kw)
        if scaled:
            self.scaler.update(self.gnorm_sq)
        self._refresh_wn()
        ops.center_update(self.center, self.center_sum, self.K, self.cm, 1.0 / (self.G * self.B * self.reducer.world * self._n_micro))

    def step_micro(self, tile_batches: Sequence[torch.Tensor], **sched) -> torch.Tensor:
        """One optimizer step over ``len(tile_batches)`` micro-batches of B tiles each (gradient
        accumulation; BASELINE config 5 reaches 512 tiles per GPU this way).  Equals ``step`` on the
        concatenated batch up to summation order."""
        n = len(tile_batches)
        self.set_hyper(n_micro=n, **sched)
        for j, tb in enumerate(tile_batches):
            assert tb.shape == (self.B, self.tile, self.tile, 3) and tb.dtype == torch.uint8
            self.forward_backward(tb, micro=(j, n))
        self._n_micro = n
        try:
            self.optimizer_step()
        finally:
            self._n_micro = 1
        return self.loss

    def step(self, tiles_u8: torch.Tensor, boxes=None, fill=None, views=None, **sched) -> torch.Tensor:
        """One full training step on [B, tile, tile, 3] uint8 NHWC tiles.  Returns the
        (device, un-synchronised) loss tensor.  ``boxes``: (global, local) int32 device tensors from
        multicrop.MultiCropSampler.sample -- random-resized crops instead of the fixed parity windows."""
        assert tiles_u8.shape == (self.B, self.tile, self.tile, 3) and tiles_u8.dtype == torch.uint8
        self.set_hyper(**sched)
        self.forward_backward(tiles_u8, boxes=boxes, fill=fill, views=views)
        self.optimizer_step()
        return self.loss


# --------------------------------------------------------------------------- #
# supervised single-crop step (reference train.py:1044-1078; BASELINE config 1)
# --------------------------------------------------------------------------- #
class SupervisedEngine:
    """ViT + Linear head, softmax -> LabelSmoothingCE (the reference's actual loss path)."""

    def __init__(self, arch="vit_tiny", img_size=64, num_classes=2, batch=8, lr=1e-3, weight_decay=0.0, betas=(0.9, 0.999),
                 eps=1e-8, smoothing=0.1, clip_grad: float = 0.0, mean=MEAN_RON, std=STD_RON, device="cuda:0", reducer=None,
                 opt: str = "adamw", momentum: float = 0.9, train_backbone: bool = True, model_ema_decay: Optional[float] = None,
                 precision: str = "bf16", clip_mode: str = "norm"):
        """``precision``: "bf16" (the training path: bf16 GEMM / attention operands, f32 accumulation and residual stream) or
        "fp32" (the reference's default arithmetic: every operand f32, csrc/f32path.hip -- the mode the 1e-4 parity gates
        of SURVEY 8d are stated for; an order of magnitude slower, kept for verification)."""
        if precision not in ("bf16", "fp32"):
            raise ValueError(f"precision {precision!r}: 'bf16' or 'fp32'")
        if clip_mode not in ("norm", "value", "agc"):
            raise ValueError(f"clip_mode {clip_mode!r}: 'norm', 'value' or 'agc' (train.py:1072-1077)")
        if clip_mode == "value" and opt == "lamb":
            raise ValueError("--clip-mode value with --opt lamb is not built (Lamb's own global-norm clip needs the norm of the clamped gradient)")
        self.clip_mode = clip_mode
        fp32 = precision == "fp32"
        act = f32 if fp32 else bf16
        self.precision = precision
        dev = torch.device(device)
        # float16 library build: dynamic loss scaling (torch GradScaler via timm NativeScaler, reference train.py:585-602, 1061-1070)
        self.scaler = ops.LossScaler(dev) if (L.ACT_FORMAT == "f16" and not fp32) else None
        if self.scaler is not None and (opt == "lamb" or clip_mode == "agc"):
            raise ValueError("float16 loss scaling is built for adamw / adam / sgd with --clip-mode norm | value; "
                             "--opt lamb and --clip-mode agc run with --amp-dtype bfloat16")
        self.dev, self.arch, self.B, self.img, self.C = dev, arch, batch, img_size, num_classes
        D = ARCHS[arch]["embed_dim"]
        self.D = D
        self.mean, self.std = tuple(mean), tuple(std)
        # --model-ema (train.py:615-622, ModelEmaV2): the EMA copy lives in the arena's teacher slot and is
        # updated by the same fused optimizer pass (train.py:1080-1081)
        self.ema_decay = model_ema_decay
        self.arena = Arena(vit_param_specs(arch, img_size, num_classes), dev, teacher=model_ema_decay is not None)
        self.W = Weights(self.arena, "", fp32=fp32)
        self.Wema = Weights(self.arena, "", teacher=True, fp32=fp32) if model_ema_decay is not None else None
        self.vit = VitRunner(arch, img_size, dev, fp32=fp32)
        self.grp = VitGroup(arch, [(batch, img_size)], img_size, dev, save=True, act=act)
        e = lambda shape, dt: _empty(shape, dt, dev)
        self.feats, self.dfeats = e((batch, D), act), e((batch, D), act)
        self.logits, self.dlogits, self.prob = e((batch, num_classes), f32), e((batch, num_classes), f32), e((batch, num_classes), f32)
        self.loss = torch.zeros(1, dtype=f32, device=dev)
        self.ones = torch.ones(batch, dtype=f32, device=dev)
        self.gnorm_sq = torch.zeros(1, dtype=f32, device=dev)
        self.red_ws = _empty((1024,), f32, dev)
        self.lr, self.wd, self.betas, self.eps, self.smoothing, self.clip = lr, weight_decay, betas, eps, smoothing, clip_grad
        self.t = 0
        self.reducer = reducer if reducer is not None else NoReducer()
        if opt not in ("adamw", "adam", "sgd", "lamb"):
            raise ValueError(f"--opt {opt}: this build fuses adamw / adam / sgd (nesterov) / lamb only")
        self.opt = opt
        self.opt_mode = {"adamw": 0, "adam": 1, "sgd": 2, "lamb": -1}[opt]
        if opt == "sgd":
            self.betas = (momentum, betas[1])
        if clip_mode == "agc" and clip_grad > 0:
            # adaptive gradient clipping (timm adaptive_clip_grad; the reference excludes the classifier: model_parameters(exclude_head=True))
            a = self.arena
            self._agc_units = ops.agc_units([(a.off[n], a.specs[n]) for n in a.order if not n.startswith("head.")]).to(dev)
        if opt == "lamb":
            # timm.optim.Lamb (reference train.py:161, 583): per-tensor trust ratio -> every tensor's norms are needed before its
            # update: two launches per arena segment over a block table that never crosses a tensor (gv_lamb)
            a = self.arena
            self._lamb_names = list(a.order)
            spans = [a.span(n) for n in self._lamb_names]
            self._lamb_tab = ops.lamb_block_table(spans).to(dev)
            self._lamb_decay = torch.tensor([0.0 if no_weight_decay(n, a.specs[n]) else 1.0 for n in self._lamb_names])
            # decayed tensors open the arena (Arena layout): two sub-tables, one per weight-decay value
            nd = sum(1 for n in self._lamb_names if not no_weight_decay(n, a.specs[n]))
            tid = self._lamb_tab[:, 0]
            self._lamb_tabs = (self._lamb_tab[tid < nd].contiguous(), self._lamb_tab[tid >= nd].contiguous())
            self._lamb_stats = torch.zeros(2 * len(self._lamb_names), dtype=f32, device=dev)
            self.lamb_max_grad_norm = 1.0          # timm Lamb default
        self.train_backbone = train_backbone      # False = --no-grad head-only fine-tune (train.py:497-503)

    def load_state(self, state: Dict[str, torch.Tensor], ema_state: Optional[Dict[str, torch.Tensor]] = None):
        self.arena.load(state)
        ops.cast_bf16(self.arena.p, self.arena.pb)
        if self.arena.t is not None:       # ModelEmaV2 starts as a deep copy of the model; a resumed run restores it
            if ema_state is not None:
                self.arena.load(ema_state, self.arena.t)
            else:
                self.arena.t.copy_(self.arena.p)
            ops.cast_bf16(self.arena.t, self.arena.tb)

    def state_dict(self, ema: bool = False):
        return self.arena.state_dict(self.arena.t if ema else None)

    def set_drop_path(self, per_img: Optional[torch.Tensor]):
        """Stochastic depth (--drop-path) for the next training steps: f32 [depth, 2, batch] on the device = keep / (1 - p)
        per image and residual branch, or None (evaluation: ``forward`` with it set would scale the branches too)."""
        self.grp.set_drop(per_img)

    def set_dropout(self, p: float, seed: int = 0):
        """--drop for the next training steps' forward / backward: probability and the step's 32-bit seed (evaluation: p = 0)."""
        self.grp.set_dropout(p, seed)

    def grads(self):
        """As DinoEngine.grads: the gradient itself, whatever the loss scale."""
        inv = None if self.scaler is None else 1.0 / self.scaler.state[0]
        return {n: self.arena.view(self.arena.g, n).detach().clone() if inv is None else self.arena.view(self.arena.g, n).detach() * inv
                for n in self.arena.specs}

    def forward(self, tiles_u8, ema: bool = False, fill=None):
        """Inference / features: returns (logits f32 [B,C], CLS features bf16 [B,D]); ``ema``: with the EMA weights."""
        B, C, D, W = self.B, self.C, self.D, (self.Wema if ema else self.W)
        self.vit.forward(W, self.grp, tiles_u8, [[(0, 0)]], self.mean, self.std, self.feats, fill=fill)
        ops.small_matmul(self.feats, W.f("head.weight"), self.logits, B, C, D, sam=D, sak=1, sbk=1, sbn=D, bias=W.f("head.bias"))
        return self.logits, self.feats

    def forward_backward(self, tiles_u8, target, fill=None):
        B, C, D, W = self.B, self.C, self.D, self.W
        self.arena.g.zero_()
        self.forward(tiles_u8, fill=fill)
        ops.softmax_lsce(self.logits, target.view(-1), self.loss, self.dlogits, self.prob, B, C, self.smoothing,
                         loss_scale=self.scaler.scale if self.scaler is not None else None)
        # head backward: dW = dlogits^T f, db = colsum(dlogits), df = dlogits W
        ops.small_matmul(self.dlogits, self.feats, W.g("head.weight"), C, D, B, sam=1, sak=C, sbk=D, sbn=1, accumulate=True)
        ops.small_matmul(self.ones, self.dlogits, W.g("head.bias").view(1, C), 1, C, B, sam=0, sak=1, sbk=C, sbn=1, accumulate=True)
        ops.small_matmul(self.dlogits, W.f("head.weight"), self.dfeats, B, D, C, sam=C, sak=1, sbk=D, sbn=1)
        if self.train_backbone:
            self.vit.backward(W, self.grp, self.dfeats)
        self.reducer.reduce_range(self.arena.g, 0, self.arena.n)
        self.reducer.finish()

    def optimizer_step(self, lr=None):
        a = self.arena
        self.t += 1
        by_norm = self.clip > 0 and self.clip_mode == "norm"
        if self.clip > 0 and self.clip_mode == "agc":
            ops.agc(a.p, a.g, self._agc_units, self.clip, 1e-3, 1.0 / self.reducer.world)
        scaled = self.scaler is not None
        if by_norm or scaled:
            ops.sumsq(a.g, self.red_ws, self.gnorm_sq)
        kw = dict(lr=self.lr if lr is None else lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, step=self.t,
                  grad_scale=1.0 / self.reducer.world, clip_norm=self.clip if by_norm else 0.0,
                  gnorm_sq=self.gnorm_sq if (by_norm or scaled) else None,
                  mode=self.opt_mode, teacher_momentum=self.ema_decay or 0.0,
                  clip_value=self.clip if (self.clip > 0 and self.clip_mode == "value") else 0.0,
                  loss_scale=self.scaler.scale if scaled else None)
        tt = (lambda sl: (a.t[sl], a.tb[sl])) if a.t is not None else (lambda sl: (None, None))
        if self.opt_mode < 0:
            if not self.train_backbone:
                raise ValueError("--opt lamb with --no-grad (head-only fine-tune) is not built: use adamw / adam / sgd")
            ops.sumsq(a.g, self.red_ws, self.gnorm_sq)              # Lamb clips by the global norm itself (max_grad_norm = 1)
            self._lamb_stats.zero_()
            kl = dict(lr=kw["lr"], beta1=kw["beta1"], beta2=kw["beta2"], eps=self.eps, step=self.t, grad_scale=kw["grad_scale"],
                      clip_norm=self.clip if by_norm else 0.0, max_grad_norm=self.lamb_max_grad_norm, teacher_momentum=self.ema_decay or 0.0)
            for phase in (0, 1):
                for tab, wd in zip(self._lamb_tabs, (self.wd, 0.0)):
                    if tab.shape[0]:
                        ops.lamb(a.p, a.g, a.m, a.v, a.pb, a.t, a.tb, tab, self._lamb_stats, self.gnorm_sq, phase=phase, weight_decay=wd, **kl)
            return
        if self.train_backbone:
            ranges = [(0, a.n_decay, self.wd), (a.n_decay, a.n, 0.0)]
        else:
            # --no-grad (train.py:497-503): the encoder has requires_grad=False, so the optimizer never sees it -- no
            # weight decay, no moments; only the classifier's two tensors are stepped (the EMA of the frozen part
            # stays equal to the model it was copied from)
            ranges = [a.span("head.weight") + (self.wd,), a.span("head.bias") + (0.0,)]
        for lo, hi, wd in ranges:
            sl = slice(lo, hi)
            ops.adamw_ema(a.p[sl], a.g[sl], a.m[sl], a.v[sl], a.pb[sl], *tt(sl), hi - lo, weight_decay=wd, **kw)
        if scaled:
            self.scaler.update(self.gnorm_sq)

    def step(self, tiles_u8, target, lr=None, fill=None):
        assert tiles_u8.dtype == torch.uint8 and target.dtype == torch.int64
        self.forward_backward(tiles_u8, target, fill=fill)
        self.optimizer_step(lr)
        return self.loss


# --------------------------------------------------------------------------- #
# forward-only encoder: slide-level inference / feature extraction
# --------------------------------------------------------------------------- #
class FeatureExtractor:
    """The encoder run forward-only on the same kernels: per-tile CLS features (what the reference's
    ``validate()`` writes to ``<slide>_features.pt``, train.py:1281-1282) and, with a classifier head,
    per-tile logits / positive-class scores (train.py:1186-1343).  Nothing is saved for a backward pass:
    the group rotates three residual buffers and the fc1 epilogue skips the pre-activation store.

    ``weights``: a ``Weights`` view of another engine's arena (student, EMA or DINO teacher) -- the
    extractor then evaluates those live parameters instead of owning a copy (validation between epochs)."""

    def __init__(self, arch="vit_small", img_size=256, batch=256, num_classes=0, mean=MEAN_RON, std=STD_RON, device="cuda:0",
                 weights: Optional[Weights] = None, precision: str = "bf16"):
        """``precision``: as for the engines; with ``weights`` it follows the owning engine's mode."""
        if precision not in ("bf16", "fp32"):
            raise ValueError(f"precision {precision!r}: 'bf16' or 'fp32'")
        fp32 = weights.fp32 if weights is not None else precision == "fp32"
        act = f32 if fp32 else bf16
        dev = torch.device(device)
        self.dev, self.arch, self.B, self.img, self.C = dev, arch, batch, img_size, num_classes
        self.D = ARCHS[arch]["embed_dim"]
        self.mean, self.std = tuple(mean), tuple(std)
        if weights is None:
            self.arena = Arena(vit_param_specs(arch, img_size, num_classes), dev, teacher=False)
            self.W = Weights(self.arena, "", fp32=fp32)
        else:
            self.arena, self.W = weights.a, weights
        self.vit = VitRunner(arch, img_size, dev, fp32=fp32)
        self.vit.side = None                       # a forward-only pass has nothing to put on a side stream
        self.grp = VitGroup(arch, [(batch, img_size)], img_size, dev, save=False, act=act)
        self.feats = _empty((batch, self.D), act, dev)
        self.logits = _empty((batch, num_classes), f32, dev) if num_classes else None
        self._pad = None

    def load_state(self, state: Dict[str, torch.Tensor]):
        self.arena.load(state)
        ops.cast_bf16(self.arena.p, self.arena.pb)

    def forward(self, tiles_u8: torch.Tensor):
        """tiles_u8 [B, img, img, 3] u8 NHWC -> (CLS features bf16 [B, D], logits f32 [B, C] or None)."""
        assert tiles_u8.shape == (self.B, self.img, self.img, 3) and tiles_u8.dtype == torch.uint8
        self.vit.forward(self.W, self.grp, tiles_u8, [[(0, 0)]], self.mean, self.std, self.feats)
        if self.C:
            ops.small_matmul(self.feats, self.W.f("head.weight"), self.logits, self.B, self.C, self.D, sam=self.D, sak=1, sbk=1, sbn=self.D,
                             bias=self.W.f("head.bias"))
        return self.feats, self.logits

    def run(self, tiles_u8: torch.Tensor):
        """Any number of tiles (one chunk of a slide, datasets.py:699-700 ``tiles_per_iter``): batches of B, the last
        one padded.  -> (features f32 [n, D], logits f32 [n, C] or None), device tensors owned by the caller."""
        n = tiles_u8.shape[0]
        feats = torch.empty(n, self.D, dtype=f32, device=self.dev)
        logits = torch.empty(n, self.C, dtype=f32, device=self.dev) if self.C else None
        for lo in range(0, n, self.B):
            hi = min(n, lo + self.B)
            part = tiles_u8[lo:hi]
            if hi - lo < self.B:
                if self._pad is None:
                    self._pad = torch.zeros(self.B, self.img, self.img, 3, dtype=torch.uint8, device=self.dev)
                self._pad[: hi - lo].copy_(part)
                part = self._pad
            f, l = self.forward(part)
            feats[lo:hi].copy_(f[: hi - lo])
            if logits is not None:
                logits[lo:hi].copy_(l[: hi - lo])
        return feats, logits
