"""CPU: pin the oracle (oracle/) against independent implementations that are installed here
-- the reference holds no tests or golden vectors, so this is what stands
behind "the oracle restates the reference's arithmetic" -- and against the committed fixtures."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import step_oracle as so, vit_oracle as vo

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_primitives_match_torch_nn():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 17, 192, generator=g, dtype=torch.float64)
    w, b = torch.randn(192, generator=g, dtype=torch.float64), torch.randn(192, generator=g, dtype=torch.float64)
    assert torch.allclose(vo.layer_norm(x, w, b), F.layer_norm(x, (192,), w, b, 1e-6), atol=1e-12)
    assert torch.allclose(vo.gelu(x), F.gelu(x), atol=1e-12)


def test_attention_matches_sdpa():
    p = vo.init_vit("vit_tiny", 64, 0, seed=3, dtype=torch.float64)
    x = torch.randn(2, 17, 192, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    got = vo.attention(x, p, "blocks.0.attn.", 3)
    qkv = (x @ p["blocks.0.attn.qkv.weight"].t() + p["blocks.0.attn.qkv.bias"]).reshape(2, 17, 3, 3, 64).permute(2, 0, 3, 1, 4)
    o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(2, 17, 192)
    ref = o @ p["blocks.0.attn.proj.weight"].t() + p["blocks.0.attn.proj.bias"]
    assert torch.allclose(got, ref, atol=1e-10)


def test_vit_matches_hf_vitmodel():
    """Same weights in a locally configured HF ViTModel (no download) give the same tokens."""
    from transformers import ViTConfig, ViTModel
    arch, img = "vit_tiny", 64
    a = vo.ARCHS[arch]
    cfg = ViTConfig(hidden_size=a["embed_dim"], num_hidden_layers=a["depth"], num_attention_heads=a["num_heads"],
                    intermediate_size=4 * a["embed_dim"], image_size=img, patch_size=16, layer_norm_eps=1e-6, qkv_bias=True,
                    hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    hf = ViTModel(cfg, add_pooling_layer=False).eval()
    p = vo.init_vit(arch, img, 0, seed=0)
    D = a["embed_dim"]
    sd = {"embeddings.cls_token": p["cls_token"], "embeddings.position_embeddings": p["pos_embed"],
          "embeddings.patch_embeddings.projection.weight": p["patch_embed.proj.weight"],
          "embeddings.patch_embeddings.projection.bias": p["patch_embed.proj.bias"],
          "layernorm.weight": p["norm.weight"], "layernorm.bias": p["norm.bias"]}
    for i in range(a["depth"]):        # key names of transformers 5.x ViTModel
        b, h = f"blocks.{i}.", f"layers.{i}."
        qw, qb = p[b + "attn.qkv.weight"], p[b + "attn.qkv.bias"]
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            sd[h + f"attention.{nm}.weight"] = qw[j * D:(j + 1) * D]
            sd[h + f"attention.{nm}.bias"] = qb[j * D:(j + 1) * D]
        sd[h + "attention.o_proj.weight"] = p[b + "attn.proj.weight"]; sd[h + "attention.o_proj.bias"] = p[b + "attn.proj.bias"]
        sd[h + "layernorm_before.weight"] = p[b + "norm1.weight"]; sd[h + "layernorm_before.bias"] = p[b + "norm1.bias"]
        sd[h + "layernorm_after.weight"] = p[b + "norm2.weight"]; sd[h + "layernorm_after.bias"] = p[b + "norm2.bias"]
        sd[h + "mlp.fc1.weight"] = p[b + "mlp.fc1.weight"]; sd[h + "mlp.fc1.bias"] = p[b + "mlp.fc1.bias"]
        sd[h + "mlp.fc2.weight"] = p[b + "mlp.fc2.weight"]; sd[h + "mlp.fc2.bias"] = p[b + "mlp.fc2.bias"]
    have = set(hf.state_dict())
    sd = {k: v for k, v in sd.items()}
    assert set(sd) == have, (sorted(set(sd) - have)[:5], sorted(have - set(sd))[:5])
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not [m for m in missing if "pooler" not in m] and not unexpected, (missing, unexpected)
    x = torch.randn(2, 3, img, img, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ref = hf(pixel_values=x).last_hidden_state
        got = vo.vit_features(p, x, arch, return_tokens=True)
    assert torch.allclose(got, ref, atol=2e-5), float((got - ref).abs().max())


def test_dino_head_matches_weight_norm_module():
    hp = vo.init_dino_head(192, 512, seed=2)
    mlp = nn.Sequential(nn.Linear(192, 2048), nn.GELU(), nn.Linear(2048, 2048), nn.GELU(), nn.Linear(2048, 256))
    last = nn.utils.weight_norm(nn.Linear(256, 512, bias=False))
    with torch.no_grad():
        for i in (0, 2, 4):
            mlp[i].weight.copy_(hp[f"mlp.{i}.weight"]); mlp[i].bias.copy_(hp[f"mlp.{i}.bias"])
        last.weight_g.copy_(hp["last_layer.weight_g"]); last.weight_v.copy_(hp["last_layer.weight_v"])
    x = torch.randn(5, 192, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ref = last(F.normalize(mlp(x), dim=-1, p=2))
    assert torch.allclose(vo.dino_head(hp, x), ref, atol=1e-5)


def test_softmax_lsce_matches_reference_formula():
    """train.py:1046 softmax then timm LabelSmoothingCrossEntropy with the gather patched to
    target[B,1] (train_instruct.txt:3-7): restated here independently with F.* calls."""
    z = torch.randn(9, 2, generator=torch.Generator().manual_seed(1))
    t = torch.randint(0, 2, (9, 1), generator=torch.Generator().manual_seed(2))
    out = F.softmax(z, dim=1)
    logp = F.log_softmax(out, dim=-1)
    ref = (0.9 * (-logp.gather(-1, t).squeeze(1)) + 0.1 * (-logp.mean(-1))).mean()
    assert torch.allclose(vo.softmax_lsce(z, t, 0.1), ref, atol=1e-7)


def test_dino_loss_matches_explicit_pairs():
    B, V, G, K = 3, 5, 2, 64
    g = torch.Generator().manual_seed(0)
    s, t, c = torch.randn(V * B, K, generator=g), torch.randn(G * B, K, generator=g), 0.1 * torch.randn(1, K, generator=g)
    loss, bsum = vo.dino_loss(s, t, c, V, G, 0.1, 0.04)
    tp = F.softmax((t - c) / 0.04, -1).view(G, B, K); ls = F.log_softmax(s / 0.1, -1).view(V, B, K)
    tot = sum(-(tp[iq] * ls[v]).sum(-1).mean() for iq in range(G) for v in range(V) if v != iq) / (G * (V - 1))
    assert torch.allclose(loss, tot, atol=1e-6) and torch.allclose(bsum[0], t.sum(0))


def test_interpolate_pos_encoding_matches_linear_map():
    """The engine resamples the pos-embed with a fixed matrix; it must equal the bicubic call."""
    import gipvit.engine as E
    pos = torch.randn(1, 197, 32, generator=torch.Generator().manual_seed(0))
    ref = vo.interpolate_pos_encoding(pos, 36, 96, 96)
    M = E.pos_interp_matrix(14, 96)
    got = M @ pos[0, 1:]
    assert torch.allclose(got, ref[0, 1:], atol=1e-5) and torch.equal(ref[0, 0], pos[0, 0])


def test_adamw_matches_torch():
    p0 = {"w": torch.randn(7, 5, generator=torch.Generator().manual_seed(0)), "b.bias": torch.randn(5, generator=torch.Generator().manual_seed(1))}
    mine = {k: v.clone() for k, v in p0.items()}
    opt = vo.AdamW(mine, 1e-2, 0.1)
    ref = [nn.Parameter(v.clone()) for v in p0.values()]
    topt = torch.optim.AdamW([{"params": [ref[0]], "weight_decay": 0.1}, {"params": [ref[1]], "weight_decay": 0.0}], lr=1e-2)
    for s in range(3):
        gs = {k: torch.randn_like(v, generator=None) * 0 + (s + 1) * 0.1 * torch.sign(v) for k, v in p0.items()}
        for r, g in zip(ref, gs.values()):
            r.grad = g.clone()
        topt.step(); opt.step(gs)
    assert torch.allclose(mine["w"], ref[0].detach(), atol=1e-6) and torch.allclose(mine["b.bias"], ref[1].detach(), atol=1e-6)


def test_golden_supervised_c1():
    """BASELINE config 1 (ViT-T/16, 64x64, supervised head, batch 8) reproduces its fixture."""
    gold = np.load(os.path.join(GOLD, "supervised_c1.npz"))
    orc = so.SupervisedOracle(arch="vit_tiny", img_size=64, num_classes=2, seed=0, lr=1e-4, wd=0.05)
    tiles = vo.synth_tiles(8, 64, seed=1234)
    tgt = torch.randint(0, 2, (8, 1), generator=torch.Generator().manual_seed(5))
    loss0, grads0, logits0 = orc.forward_backward(tiles, tgt)
    assert abs(float(loss0) - float(gold["loss0"])) < 1e-5
    assert np.allclose(logits0.numpy(), gold["logits0"], atol=1e-5)
    assert abs(so.grad_norm(grads0) - float(gold["grad_norm0"])) < 1e-4 * float(gold["grad_norm0"])
    curve = [orc.step(tiles, tgt)["loss"] for _ in range(5)]
    assert np.allclose(curve, gold["curve"][:5], atol=2e-4)


def test_golden_dino_tiny():
    gold = np.load(os.path.join(GOLD, "dino_tiny.npz"))
    orc = so.DinoOracle(arch="vit_tiny", img_size=224, out_dim=4096, seed=0)
    loss, grads, s_out, t_out, bsum = orc.forward_backward(vo.synth_tiles(2, 256, seed=1234))
    assert abs(float(loss) - float(gold["loss"])) < 1e-4
    assert np.allclose(s_out.numpy()[:, :64], gold["student"], atol=1e-4)
    assert np.allclose(grads["head.last_layer.weight_v"].numpy()[:4, :32], gold["g_last"], atol=1e-6, rtol=1e-3)


def test_crop_resize_oracle_vs_torch_interpolate():
    """oracle/augment_oracle.crop_resize restates torchvision's tensor-mode resized_crop (float32 bilinear,
    align_corners=False, no antialias, round, clamp); torch.nn.functional.interpolate pins it.  The two may
    differ by one count where the float result sits on a rounding boundary (association / FMA), nowhere else."""
    import torch.nn.functional as F
    from oracle import augment_oracle as ao
    rng = np.random.default_rng(0)
    tiles = rng.integers(0, 256, (3, 64, 64, 3), dtype=np.uint8)
    boxes = np.array([[0, 0, 0, 64, 64, 0], [1, 5, 9, 40, 31, 0], [2, 10, 3, 17, 50, 1], [0, 20, 20, 8, 8, 1], [1, 0, 0, 64, 64, 1]], np.int32)
    for out in (32, 96):
        got = ao.crop_resize(tiles, boxes, out)
        for n, (t, y0, x0, h, w, flip) in enumerate(boxes.tolist()):
            src = torch.from_numpy(tiles[t, y0:y0 + h, x0:x0 + w].copy()).permute(2, 0, 1)[None].float()
            ref = F.interpolate(src, size=(out, out), mode="bilinear", align_corners=False).round().clamp(0, 255)[0].permute(1, 2, 0).numpy()
            ref = ref[:, ::-1] if flip else ref
            d = np.abs(got[n].astype(np.int32) - ref.astype(np.int32))
            assert d.max() <= 1 and (d != 0).mean() < 5e-3, (out, n, d.max(), (d != 0).mean())   # ties only (torch contracts to FMA)
    # a box of the output's own size is a copy (and a mirrored copy)
    same = ao.crop_resize(tiles, np.array([[2, 7, 11, 32, 32, 0], [2, 7, 11, 32, 32, 1]], np.int32), 32)
    assert np.array_equal(same[0], tiles[2, 7:39, 11:43]) and np.array_equal(same[1], tiles[2, 7:39, 11:43][:, ::-1])


def test_multicrop_sampler_boxes():
    """Host sampler: boxes stay inside the tile, areas follow the scale ranges, rows are crop-major."""
    from gipvit.multicrop import MultiCropSampler
    sm = MultiCropSampler(batch=16, tile=256, seed=3)
    g, l = sm.sample()
    assert g.shape == (32, 6) and l.shape == (128, 6) and g.dtype == torch.int32
    for b, lo, hi in ((g, 0.4, 1.0), (l, 0.05, 0.4)):
        b = b.numpy()
        assert (b[:, 0] == np.tile(np.arange(16), len(b) // 16)).all()
        assert (b[:, 1] >= 0).all() and (b[:, 2] >= 0).all() and (b[:, 1] + b[:, 3] <= 256).all() and (b[:, 2] + b[:, 4] <= 256).all()
        frac = b[:, 3] * b[:, 4] / 65536.0
        assert frac.min() >= lo * 0.97 and frac.max() <= hi * 1.03
        assert 0.2 < b[:, 5].mean() < 0.8
    from oracle import augment_oracle as ao        # same algorithm, restated
    y0, x0, h, w = ao.sample_box(np.random.default_rng(1), 256, 256, (0.05, 0.4))
    assert 0 <= y0 and y0 + h <= 256 and 0 <= x0 and x0 + w <= 256


def test_weight_decay_sets_agree_name_by_name():
    """Oracle (AdamW restatement) and engine (arena layout) must decay exactly the same parameters -- timm's
    create_optimizer_v2 filter: 1-D tensors, *.bias, pos_embed, cls_token are excluded -- for the
    supervised model (bare names) and for the DINO student ('backbone.' / 'head.' prefixes)."""
    from gipvit import engine as E
    sup = so.SupervisedOracle(arch="vit_tiny", img_size=64, num_classes=2)
    specs = E.vit_param_specs("vit_tiny", 64, 2)
    assert list(specs) == list(sup.p)
    for n, t in sup.p.items():
        assert vo.no_weight_decay(n, t) == E.no_weight_decay(n, specs[n]), n
    d = so.DinoOracle(arch="vit_tiny", img_size=64, out_dim=256)
    from collections import OrderedDict
    specs = OrderedDict(("backbone." + k, v) for k, v in E.vit_param_specs("vit_tiny", 64, 0).items())
    specs.update(("head." + k, v) for k, v in E.dino_head_specs(192, 256).items())
    assert set(specs) == set(d.all)
    nd_o = {n for n, t in d.all.items() if vo.no_weight_decay(n, t)}
    nd_e = {n for n, sh in specs.items() if E.no_weight_decay(n, sh)}
    assert nd_o == nd_e, (nd_o ^ nd_e)
    assert {"backbone.pos_embed", "backbone.cls_token", "backbone.norm.weight", "head.mlp.0.bias", "head.last_layer.weight_g"} <= nd_o
    assert "backbone.patch_embed.proj.weight" not in nd_o and "head.last_layer.weight_v" not in nd_o


def test_golden_dino_curve_fixture_reproduces():
    """First steps of the 100-step DINO curve fixture (schedules, frozen last layer) re-run on the oracle."""
    from oracle.make_golden import dino_curve_schedule
    gold = np.load(os.path.join(GOLD, "dino_tiny_curve.npz"))
    assert len(gold["curve"]) == 100
    orc = so.DinoOracle(arch="vit_tiny", img_size=224, out_dim=4096, seed=0, clip_grad=3.0)
    for t in range(2):
        r = orc.step(vo.synth_tiles(8, 256, seed=5000 + t), **dino_curve_schedule(t, 100))
        assert abs(r["loss"] - gold["curve"][t]) < 2e-4 and abs(r["grad_norm"] - gold["grad_norm"][t]) < 1e-3 * gold["grad_norm"][t]
    s0, s99 = dino_curve_schedule(0), dino_curve_schedule(99)
    assert s0["lr"] == 1e-6 and not s0["train_last_layer"] and s0["teacher_temp"] == 0.04 and abs(s0["wd"] - 0.04) < 1e-12 and abs(s0["momentum_teacher"] - 0.996) < 1e-12
    assert s99["train_last_layer"] and s99["teacher_temp"] == 0.07 and abs(s99["wd"] - 0.4) < 1e-12 and abs(s99["momentum_teacher"] - 1.0) < 1e-12
    import math
    peak = 5e-4 * 8 / 256
    assert abs(dino_curve_schedule(24)["lr"] - (1e-6 + (peak - 1e-6) * 24 / 25)) < 1e-12                     # end of the linear warm-up
    assert abs(dino_curve_schedule(25)["lr"] - (1e-6 + 0.5 * (peak - 1e-6) * (1 + math.cos(math.pi / 4)))) < 1e-12   # timm cosine: t / epochs


# --------------------------------------------------------------------------- augmentation oracle vs PIL
def test_augment_oracle_colour_ops_match_pil():
    """oracle/augment_oracle.py restates the PIL operations torchvision's ColorJitter runs on PIL images
    (transformations.py:143-176): brightness / contrast / saturation byte-exact, HSV -> RGB exact over a dense grid,
    RGB -> HSV exact on S and V and on the H byte for > 99.5 % of the colours (+-1 otherwise)."""
    from PIL import Image, ImageEnhance
    from oracle import augment_oracle as ao
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (96, 96, 3), dtype=np.uint8)
    im = Image.fromarray(img)
    for f in (0.85, 1.0, 1.15, 0.3, 1.7):
        assert np.array_equal(ao.color_op(img, 0, {"bf": f}), np.asarray(ImageEnhance.Brightness(im).enhance(f)))
        assert np.array_equal(ao.color_op(img, 1, {"cf": f}), np.asarray(ImageEnhance.Contrast(im).enhance(f)))
        assert np.array_equal(ao.color_op(img, 2, {"sf": f}), np.asarray(ImageEnhance.Color(im).enhance(f)))
    grid = np.stack(np.meshgrid(np.arange(0, 256, 3), np.arange(256), np.arange(0, 256, 2), indexing="ij"), -1).reshape(-1, 256, 3).astype(np.uint8)
    assert np.array_equal(ao.hsv_to_rgb(grid), np.asarray(Image.fromarray(grid, mode="HSV").convert("RGB")))
    hsv_pil = np.asarray(Image.fromarray(grid).convert("HSV"))
    mine = ao.rgb_to_hsv(grid)
    assert np.array_equal(mine[..., 1:], hsv_pil[..., 1:])
    dh = np.abs(mine[..., 0].astype(int) - hsv_pil[..., 0].astype(int))
    dh = np.minimum(dh, 256 - dh)
    assert dh.max() <= 1 and (dh == 0).mean() > 0.995, (dh.max(), (dh == 0).mean())
    # torchvision's adjust_hue on a PIL image: +shift on the H byte with wrap-around
    for shift in (0, 10, 231):
        h, s_, v = im.convert("HSV").split()
        hh = (np.asarray(h).astype(np.int64) + shift) & 255
        ref = np.asarray(Image.merge("HSV", (Image.fromarray(hh.astype(np.uint8), "L"), s_, v)).convert("RGB"))
        got = ao.color_op(img, 3, {"hue": shift})
        d = np.abs(got.astype(int) - ref.astype(int))
        assert (d.max(-1) == 0).mean() > 0.99 and d.max() <= 6, ((d.max(-1) == 0).mean(), d.max())


def test_view_augment_oracle_matches_pil_and_sampler_statistics():
    """DINO view augmentation (oracle/augment_oracle.py view_augment, the checker of gv_crop_augment): grayscale == PIL
    convert("L") replicated, solarise == PIL ImageOps.solarize, a whole chain (brightness -> contrast -> saturation ->
    grayscale -> solarise) == the same chain run through PIL's own modules; the host sampler draws DINO's probabilities."""
    from PIL import Image, ImageEnhance, ImageOps
    from oracle import augment_oracle as ao
    from gipvit.multicrop import ViewAugmentSampler
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (96, 96, 3), dtype=np.uint8)
    im = Image.fromarray(img)
    assert np.array_equal(ao.to_gray(img), np.asarray(im.convert("L").convert("RGB")))
    for th in (128, 0, 255, 77):
        assert np.array_equal(ao.solarize(img, th), np.asarray(ImageOps.solarize(im, th)))
    p = dict(order=[0, 1, 2], bf=1.3, cf=0.7, sf=1.15, hue=0, gray=False, blur=None, solar=128)
    ref = ImageEnhance.Color(ImageEnhance.Contrast(ImageEnhance.Brightness(im).enhance(1.3)).enhance(0.7)).enhance(1.15)
    assert np.array_equal(ao.view_augment(img, p), np.asarray(ImageOps.solarize(ref, 128)))
    p = dict(order=[2, 0], bf=0.8, sf=0.9, gray=True, blur=None, solar=-1)
    ref = ImageEnhance.Brightness(ImageEnhance.Color(im).enhance(0.9)).enhance(0.8).convert("L").convert("RGB")
    assert np.array_equal(ao.view_augment(img, p), np.asarray(ref))
    # blur runs on the jittered / grey view, before the solarisation
    p = dict(order=[], gray=True, blur=(0.6, 0.2), solar=100)
    assert np.array_equal(ao.view_augment(img, p), ao.solarize(ao.blur3(ao.to_gray(img), 0.6, 0.2), 100))
    # the sampler: DINO's probabilities per crop kind (jitter 0.8, grey 0.2, blur 1.0 / 0.1 / 0.5, solarise 0 / 0.2 / 0)
    sp = ViewAugmentSampler(batch=4000, n_global=2, n_local=2, seed=1)
    g, l = sp.sample_host()
    g1, g2 = g[:4000], g[4000:]
    assert abs((g["n_color"] == 4).mean() - 0.8) < 0.03 and abs(g["gray"].mean() - 0.2) < 0.03
    assert g1["blur"].all() and abs(g2["blur"].mean() - 0.1) < 0.03 and abs(l["blur"].mean() - 0.5) < 0.03
    assert (g1["solar"] == -1).all() and abs((g2["solar"] == 128).mean() - 0.2) < 0.03 and (l["solar"] == -1).all()
    assert 0.6 <= g["bf"].min() and g["bf"].max() <= 1.4 and 0.8 <= g["sf"].min() and g["sf"].max() <= 1.2
    assert np.allclose(g["kc"] + 2 * g["ks"], 1.0, atol=1e-6)
    d = ViewAugmentSampler.to_dicts(g[:3])
    assert set(d[0]) == {"order", "bf", "cf", "sf", "hue", "gray", "blur", "solar"}
    import ctypes
    from gipvit import _lib
    assert ViewAugmentSampler.DT.itemsize == ctypes.sizeof(_lib.gv_view_params) == 56


def test_augment_oracle_geometry_and_host_sampler():
    """NEAREST zoom == PIL's affine transform (torchvision RandomAffine(degrees=0, scale) on PIL images); the dihedral
    composition and the blur weights of the host sampler; recipes draw what the reference's Compose would."""
    from PIL import Image
    from oracle import augment_oracle as ao
    from gipvit import augment as A
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    im = Image.fromarray(img)
    for s in (1.0, 1.1234, 1.2, 1.19999, 1.05):
        a = 1.0 / s
        m = [a, 0.0, a * -128.0 + 128.0, 0.0, a, a * -128.0 + 128.0]
        ref = np.asarray(im.transform((256, 256), Image.AFFINE, m, Image.NEAREST))
        got = ao.zoom(img, *ao.zoom_fixed(s, 256))
        assert (got == ref).all(-1).mean() > 0.95, s                      # identical except where the float matrix sits on a pixel edge
        assert A.zoom_fixed(s, 256) == ao.zoom_fixed(s, 256)
    assert np.array_equal(ao.zoom(img, *ao.zoom_fixed(1.0, 256)), img)
    # dihedral composition: every sequence of flips / rotations is reproduced by the 3-bit element
    small = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
    for seq in ([], ["v"], ["h"], ["r1"], ["r2"], ["r3"], ["v", "r1"], ["r1", "v"], ["h", "r3"], ["v", "h"], ["r2", "v"], ["v", "r3", "h"]):
        code = A.compose_d4(seq)
        ys, xs = np.mgrid[0:8, 0:8]
        u, v = (xs, ys) if code & 1 else (ys, xs)
        u = 7 - u if code & 2 else u
        v = 7 - v if code & 4 else v
        assert np.array_equal(small[u, v], ao.augment_tile(small, {"geo": seq}, None)), seq
    # blur weights = torchvision's _get_gaussian_kernel1d(3, sigma) (float32)
    for sigma in (0.05, 0.5, 1.0):
        x = torch.linspace(-1, 1, 3)
        k = torch.exp(-0.5 * (x / sigma) ** 2); k = k / k.sum()
        kc, ks = A.blur_weights(sigma)
        assert abs(kc - float(k[1])) < 1e-7 and abs(ks - float(k[0])) < 1e-7
    assert A.blur_weights(0.1)[1] < 1e-20                                    # the recipes' sigma <= 0.1: the blur is the identity
    z = A.normal_table().numpy()
    assert z.shape == (1024,) and abs(z.mean()) < 1e-6 and abs(z.std() - 1.0) < 5e-3 and np.all(np.diff(z) > 0)
    aug = A.TileAugmenter("pcbnfrsc", 256, color_param=0.1, seed=0)
    ps = [aug.sample_one() for _ in range(200)]
    assert all(sorted(p["order"]) == [0, 1, 2, 3] and 0.9 <= p["bf"] <= 1.1 and 0.8 <= p["cf"] <= 1.2 and 0 <= p["sigma"] <= 0.05 for p in ps)
    assert all(p["blur"] is None and p["fill"][4:] == (0.0, 0.0, 0.0) and p["fill"][1] - p["fill"][0] <= 100 for p in ps)
    assert 0.3 < np.mean(["v" in p["geo"] for p in ps]) < 0.7 and {g for p in ps for g in p["geo"] if g[0] == "r"} == {"r0", "r1", "r2", "r3"}
    assert all(65536 / 1.2 - 1 <= p["zoom"][0] <= 65536 for p in ps)
    rp = A.TileAugmenter("aug_receptornet", 256, seed=1)
    qs = [rp.sample_one() for _ in range(200)]
    assert 0.6 < np.mean([q["fill"] is not None for q in qs]) < 0.9 and all("cut" in q for q in qs) and all(not q.get("zoom") for q in qs)
    assert A.TileAugmenter("rvf", 64, seed=0).sample_one().keys() >= {"geo", "fill"}
    packed = A.TileAugmenter.pack(ps[:3])
    assert packed.dtype == np.uint8 and packed.size == 3 * 88
    # the vectorised path the training loop uses packs exactly the records of the per-tile (oracle-format) draws
    for rec in A.RECIPES:
        au = A.TileAugmenter(rec, 256, 0.15, seed=3)
        cols = au.sample_batch(32)
        pk, fill = au.pack_batch(cols)
        dicts = au.to_dicts(cols)
        assert np.array_equal(pk, A.TileAugmenter.pack(dicts)), rec
        assert (fill is not None) == any(d["fill"] for d in dicts)
        if fill is not None:
            for i, d in enumerate(dicts):
                assert fill[i, 7] == (1.0 if d["fill"] else 0.0) and (not d["fill"] or np.allclose(fill[i, :7], d["fill"]))


def test_drop_path_oracle_matches_stochastic_depth_module():
    """vo.block with explicit factors == x + DropPath(branch) written with torch modules semantics (timm drop_path: keep mask
    / keep_prob per sample), and drop_path_factors follows the linspace rule (block 0 never drops; expectation 1)."""
    from oracle import vit_oracle as vo
    torch.manual_seed(0)
    p = vo.init_vit("vit_tiny", 64, 0, seed=3)
    x = torch.randn(5, 17, 192)
    f = torch.tensor([[1.0, 0.0, 2.0, 0.0, 2.0], [0.0, 1.25, 1.25, 0.0, 1.25]])
    y = vo.block(x, p, 4, 3, drop=f)
    b = "blocks.4."
    a = vo.attention(vo.layer_norm(x, p[b + "norm1.weight"], p[b + "norm1.bias"]), p, b + "attn.", 3)
    x1 = x + a * f[0].view(5, 1, 1)
    h = vo.gelu(vo.layer_norm(x1, p[b + "norm2.weight"], p[b + "norm2.bias"]) @ p[b + "mlp.fc1.weight"].t() + p[b + "mlp.fc1.bias"])
    ref = x1 + (h @ p[b + "mlp.fc2.weight"].t() + p[b + "mlp.fc2.bias"]) * f[1].view(5, 1, 1)
    assert torch.allclose(y, ref, atol=1e-6)
    assert torch.equal(y[3], x[3])                                   # both branches dropped: the sample passes through
    d = vo.drop_path_factors(12, 4000, 0.2, torch.Generator().manual_seed(1))
    assert float(d[0].min()) == 1.0 and float(d[0].max()) == 1.0
    keep_last = 1.0 - 0.2
    assert set(torch.unique(d[11]).tolist()) == {0.0, 1.0 / keep_last}
    assert abs(float(d[11].mean()) - 1.0) < 0.03 and abs(float((d[6] > 0).float().mean()) - (1 - 0.2 * 6 / 11)) < 0.03
