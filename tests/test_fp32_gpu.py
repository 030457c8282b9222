"""fp32 operand mode (csrc/f32path.hip + the f32-I/O instantiations of LayerNorm / patchify / token scatter) against the
fp32 CPU oracle.  The reference's default arithmetic is fp32 (train.py without --amp), and SURVEY 8d states the
fp32-column gates this file holds:
    logits max-abs err <= 1e-4,  loss |d| <= 1e-4,  global gradient-norm rel err <= 1e-3
(per-parameter gradients are additionally held to ||g - ref|| / ||ref|| <= 1e-3).  Kernel-level checks compare against
torch float64 on the CPU."""
import math
import os

import numpy as np
import pytest
import torch

gpu = pytest.mark.gpu
f32 = torch.float32

LOGIT_TOL, LOSS_TOL, GNORM_TOL, GRAD_TOL = 1e-4, 1e-4, 1e-3, 1e-3


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def _dgelu(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


@gpu
@pytest.mark.parametrize("M,N,K", [(136, 192, 768), (136, 576, 192), (300, 130, 70), (17, 5, 3), (394, 384, 1536)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True), (True, False)])
def test_linear_f32_all_layouts(dev, M, N, K, ta, tb):
    """C = op(A) op(B) for every operand layout, ragged M / N / K included (edges are zero-filled while staging)."""
    from gipvit import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((K, N) if tb else (N, K), generator=g)
    ref = (A.double().t() if ta else A.double()) @ (B.double() if tb else B.double().t())
    C = torch.empty(M, N, device=dev)
    ops.linear(A.to(dev), B.to(dev), C, M, N, K, trans_a=ta, trans_b=tb)
    assert _rel(C, ref) < 2e-6, _rel(C, ref)


@gpu
def test_linear_f32_epilogues(dev):
    from gipvit import ops, _lib as L
    g = torch.Generator().manual_seed(3)
    M, N, K = 150, 200, 96
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    bias, resid, rs = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.rand(M, generator=g) * 2
    pre_ref = A.double() @ W.double().t() + bias.double()
    d = lambda t: t.to(dev)
    # fc1: BIAS | GELU | SAVE_PRE
    C, pre = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.linear(d(A), d(W), C, M, N, K, epilogue=L.EPI_BIAS | L.EPI_GELU | L.EPI_SAVE_PRE, bias=d(bias), aux_out=pre)
    assert _rel(pre, pre_ref) < 2e-6 and _rel(C, _gelu(pre_ref)) < 2e-6
    # proj / fc2: BIAS | RESID with stochastic-depth row factors
    ops.linear(d(A), d(W), C, M, N, K, epilogue=L.EPI_BIAS | L.EPI_RESID, bias=d(bias), resid=d(resid), row_scale=d(rs))
    assert _rel(C, pre_ref * rs.double()[:, None] + resid.double()) < 2e-6
    # dX of fc2 through GELU': DGELU on the saved pre-activation
    aux = torch.randn(M, N, generator=g)
    ops.linear(d(A), d(W), C, M, N, K, epilogue=L.EPI_DGELU, aux_in=d(aux))
    assert _rel(C, (A.double() @ W.double().t()) * _dgelu(aux.double())) < 2e-6
    # dW: ACCUM + the bias gradient from the staged A tiles
    dY, X = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    C0, cs0 = torch.randn(M, N, generator=g), torch.randn(M, generator=g)
    Cg, cs = d(C0).clone(), d(cs0).clone()
    ops.linear(d(dY), d(X), Cg, M, N, K, trans_a=True, trans_b=True, epilogue=L.EPI_ACCUM, colsum_a=cs)
    assert _rel(Cg, C0.double() + dY.double().t() @ X.double()) < 2e-6
    assert _rel(cs, cs0.double() + dY.double().sum(0)) < 2e-6
    # ... and over many token rows, where the launch splits K and the partial tiles meet in C through f32 atomics
    Kt, Md, Nd = 5000, 192, 390
    dY, X = torch.randn(Kt, Md, generator=g), torch.randn(Kt, Nd, generator=g)
    C0, cs0 = torch.randn(Md, Nd, generator=g), torch.randn(Md, generator=g)
    Cg, cs = d(C0).clone(), d(cs0).clone()
    ops.linear(d(dY), d(X), Cg, Md, Nd, Kt, trans_a=True, trans_b=True, epilogue=L.EPI_ACCUM, colsum_a=cs)
    assert _rel(Cg, C0.double() + dY.double().t() @ X.double()) < 2e-6
    assert _rel(cs, cs0.double() + dY.double().sum(0)) < 2e-6
    # patch embedding: BIAS | POS (token-row remap past the CLS rows)
    n_img, P, Dm = 3, 16, 192
    patches, Wp, bp, pos = torch.randn(n_img * P, 768, generator=g), torch.randn(Dm, 768, generator=g) * 0.05, torch.randn(Dm, generator=g), torch.randn(P + 1, Dm, generator=g)
    x = torch.zeros(n_img * (P + 1), Dm, device=dev)
    ops.linear(d(patches), d(Wp), x, n_img * P, Dm, 768, epilogue=L.EPI_BIAS | L.EPI_POS, bias=d(bp), pos=d(pos), P=P)
    ref = (patches.double() @ Wp.double().t() + bp.double()).view(n_img, P, Dm) + pos.double()[1:]
    got = x.view(n_img, P + 1, Dm)
    assert _rel(got[:, 1:], ref) < 2e-6 and float(got[:, 0].abs().max()) == 0.0
    # mixed operand types are refused, not silently converted
    with pytest.raises(TypeError):
        ops.linear(d(A), d(W).bfloat16(), C, M, N, K)


@gpu
@pytest.mark.parametrize("n_img,N,H", [(3, 17, 3), (2, 197, 6), (2, 257, 12), (5, 64, 1), (1, 65, 2)])
def test_attention_f32(dev, n_img, N, H):
    from gipvit import ops
    g = torch.Generator().manual_seed(N)
    D = H * 64
    qkv = torch.randn(n_img * N, 3 * D, generator=g)
    d_o = torch.randn(n_img * N, D, generator=g)
    scale = 64 ** -0.5
    x = qkv.double().view(n_img, N, 3, H, 64).permute(2, 0, 3, 1, 4).requires_grad_(True)
    q, k, v = x[0], x[1], x[2]
    s = (q @ k.transpose(-1, -2)) * scale
    o_ref = (s.softmax(-1) @ v).transpose(1, 2).reshape(n_img * N, D)
    o_ref.backward(d_o.double())
    dqkv_ref = x.grad.permute(1, 3, 0, 2, 4).reshape(n_img * N, 3 * D)
    o, lse = ops.attention_fwd(qkv.to(dev), n_img, N, H, scale)
    assert o.dtype == f32
    assert _rel(o, o_ref.detach()) < 2e-6
    assert float((lse.cpu().double() - torch.logsumexp(s.detach(), -1)).abs().max()) < 1e-5
    dqkv = ops.attention_bwd(qkv.to(dev), o, d_o.to(dev), lse, n_img, N, H, scale)
    assert _rel(dqkv, dqkv_ref) < 5e-6, _rel(dqkv, dqkv_ref)


@gpu
@pytest.mark.parametrize("D", [192, 384, 768])
def test_layernorm_f32_io(dev, D):
    from gipvit import ops, _lib as L
    g = torch.Generator().manual_seed(D)
    rows = 333
    x, gamma, beta = torch.randn(rows, D, generator=g) * 2 + 0.3, torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g)
    dy, g_in = torch.randn(rows, D, generator=g), torch.randn(rows, D, generator=g)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    y_ref.backward(dy.double())
    y = torch.empty(rows, D, device=dev)
    _, mean, rstd = ops.layernorm_fwd(x.to(dev), gamma.to(dev), beta.to(dev), rows, D, y=y)
    assert _rel(y, y_ref.detach()) < 2e-6
    gbuf, gb = g_in.to(dev).clone(), torch.empty(rows, D, device=dev)
    partials = torch.zeros(L.LN_PARTIAL_BLOCKS, 3, D, device=dev)
    ops.layernorm_bwd(dy.to(dev), x.to(dev), mean, rstd, gamma.to(dev), gbuf, gb, partials, rows, D)
    assert _rel(gbuf, g_in.double() + xr.grad) < 2e-6
    assert torch.equal(gb, gbuf)                       # the f32 "copy for the next GEMM" is the residual gradient itself
    dgamma, dbeta = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    ops.ln_finalize(partials, L.LN_PARTIAL_BLOCKS, D, dgamma, dbeta, None)
    assert _rel(dgamma, gr.grad) < 5e-6 and _rel(dbeta, br.grad) < 5e-6


def _gates(eng, loss_r, grads_r, logits_r):
    torch.cuda.synchronize()
    dl = float((eng.logits.cpu() - logits_r).abs().max())
    assert dl <= LOGIT_TOL, f"logits max-abs err {dl}"
    assert abs(float(eng.loss) - float(loss_r)) <= LOSS_TOL, (float(eng.loss), float(loss_r))
    got = eng.grads()
    gn_g = math.sqrt(sum(float((got[k].double() ** 2).sum()) for k in grads_r))
    gn_r = math.sqrt(sum(float((r.double() ** 2).sum()) for r in grads_r.values()))
    assert abs(gn_g - gn_r) <= GNORM_TOL * gn_r, (gn_g, gn_r)
    worst = max((_rel(got[k], r), k) for k, r in grads_r.items() if float(r.abs().max()) > 1e-12)
    assert worst[0] <= GRAD_TOL, f"gradient mismatch {worst}"
    return dl, abs(gn_g - gn_r) / gn_r, worst


@gpu
def test_fp32_supervised_config1_gates(dev):
    """BASELINE config 1 (ViT-T/16, 64 x 64 tiles, batch 8, supervised head) in the fp32 operand mode: the fp32-column gates,
    then 20 AdamW steps at the reference's learning rate against the oracle's trajectory at 1e-4."""
    from gipvit.engine import SupervisedEngine
    from oracle import step_oracle as so, vit_oracle as vo
    orc = so.SupervisedOracle(arch="vit_tiny", img_size=64, num_classes=2, seed=0, lr=1e-3, wd=0.05)
    eng = SupervisedEngine(arch="vit_tiny", img_size=64, num_classes=2, batch=8, lr=1e-3, weight_decay=0.05, device=dev, precision="fp32")
    assert eng.grp.qkv[0].dtype == f32 and eng.feats.dtype == f32 and eng.W.w("blocks.0.attn.qkv.weight").dtype == f32
    eng.load_state(orc.p)
    tiles = vo.synth_tiles(8, 64, seed=1234)
    tgt = torch.randint(0, 2, (8, 1), generator=torch.Generator().manual_seed(5))
    loss_r, grads_r, logits_r = orc.forward_backward(tiles, tgt)
    eng.forward_backward(tiles.to(dev), tgt.to(dev))
    dl, dgn, worst = _gates(eng, loss_r, grads_r, logits_r)
    print(f"[fp32 config 1] logits {dl:.2e}, grad-norm rel {dgn:.2e}, worst parameter gradient {worst[0]:.2e} ({worst[1]})")
    for i in range(20):
        r = orc.step(tiles, tgt)
        l = eng.step(tiles.to(dev), tgt.to(dev))
        assert abs(float(l) - r["loss"]) <= 1e-4, (i, float(l), r["loss"])


@gpu
def test_fp32_golden_supervised(dev):
    """... and against the committed fixture (tests/golden/supervised_c1.npz): step-0 logits / loss / gradient norm at the
    fp32 gates, the 100-step AdamW curve at 1e-4 (north_star states 1e-3 for it)."""
    from gipvit.engine import SupervisedEngine
    from oracle import vit_oracle as vo
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "supervised_c1.npz"))
    eng = SupervisedEngine(arch="vit_tiny", img_size=64, num_classes=2, batch=8, lr=1e-4, weight_decay=0.05, device=dev, precision="fp32")
    eng.load_state(vo.init_vit("vit_tiny", 64, 2, seed=0))
    tiles = vo.synth_tiles(8, 64, seed=1234).to(dev)
    tgt = torch.randint(0, 2, (8, 1), generator=torch.Generator().manual_seed(5)).to(dev)
    eng.forward_backward(tiles, tgt)
    torch.cuda.synchronize()
    assert abs(float(eng.loss) - float(gold["loss0"])) <= LOSS_TOL
    assert float(np.abs(eng.logits.cpu().numpy() - gold["logits0"]).max()) <= LOGIT_TOL
    gn = math.sqrt(sum(float((g.double() ** 2).sum()) for g in eng.grads().values()))
    assert abs(gn - float(gold["grad_norm0"])) <= GNORM_TOL * float(gold["grad_norm0"])
    assert _rel(eng.grads()["head.weight"], torch.from_numpy(gold["g_head"])) <= GRAD_TOL
    curve = [float(eng.step(tiles, tgt)) for _ in range(len(gold["curve"]))]
    err = np.abs(np.array(curve) - gold["curve"])
    print(f"[fp32 golden curve] {len(curve)} steps, max |dloss| {err.max():.2e} at step {int(err.argmax())}")
    assert float(err.max()) <= 1e-4, (float(err.max()), int(err.argmax()))


@gpu
@pytest.mark.parametrize("arch,img,B", [("vit_small", 224, 2), ("vit_base", 64, 2)])
def test_fp32_supervised_wider_archs(dev, arch, img, B):
    """ViT-S at 224 px (197 tokens: the four-chunk attention path, D = 384 without the fused LayerNorm kernels) and the
    ViT-B width, with stochastic depth on: same gates."""
    from gipvit.engine import SupervisedEngine
    from oracle import step_oracle as so, vit_oracle as vo
    orc = so.SupervisedOracle(arch=arch, img_size=img, num_classes=2, seed=1, lr=1e-3, wd=0.05)
    eng = SupervisedEngine(arch=arch, img_size=img, num_classes=2, batch=B, lr=1e-3, weight_decay=0.05, device=dev, precision="fp32")
    eng.load_state(orc.p)
    tiles = vo.synth_tiles(B, img, seed=77)
    tgt = torch.randint(0, 2, (B, 1), generator=torch.Generator().manual_seed(6))
    drop = vo.drop_path_factors(12, B, 0.3, torch.Generator().manual_seed(2))
    loss_r, grads_r, logits_r = orc.forward_backward(tiles, tgt, drop)
    eng.set_drop_path(drop.to(dev))
    eng.forward_backward(tiles.to(dev), tgt.to(dev))
    dl, dgn, worst = _gates(eng, loss_r, grads_r, logits_r)
    print(f"[fp32 {arch}] logits {dl:.2e}, grad-norm rel {dgn:.2e}, worst parameter gradient {worst[0]:.2e} ({worst[1]})")


@gpu
@pytest.mark.parametrize("arch,n_local,K", [("vit_tiny", 8, 4096), ("vit_tiny", 0, 4096), ("vit_small", 8, 2048)])
def test_fp32_dino_step_gates(dev, arch, n_local, K):
    """The DINO multi-crop step (2 x 224 global + 8 x 96 local crops of 256-px tiles, B = 2; ViT-T and the ViT-S of the
    headline config) in the fp32 operand mode: teacher and student head outputs, loss, centre sum and every gradient at
    the fp32 gates, then three optimizer + EMA + centre steps on the oracle's trajectory at 1e-4."""
    from gipvit.engine import DinoEngine
    from oracle import step_oracle as so
    from oracle import vit_oracle as vo
    B = 2
    orc = so.DinoOracle(arch=arch, img_size=224, out_dim=K, seed=0, lr=5e-4, wd=0.04, n_local=n_local)
    eng = DinoEngine(arch=arch, img_size=224, out_dim=K, batch=B, n_local=n_local, lr=5e-4, weight_decay=0.04, device=dev, precision="fp32")
    assert eng.hb_s.dlogits.dtype == f32 and eng.wn_s.dtype == f32 and eng.g_stu.qkv[0].dtype == f32
    eng.load_state(orc.p, orc.hp)
    c = 0.05 * torch.randn(1, K, generator=torch.Generator().manual_seed(3))
    orc.center = c.clone(); eng.center.copy_(c[0])
    tiles = vo.synth_tiles(B, 256, seed=1234)
    loss_r, grads_r, s_out, t_out, bsum = orc.forward_backward(tiles)
    eng.set_hyper()
    eng.forward_backward(tiles.to(dev))
    torch.cuda.synchronize()
    for got, ref, nm in ((eng.hb_t.logits, t_out, "teacher"), (eng.hb_s.logits, s_out, "student")):
        err = float((got.cpu() - ref).abs().max())
        assert err <= LOGIT_TOL * max(1.0, float(ref.abs().max())), (nm, err, float(ref.abs().max()))
    assert abs(float(eng.loss) - float(loss_r)) <= LOSS_TOL, (float(eng.loss), float(loss_r))
    assert _rel(eng.center_sum, bsum[0]) < 1e-5
    got = eng.grads()
    keys = [k for k, r in grads_r.items() if r is not None and k != "head.last_layer.weight_g"]
    gn_g = math.sqrt(sum(float((got[k].double() ** 2).sum()) for k in keys))
    gn_r = math.sqrt(sum(float((grads_r[k].double() ** 2).sum()) for k in keys))
    assert abs(gn_g - gn_r) <= GNORM_TOL * gn_r, (gn_g, gn_r)
    worst = max((_rel(got[k], grads_r[k]), k) for k in keys if float(grads_r[k].abs().max()) > 1e-12)
    assert worst[0] <= GRAD_TOL, f"gradient mismatch {worst}"
    print(f"[fp32 dino {arch} n_local={n_local}] loss {abs(float(eng.loss) - float(loss_r)):.2e}, grad-norm rel {abs(gn_g - gn_r) / gn_r:.2e}, "
          f"worst parameter gradient {worst[0]:.2e} ({worst[1]})")
    eng.t = 0
    for i in range(3):
        r = orc.step(tiles)
        l = eng.step(tiles.to(dev))
        assert abs(float(l) - r["loss"]) <= 1e-4, (i, float(l), r["loss"])
    torch.cuda.synchronize()
    assert _rel(eng.center, orc.center[0]) < 1e-4
    sd, td = eng.backbone_state_dict(), eng.backbone_state_dict(teacher=True)
    for k in ("blocks.0.attn.qkv.weight", "blocks.11.mlp.fc2.weight", "pos_embed", "norm.weight"):
        assert _rel(sd[k], orc.p[k]) < 1e-4, (k, _rel(sd[k], orc.p[k]))
        assert _rel(td[k], orc.tp[k]) < 1e-5, k


@gpu
@pytest.mark.parametrize("arch,D,steps", [("vit_small", 384, 10), ("vit_base", 768, 0)])
def test_headline_config_bf16_step_against_fp32_mode(dev, arch, D, steps):
    """BASELINE config 3 at its FULL size (ViT-S/16, 64 tiles of 256 px, 2 x 224 + 8 x 96 crops, K = 65536, stochastic
    depth 0.1) -- a size the CPU oracle cannot reach in test time.  The fp32 operand mode, pinned to the oracle at the
    fp32 gates by the tests above, is the reference here: the bf16 training path (full-row fused kernels, grouped dW, 251
    workgroups per launch -- none of which the small-shape tests reach at this row count) must agree with it at the bf16
    gates: logits 2e-2 of the largest, loss 1e-3, per-parameter gradient 5e-2, gradient norm 1e-2; ten optimizer steps stay
    within 1e-3 of the fp32 loss curve.  The ViT-B case is config 5's per-micro-batch shape (64 tiles; the 128 x 128-tile
    GEMMs and the stand-alone LayerNorm kernels at 44 160 token rows)."""
    from gipvit.engine import DinoEngine
    from gipvit.models import init_vit_state, init_dino_head_state
    from oracle import vit_oracle as vo
    B, K = 64, 65536
    mk = lambda prec: DinoEngine(arch=arch, img_size=224, out_dim=K, batch=B, n_local=8, lr=5e-4 * B / 256, weight_decay=0.04,
                                 clip_grad=3.0, device=dev, precision=prec)
    bb, hd = init_vit_state(arch, 224, 0, seed=0), init_dino_head_state(D, K, seed=1)
    tiles = vo.synth_tiles(B, 256, seed=99).to(dev)
    drop = vo.drop_path_factors(12, 10 * B, 0.1, torch.Generator().manual_seed(4)).to(dev)
    c = (0.05 * torch.randn(K, generator=torch.Generator().manual_seed(3))).to(dev)
    out = {}
    for prec in ("fp32", "bf16"):
        eng = mk(prec)
        eng.load_state(bb, hd)
        eng.center.copy_(c)
        eng.set_drop_path(drop)
        eng.set_hyper()
        eng.forward_backward(tiles)
        torch.cuda.synchronize()
        out[prec] = dict(loss=float(eng.loss), s=eng.hb_s.logits.float().cpu(), t=eng.hb_t.logits.float().cpu(),
                         cs=eng.center_sum.cpu(), g={k: v.cpu() for k, v in eng.grads().items()})
        # ... then ten full steps (optimizer, teacher EMA, centre, gradient clipping) at a warm-up learning rate
        eng.t = 0
        out[prec]["curve"] = [float(eng.step(tiles, lr=1e-5)) for _ in range(steps)]
        out[prec]["w"] = eng.backbone_state_dict()["blocks.11.mlp.fc2.weight"].cpu()
        del eng
        torch.cuda.empty_cache()
    r, b = out["fp32"], out["bf16"]
    assert math.isfinite(r["loss"]) and 10.0 < r["loss"] < 12.0          # ln(65536) = 11.09 at initialisation
    for nm in ("s", "t"):
        err, top = float((b[nm] - r[nm]).abs().max()), float(r[nm].abs().max())
        assert err <= 2e-2 * top, (nm, err, top)
    assert abs(b["loss"] - r["loss"]) <= 1e-3, (b["loss"], r["loss"])
    assert _rel(b["cs"], r["cs"]) < 1e-2
    keys = [k for k in r["g"] if k != "head.last_layer.weight_g" and float(r["g"][k].abs().max()) > 1e-12]
    worst = max((_rel(b["g"][k], r["g"][k]), k) for k in keys)
    gn_b = math.sqrt(sum(float((b["g"][k].double() ** 2).sum()) for k in keys))
    gn_r = math.sqrt(sum(float((r["g"][k].double() ** 2).sum()) for k in keys))
    print(f"[{arch}, full size] loss bf16 {b['loss']:.5f} vs fp32 {r['loss']:.5f}, worst parameter gradient {worst[0]:.2e} ({worst[1]}), "
          f"grad-norm rel {abs(gn_b - gn_r) / gn_r:.2e}")
    assert worst[0] <= 5e-2, worst
    assert abs(gn_b - gn_r) <= 1e-2 * gn_r, (gn_b, gn_r)
    if not steps:
        return
    dcurve = max(abs(x - y) for x, y in zip(b["curve"], r["curve"]))
    print(f"[{arch}, full size] {steps} steps: max |dloss| {dcurve:.2e}, final {b['curve'][-1]:.5f} vs {r['curve'][-1]:.5f}, "
          f"weights rel {_rel(b['w'], r['w']):.2e}")
    assert dcurve <= 1e-3, (b["curve"], r["curve"])
    assert _rel(b["w"], r["w"]) < 1e-3


@gpu
def test_fp32_feature_extractor_and_model_seam(dev):
    """The forward-only encoder (slide validation / --extract_features) and the create_model seam in the fp32 operand mode:
    CLS features and logits against the oracle's forward at 1e-5 relative (2e-2 / 3e-2 in bf16), with any number of tiles
    through the padded last batch."""
    from gipvit.engine import FeatureExtractor
    from gipvit import models
    from oracle import vit_oracle as vo
    p = vo.init_vit("vit_small", 64, 2, seed=4)
    fx = FeatureExtractor(arch="vit_small", img_size=64, batch=6, num_classes=2, device=dev, precision="fp32")
    assert fx.feats.dtype == f32
    fx.load_state(p)
    tiles = vo.synth_tiles(10, 64, seed=21)
    feats, logits = fx.run(tiles.to(dev))                       # 6 + 4 (padded)
    torch.cuda.synchronize()
    x = vo.normalize_window(tiles, (0, 0, 64))
    ref_f, ref_l = vo.vit_features(p, x, "vit_small"), vo.vit_logits(p, x, "vit_small")
    assert _rel(feats, ref_f) < 1e-5 and _rel(logits, ref_l) < 1e-5, (_rel(feats, ref_f), _rel(logits, ref_l))
    m = models.create_model("vit_small_patch16_224", num_classes=2, img_size=64, batch=10, device=dev, precision="fp32")
    m.load_state_dict(p)
    out = m(tiles.to(dev))
    assert out.dtype == f32 and _rel(out, ref_l) < 1e-5


@gpu
def test_lamb_supervised_steps_match_oracle_fp32(dev):
    """--opt lamb end to end (config 1's model, fp32 operand mode so that no bf16 noise sits between the two optimizers): five
    steps of SupervisedEngine(opt="lamb") against the oracle's Lamb restatement -- losses within 1e-4, weights within 1e-4."""
    from gipvit.engine import SupervisedEngine
    from oracle import step_oracle as so, vit_oracle as vo
    orc = so.SupervisedOracle(arch="vit_tiny", img_size=64, num_classes=2, seed=0, lr=5e-3, wd=0.05, opt="lamb")
    eng = SupervisedEngine(arch="vit_tiny", img_size=64, num_classes=2, batch=8, lr=5e-3, weight_decay=0.05, eps=1e-6, opt="lamb", device=dev,
                           precision="fp32")
    eng.load_state(orc.p)
    tiles = vo.synth_tiles(8, 64, seed=1234)
    tgt = torch.randint(0, 2, (8, 1), generator=torch.Generator().manual_seed(5))
    for i in range(5):
        r = orc.step(tiles, tgt)
        l = eng.step(tiles.to(dev), tgt.to(dev))
        assert abs(float(l) - r["loss"]) <= 1e-4, (i, float(l), r["loss"])
    sd = eng.state_dict()
    for k in ("blocks.0.attn.qkv.weight", "blocks.11.mlp.fc2.weight", "head.weight", "pos_embed", "blocks.3.norm1.weight"):
        d = float((sd[k].cpu() - orc.p[k]).norm() / orc.p[k].norm())
        assert d < 1e-4, (k, d)
