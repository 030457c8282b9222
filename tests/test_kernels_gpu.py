"""Per-kernel parity: every C-ABI entry point against a plain PyTorch fp32 reference of
the same op on the same (bf16-rounded) inputs.  Tolerances are stated per test; integer
data is used where the result must be bit exact (GEMM operand layouts)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
bf16, f32 = torch.bfloat16, torch.float32


def ops():
    from gipvit import ops as o
    return o


def L():
    from gipvit import _lib
    return _lib


def close(got, ref, rtol, atol, what=""):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.4g} (ref max {float(ref.abs().max()):.4g})"


def ints(shape, dev, lo=-2, hi=3, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).to(dev).to(bf16)


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 384), (197 * 2, 1152, 384), (100, 72, 192), (37 * 8, 1536, 384),
                                   (130, 8, 128)])
def test_linear_nt_exact(dev, M, N, K):
    A, B = ints((M, K), dev, seed=1), ints((N, K), dev, seed=2)
    C = torch.full((M, N), 7.0, dtype=f32, device=dev)
    ops().linear(A, B, C, M, N, K)
    assert torch.equal(C, A.float() @ B.float().t())
    Cb = torch.empty(M, N, dtype=bf16, device=dev)
    ops().linear(A, B, Cb, M, N, K)
    assert torch.equal(Cb.float(), (A.float() @ B.float().t()).to(bf16).float())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (394, 384, 1152), (300, 192, 576 + 64), (256, 768, 128)])
def test_linear_nn_exact(dev, M, N, K):
    """dX = dY W : A [M,K] k-contiguous, B stored [K,N]."""
    A, B = ints((M, K), dev, seed=3), ints((K, N), dev, seed=4)
    C = torch.empty(M, N, dtype=f32, device=dev)
    ops().linear(A, B, C, M, N, K, trans_b=True)
    assert torch.equal(C, A.float() @ B.float())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (384, 1152, 394), (1536, 384, 1000), (192, 576, 37 * 8 + 5), (256, 256, 8192),
                                   (1536, 384, 6000), (1152, 384, 4100), (384, 1536, 5008), (2048, 384, 4096), (384, 1664, 4160), (768, 3072, 2560)])
def test_linear_tn_exact(dev, M, N, K):
    """dW = dY^T X : A stored [K,M], B stored [K,N]; K = tokens (any length)."""
    A, B = ints((K, M), dev, -1, 2, seed=5), ints((K, N), dev, -1, 2, seed=6)
    ref = A.float().t() @ B.float()
    C = torch.empty(M, N, dtype=f32, device=dev)
    ops().linear(A, B, C, M, N, K, trans_a=True, trans_b=True)
    assert torch.equal(C, ref)
    # ACCUM (+ split-K with f32 atomics when the grid is small): integers stay exact
    C2 = torch.ones(M, N, dtype=f32, device=dev)
    cs = torch.full((M,), 2.0, device=dev)          # fused bias gradient: column sums of A
    ops().linear(A, B, C2, M, N, K, trans_a=True, trans_b=True, epilogue=L().EPI_ACCUM, colsum_a=cs)
    assert torch.equal(C2, ref + 1.0)
    assert torch.equal(cs, 2.0 + A.float().sum(0))
    # same with slab-reduced split-K (caller workspace) instead of atomics
    ws = torch.empty(L().lib.gv_linear_workspace_bytes() // 4, dtype=f32, device=dev)
    C3 = torch.ones(M, N, dtype=f32, device=dev)
    cs3 = torch.full((M,), 2.0, device=dev)
    ops().linear(A, B, C3, M, N, K, trans_a=True, trans_b=True, epilogue=L().EPI_ACCUM, workspace=ws, colsum_a=cs3)
    # K >= 2048 with M % 128 == 0, N % 384 == 0 runs the ping-pong kernel; (384, 1664): its swapped orientation (C = [Q x P])
    assert torch.equal(C3, ref + 1.0)
    assert torch.equal(cs3, 2.0 + A.float().sum(0))


def test_linear_epilogues(dev):
    o, l = ops(), L()
    M, N, K = 394, 1536, 384
    g = torch.Generator().manual_seed(7)
    A = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(bf16)
    B = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(bf16)
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev)
    ref = A.float() @ B.float().t() + bias
    # bias + save_pre + gelu
    C = torch.empty(M, N, dtype=bf16, device=dev); pre = torch.empty(M, N, dtype=bf16, device=dev)
    o.linear(A, B, C, M, N, K, epilogue=l.EPI_BIAS | l.EPI_GELU | l.EPI_SAVE_PRE, bias=bias, aux_out=pre)
    close(pre, ref, 1e-2, 1e-2, "pre")
    close(C, torch.nn.functional.gelu(ref), 1e-2, 1e-2, "gelu")
    # bias + resid, f32 out
    Cf = torch.empty(M, N, dtype=f32, device=dev)
    o.linear(A, B, Cf, M, N, K, epilogue=l.EPI_BIAS | l.EPI_RESID, bias=bias, resid=resid)
    close(Cf, ref + resid, 1e-4, 1e-4, "resid")
    # dgelu: v * gelu'(aux)
    aux = torch.randn(M, N, generator=g).to(dev).to(bf16)
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    o.linear(A, B, C, M, N, K, epilogue=l.EPI_DGELU, aux_in=aux)
    close(C, (A.float() @ B.float().t()) * x.grad, 1e-2, 1e-2, "dgelu")
    # alpha
    o.linear(A, B, Cf, M, N, K, alpha=0.25)
    close(Cf, 0.25 * (A.float() @ B.float().t()), 1e-4, 1e-4, "alpha")


@pytest.mark.parametrize("M,N,K", [(640, 2048, 2048), (128, 256, 2048), (640, 384, 1536), (100, 384, 1024), (640, 1536, 384)])
def test_linear_few_rows_split_k_epilogue(dev, M, N, K):
    """Few tiles and a long reduction: handed a scratch buffer, gv_linear splits K over the idle CUs and one pass sums the partial
    tiles and applies the epilogue (the DINO head's 2048-wide layers, the CLS-only tail).  Exact on integer data against the
    unsplit call (no scratch), every epilogue of the path against fp32 torch, NN and NT forms, one guard row behind the outputs.
    (640 x 1536 x 384: too short a reduction to split -- the scratch must change nothing.)"""
    o, l = ops(), L()
    ws = torch.empty(32 << 20, dtype=f32, device=dev)
    A, B = ints((M, K), dev, seed=41), ints((N, K), dev, seed=42)
    bias_i = torch.arange(N, dtype=f32, device=dev) % 5 - 2
    ref = A.float() @ B.float().t()
    for tb, Bop in ((False, B), (True, B.t().contiguous())):
        C0 = torch.full((M + 1, N), 9.0, dtype=f32, device=dev); C1 = C0.clone()
        o.linear(A, Bop, C0, M, N, K, trans_b=tb, epilogue=l.EPI_BIAS, bias=bias_i)
        o.linear(A, Bop, C1, M, N, K, trans_b=tb, epilogue=l.EPI_BIAS, bias=bias_i, workspace=ws)
        assert torch.equal(C0, C1) and torch.equal(C1[:M], ref + bias_i) and float(C1[M].min()) == 9.0, f"integer data, trans_b={tb}"
    g = torch.Generator().manual_seed(43)
    Af = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(bf16)
    Bf = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(bf16)
    bias = torch.randn(N, generator=g).to(dev); resid = torch.randn(M, N, generator=g).to(dev)
    rs = (torch.rand(M, generator=g) + 0.5).to(dev)
    reff = Af.float() @ Bf.float().t()
    C = torch.full((M + 1, N), 9.0, dtype=bf16, device=dev); pre = torch.full((M + 1, N), 9.0, dtype=bf16, device=dev)
    o.linear(Af, Bf, C, M, N, K, epilogue=l.EPI_BIAS | l.EPI_GELU | l.EPI_SAVE_PRE, bias=bias, aux_out=pre, workspace=ws)
    close(pre[:M], reff + bias, 1e-2, 1e-2, "pre")
    close(C[:M], torch.nn.functional.gelu(reff + bias), 1e-2, 1e-2, "gelu")
    assert float(C[M].float().min()) == 9.0 and float(pre[M].float().min()) == 9.0
    Cf = torch.full((M + 1, N), 9.0, dtype=f32, device=dev)
    o.linear(Af, Bf, Cf, M, N, K, epilogue=l.EPI_BIAS | l.EPI_RESID, bias=bias, resid=resid, row_scale=rs, workspace=ws)
    close(Cf[:M], (reff + bias) * rs[:, None] + resid, 2e-4, 2e-4, "bias + row factor + residual")
    assert float(Cf[M].min()) == 9.0
    aux = torch.randn(M, N, generator=g).to(dev).to(bf16)
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    o.linear(Af, Bf.t().contiguous(), C, M, N, K, trans_b=True, epilogue=l.EPI_DGELU, aux_in=aux, workspace=ws)
    close(C[:M], reff * x.grad, 1e-2, 1e-2, "dgelu")
    o.linear(Af, Bf, Cf, M, N, K, alpha=0.25, workspace=ws)
    close(Cf[:M], 0.25 * reff, 1e-4, 1e-4, "alpha")


@pytest.mark.parametrize("M,N,K", [(2048, 384, 128), (2500, 1152, 384), (4099, 1536, 384), (3000, 384, 1536), (47000, 768, 128), (33000, 384, 256),
                                   (44160, 1536, 384), (44160, 1152, 384), (25216, 1536, 384), (9999, 3072, 768), (100000, 1152, 128)])
def test_linear_wide_panel(dev, M, N, K):
    """gv_linear's wide bf16 products (M >= 2048, N % 384 == 0, K % 128 == 0, hot-path epilogues) run on the full-row kernel:
    exact on integer data (NT and NN forms, ragged last row panel), epilogues against fp32 torch, and identical to the
    128x128-tile kernel's result on the same operands (a sub-range of rows below the M threshold)."""
    o, l = ops(), L()
    A, B = ints((M, K), dev, seed=31), ints((N, K), dev, seed=32)
    bias_i = torch.arange(N, dtype=f32, device=dev) % 5 - 2
    ref = A.float() @ B.float().t()
    Cb = torch.full((M + 1, N), 9.0, dtype=bf16, device=dev)                  # one guard row behind the output
    o.linear(A, B, Cb, M, N, K, epilogue=l.EPI_BIAS, bias=bias_i)
    assert torch.equal(Cb[:M].float(), (ref + bias_i).to(bf16).float()) and float(Cb[M].float().min()) == 9.0
    Bt = B.t().contiguous()                                                  # dX form: B stored [K, N]
    o.linear(A, Bt, Cb, M, N, K, trans_b=True)
    assert torch.equal(Cb[:M].float(), ref.to(bf16).float()) and float(Cb[M].float().min()) == 9.0
    # float data: GELU / saved pre-activation / GELU' epilogues
    g = torch.Generator().manual_seed(33)
    Af = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(bf16)
    Bf = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(bf16)
    bias = torch.randn(N, generator=g).to(dev)
    reff = Af.float() @ Bf.float().t() + bias
    C = torch.empty(M, N, dtype=bf16, device=dev); pre = torch.empty(M, N, dtype=bf16, device=dev)
    o.linear(Af, Bf, C, M, N, K, epilogue=l.EPI_BIAS | l.EPI_GELU | l.EPI_SAVE_PRE, bias=bias, aux_out=pre)
    close(pre, reff, 1e-2, 1e-2, "pre")
    close(C, torch.nn.functional.gelu(reff), 1e-2, 1e-2, "gelu")
    C2 = torch.empty(M, N, dtype=bf16, device=dev)
    o.linear(Af, Bf, C2, M, N, K, epilogue=l.EPI_BIAS | l.EPI_GELU, bias=bias)
    assert torch.equal(C2, C)
    aux = torch.randn(M, N, generator=g).to(dev).to(bf16)
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    Bft = Bf.t().contiguous()
    o.linear(Af, Bft, C, M, N, K, trans_b=True, epilogue=l.EPI_DGELU, aux_in=aux)
    close(C, (reff - bias) * x.grad, 1e-2, 1e-2, "dgelu")
    # the 128x128-tile kernel on the first 1000 rows (below the threshold) gives the same bits
    Cs = torch.empty(1000, N, dtype=bf16, device=dev)
    o.linear(Af, Bft, Cs, 1000, N, K, trans_b=True, epilogue=l.EPI_DGELU, aux_in=aux)
    assert torch.equal(Cs, C[:1000])
    # f32 output with bias + residual (x + proj(a) / x + fc2(h) of the widths without a fused LayerNorm kernel): exact on
    # integers, one guard row, the same bits as the tile kernel; with stochastic-depth row factors against fp32 torch
    resid_i = (torch.arange(M * N, device=dev) % 7 - 3).to(f32).view(M, N)
    Cf = torch.full((M + 1, N), 9.0, dtype=f32, device=dev)
    o.linear(A, B, Cf, M, N, K, epilogue=l.EPI_BIAS | l.EPI_RESID, bias=bias_i, resid=resid_i)
    assert torch.equal(Cf[:M], ref + bias_i + resid_i) and float(Cf[M].min()) == 9.0
    resid = torch.randn(M, N, generator=g).to(dev)
    rs = (torch.rand(M, generator=g) * 2).to(dev)
    o.linear(Af, Bf, Cf, M, N, K, epilogue=l.EPI_BIAS | l.EPI_RESID, bias=bias, resid=resid)
    close(Cf[:M], reff + resid, 1e-4, 1e-4, "bias + resid")
    Cfs = torch.empty(1000, N, dtype=f32, device=dev)
    o.linear(Af, Bf, Cfs, 1000, N, K, epilogue=l.EPI_BIAS | l.EPI_RESID, bias=bias, resid=resid)
    assert torch.equal(Cfs, Cf[:1000])
    o.linear(Af, Bf, Cf, M, N, K, epilogue=l.EPI_BIAS | l.EPI_RESID, bias=bias, resid=resid, row_scale=rs)
    close(Cf[:M], reff * rs[:, None] + resid, 1e-4, 1e-4, "row_scale")


@pytest.mark.parametrize("M", [25216, 44160, 640, 300, 113])
def test_mlp_ln_fwd(dev, M):
    """gv_mlp_ln_fwd: fc1 -> GELU -> fc2 -> + residual -> LayerNorm in one launch (the hidden activation stays in LDS).  Bit-identical to
    gv_linear (BIAS | GELU) followed by gv_linear_ln_fwd (same accumulation order, same GELU polynomial, same bf16 rounding of the
    hidden activation), and within bf16 tolerances of the fp32 torch expression; one guard row behind every output."""
    o, l = ops(), L()
    D, Hd = 384, 1536
    g = torch.Generator().manual_seed(41 + M)
    A = (torch.randn(M, D, generator=g) * 0.7).to(dev).to(bf16)
    W1 = (torch.randn(Hd, D, generator=g) * 0.05).to(dev).to(bf16)
    W2 = (torch.randn(D, Hd, generator=g) * 0.03).to(dev).to(bf16)
    b1, b2 = torch.randn(Hd, generator=g).to(dev) * 0.3, torch.randn(D, generator=g).to(dev) * 0.3
    resid = torch.randn(M, D, generator=g).to(dev)
    gam, bet = (1 + 0.2 * torch.randn(D, generator=g)).to(dev), (0.1 * torch.randn(D, generator=g)).to(dev)
    rs = (torch.rand(M, generator=g) * 2).to(dev)
    for row_scale, with_ln in ((None, True), (rs, True), (None, False)):
        out = torch.full((M + 1, D), 9.0, dtype=f32, device=dev); y = torch.full((M + 1, D), 9.0, dtype=bf16, device=dev)
        mean, rstd = torch.full((M + 1,), 9.0, device=dev), torch.full((M + 1,), 9.0, device=dev)
        kw = dict(gamma=gam, beta=bet, y=y, mean=mean, rstd=rstd) if with_ln else {}
        o.mlp_ln_fwd(A, W1, b1, W2, out, M, D, Hd, bias2=b2, resid=resid, row_scale=row_scale, **kw)
        # the unfused pair
        h = torch.empty(M, Hd, dtype=bf16, device=dev)
        o.linear(A, W1, h, M, Hd, D, epilogue=l.EPI_BIAS | l.EPI_GELU, bias=b1)
        out2 = torch.empty(M, D, dtype=f32, device=dev); y2 = torch.empty(M, D, dtype=bf16, device=dev)
        mean2, rstd2 = torch.empty(M, device=dev), torch.empty(M, device=dev)
        kw2 = dict(gamma=gam, beta=bet, y=y2, mean=mean2, rstd=rstd2) if with_ln else {}
        o.linear_ln_fwd(h, W2, out2, M, Hd, bias=b2, resid=resid, row_scale=row_scale, **kw2)
        assert torch.equal(out[:M], out2) and float(out[M].min()) == 9.0
        if with_ln:
            assert torch.equal(y[:M], y2) and torch.equal(mean[:M], mean2) and torch.equal(rstd[:M], rstd2)
            assert float(y[M].float().min()) == 9.0 and float(mean[M]) == 9.0
    # fp32 torch (bf16 hidden activation and bf16 operands: the tolerances of test_linear_epilogues)
    hr = torch.nn.functional.gelu(A.float() @ W1.float().t() + b1)
    ref = resid + hr @ W2.float().t() + b2
    out = torch.empty(M, D, dtype=f32, device=dev); y = torch.empty(M, D, dtype=bf16, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    o.mlp_ln_fwd(A, W1, b1, W2, out, M, D, Hd, bias2=b2, resid=resid, gamma=gam, beta=bet, y=y, mean=mean, rstd=rstd)
    close(out, ref, 1e-2, 2e-2, "fused MLP vs fp32 torch")
    close(y, torch.nn.functional.layer_norm(ref, (D,), gam, bet, 1e-6), 2e-2, 3e-2, "fused MLP: LayerNorm row")


def test_linear_pos_epilogue(dev):
    """patch-embed epilogue: rows remapped past the CLS slot, pos-embed added."""
    o, l = ops(), L()
    n_img, P, D, K = 3, 36, 384, 768
    M = n_img * P
    g = torch.Generator().manual_seed(8)
    A = torch.randn(M, K, generator=g).to(dev).to(bf16)
    W = (torch.randn(D, K, generator=g) * 0.03).to(dev).to(bf16)
    bias = torch.randn(D, generator=g).to(dev); pos = torch.randn(P + 1, D, generator=g).to(dev)
    x = torch.zeros(n_img * (P + 1), D, dtype=f32, device=dev)
    o.linear(A, W, x, M, D, K, epilogue=l.EPI_BIAS | l.EPI_POS, bias=bias, pos=pos, P=P)
    ref = (A.float() @ W.float().t() + bias).view(n_img, P, D) + pos[1:]
    xv = x.view(n_img, P + 1, D)
    close(xv[:, 1:], ref, 1e-4, 1e-3, "pos rows")
    assert float(xv[:, 0].abs().max()) == 0.0


def test_linear_rejects_bad_shapes(dev):
    A = torch.zeros(64, 100, dtype=bf16, device=dev); B = torch.zeros(64, 100, dtype=bf16, device=dev)
    C = torch.zeros(64, 64, dtype=f32, device=dev)
    with pytest.raises(L().GipvitError, match="multiple of 32"):
        ops().linear(A, B, C, 64, 64, 100, lda=104, ldb=104)


def test_linear_timing_rows(dev):
    """gv_linear_timing: HIP-event timing of the GEMM launches, one row per kernel instantiation."""
    o = ops()
    A, B = ints((256, 384), dev, seed=1), ints((128, 384), dev, seed=2)
    C = torch.empty(256, 128, dtype=bf16, device=dev)
    Ct, Bt = torch.zeros(384, 128, dtype=f32, device=dev), ints((256, 128), dev, seed=3)
    o.linear(A, B, C, 256, 128, 384)                                   # outside the window: not recorded
    o.linear_timing(True)
    for _ in range(3):
        o.linear(A, B, C, 256, 128, 384)                               # NT, bf16 out, epilogue 0
    o.linear(A, Bt, Ct, 384, 128, 256, trans_a=True, trans_b=True)     # TN, f32 out, epilogue 0
    o.linear_timing(False)
    o.linear(A, B, C, 256, 128, 384)
    rows = {r["kernel"]: r for r in o.linear_timing_read()}
    nt = rows["gemm_kernel<false, false, bf16, false, 0>"]
    tn = rows["gemm_kernel<true, true, float, false, 0>"]
    assert nt["bytes"] == 3 * 2.0 * (256 * 384 + 128 * 384 + 256 * 128)      # algorithmic bytes: operands and output once
    assert tn["bytes"] == 2.0 * 256 * (384 + 128) + 4.0 * 384 * 128
    assert len(rows) == 2 and nt["launches"] == 3 and tn["launches"] == 1
    assert nt["flops"] == 3 * 2.0 * 256 * 128 * 384 and tn["flops"] == 2.0 * 384 * 128 * 256
    assert 0.0 < nt["seconds"] < 1.0 and 0.0 < tn["seconds"] < 1.0
    o.linear_timing(True); o.linear_timing(False)                      # a new window starts empty
    assert o.linear_timing_read() == []


# ------------------------------------------------------- full-row Linear + LayerNorm (panel kernel)
@pytest.mark.parametrize("M,K", [(64, 64), (1380, 384), (44160, 1536), (25216, 384), (2000, 1152), (33000, 128), (1380, 192), (47000, 384)])
def test_linear_ln_fwd(dev, M, K):
    """gv_linear_ln_fwd: out = A W^T + bias + resid exactly (integer operands), and LayerNorm of the new row against
    F.layer_norm in fp32 (bf16 output tolerance); every supported rows-per-workgroup geometry (M picks FM = 4 .. 12)."""
    o = ops()
    N = 384
    g = torch.Generator().manual_seed(M + K)
    A, W = ints((M, K), dev, seed=5), ints((N, K), dev, seed=6)
    bias = torch.randint(-3, 4, (N,), generator=g).float().to(dev)
    resid = torch.randint(-8, 9, (M, N), generator=g).float().to(dev)
    gamma, beta = (1.0 + 0.1 * torch.randn(N, generator=g)).to(dev), (0.1 * torch.randn(N, generator=g)).to(dev)
    out = torch.full((M, N), 7.0, dtype=f32, device=dev)
    y = torch.empty(M, N, dtype=bf16, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    o.linear_ln_fwd(A, W, out, M, K, bias=bias, resid=resid, gamma=gamma, beta=beta, y=y, mean=mean, rstd=rstd)
    ref = A.float() @ W.float().t() + bias + resid
    assert torch.equal(out, ref)                                    # integers: exact in f32 whatever the summation order
    close(mean, ref.mean(1), 1e-5, 1e-5, "mean")
    close(rstd, 1.0 / torch.sqrt(ref.var(1, unbiased=False) + 1e-6), 1e-4, 1e-6, "rstd")
    close(y, torch.nn.functional.layer_norm(ref, (N,), gamma, beta, 1e-6), 8e-3, 8e-3, "ln out")
    # without LayerNorm outputs / bias / residual
    out2 = torch.empty(M, N, dtype=f32, device=dev)
    o.linear_ln_fwd(A, W, out2, M, K)
    assert torch.equal(out2, A.float() @ W.float().t())
    # real-valued operands: same numbers as the 128x128 kernel's f32 accumulation up to summation order
    Ar, Wr = torch.randn(M, K, generator=g).to(dev).to(bf16), (0.05 * torch.randn(N, K, generator=g)).to(dev).to(bf16)
    rr = torch.randn(M, N, generator=g).to(dev)
    o.linear_ln_fwd(Ar, Wr, out, M, K, bias=bias, resid=rr, gamma=gamma, beta=beta, y=y, mean=mean, rstd=rstd)
    ref = Ar.float() @ Wr.float().t() + bias + rr
    close(out, ref, 1e-4, 1e-3 * math.sqrt(K / 384), "out real")
    close(y, torch.nn.functional.layer_norm(ref, (N,), gamma, beta, 1e-6), 8e-3, 8e-3, "ln out real")


@pytest.mark.parametrize("M,K,g_init", [(64, 64, False), (1380, 1536, False), (44160, 1152, False), (25216, 1536, True), (3000, 384, False), (1380, 192, False), (47000, 384, False), (33000, 1152, False)])
def test_linear_ln_bwd(dev, M, K, g_init):
    """gv_linear_ln_bwd: dXn = dY W (W stored [K, N]) + the LayerNorm backward it feeds, against autograd in fp32."""
    o = ops()
    N = 384
    g = torch.Generator().manual_seed(M * 3 + K)
    dY = (0.5 * torch.randn(M, K, generator=g)).to(dev).to(bf16)
    W = (0.05 * torch.randn(K, N, generator=g)).to(dev).to(bf16)
    x = (torch.randn(M, N, generator=g) * 1.5 + 0.3).to(dev)
    gamma, beta = (1.0 + 0.1 * torch.randn(N, generator=g)).to(dev), torch.zeros(N, device=dev)
    g0 = torch.randn(M, N, generator=g).to(dev)
    _, mean, rstd = o.layernorm_fwd(x, gamma, beta, M, N)
    gbuf = g0.clone(); gb = torch.empty(M, N, dtype=bf16, device=dev)
    partials = torch.full((L().LN_PARTIAL_BLOCKS, 3, N), float("nan"), dtype=f32, device=dev)
    nb = o.linear_ln_bwd(dY, W, x, mean, rstd, gamma, gbuf, gb, partials, M, K, g_init=g_init)
    assert nb == L().lib.gv_linear_ln_blocks(M) and 0 < nb <= 256 * ((M + 16 * 12 * 256 - 1) // (16 * 12 * 256))
    dxn = dY.float() @ W.float()
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (N,), gr, br, 1e-6).backward(dxn)
    want = xr.grad + (0 if g_init else g0)
    scale = float(want.abs().max())
    close(gbuf, want, 1e-3, 2e-4 * scale, "g")
    close(gb, want, 8e-3, 8e-3 * scale, "gb")
    outs = [torch.ones(N, device=dev) for _ in range(3)]
    o.ln_finalize(partials, nb, N, outs[0], outs[1], outs[2])
    close(outs[0], 1 + gr.grad, 2e-3, 2e-3 * float(gr.grad.abs().max()), "dgamma")
    close(outs[1], 1 + br.grad, 2e-3, 2e-3 * float(br.grad.abs().max()), "dbeta")
    close(outs[2], 1 + want.sum(0), 2e-3, 2e-3 * float(want.sum(0).abs().max()), "colsum g")
    assert bool(torch.isnan(partials[nb:]).all())                   # nothing written beyond the reported block count


@pytest.mark.parametrize("T", [1500, 3000, 8192])
def test_linear_dw_group_exact(dev, T):
    """gv_linear_dw_group: the four weight gradients of a block in one split-K launch == four dY^T X products (+ column sums),
    bit exact on integer data; accumulates into dW.  T < 2048 runs the 128x128-tile kernel, T >= 2048 the ping-pong kernel
    (ragged last K-tile at 3000)."""
    o = ops()
    shapes = [(384, 1536), (1536, 384), (384, 384), (1152, 384)]
    probs, refs = [], []
    for q, (Mq, Nq) in enumerate(shapes):
        dY, X = ints((T, Mq), dev, seed=10 + q), ints((T, Nq), dev, seed=20 + q)
        dW = torch.full((Mq, Nq), float(q), dtype=f32, device=dev)
        cs = torch.full((Mq,), 2.0, dtype=f32, device=dev) if q % 2 else None
        probs.append((dY, X, dW, cs))
        refs.append((q + dY.float().t() @ X.float(), None if cs is None else 2.0 + dY.float().sum(0)))
    ws = torch.empty(L().lib.gv_linear_workspace_bytes() // 4, dtype=f32, device=dev)
    o.linear_dw_group(probs, T, ws)
    for (dY, X, dW, cs), (rw, rc) in zip(probs, refs):
        assert torch.equal(dW, rw)
        if cs is not None:
            assert torch.equal(cs, rc)


def test_linear_ln_rejects_other_widths(dev):
    A = torch.zeros(64, 64, dtype=bf16, device=dev); W = torch.zeros(192, 64, dtype=bf16, device=dev); out = torch.zeros(64, 192, device=dev)
    with pytest.raises(L().GipvitError, match="N = 384"):
        ops().linear_ln_fwd(A, W, out, 64, 64, N=192)


# ------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("D", [192, 384, 768])
def test_layernorm_fwd_bwd(dev, D):
    o = ops()
    rows = 1000 + 3
    g = torch.Generator().manual_seed(D)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.5).to(dev)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev); beta = (0.1 * torch.randn(D, generator=g)).to(dev)
    y, mean, rstd = o.layernorm_fwd(x, gamma, beta, rows, D)
    ref = torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-6)
    close(y, ref, 8e-3, 8e-3, "ln fwd")           # bf16 output rounding
    close(mean, x.mean(-1), 1e-5, 1e-5, "mean")
    # backward: g += dLN/dx, gb = bf16(g), partial column sums
    dy = torch.randn(rows, D, generator=g).to(dev).to(bf16)
    g0 = torch.randn(rows, D, generator=g).to(dev)
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6).backward(dy.float())
    gbuf = g0.clone(); gb = torch.empty(rows, D, dtype=bf16, device=dev)
    partials = torch.empty(L().LN_PARTIAL_BLOCKS, 3, D, dtype=f32, device=dev)
    o.layernorm_bwd(dy, x, mean, rstd, gamma, gbuf, gb, partials, rows, D)
    close(gbuf, g0 + xr.grad, 1e-4, 1e-4, "ln dx")
    close(gb, g0 + xr.grad, 8e-3, 8e-3, "gb")
    outs = [torch.zeros(D, device=dev) for _ in range(3)]
    for w in range(3):
        o.colsum_finalize(partials, L().LN_PARTIAL_BLOCKS, 3, w, D, outs[w], False)
    close(outs[0], gr.grad, 1e-3, 1e-2, "dgamma")
    close(outs[1], br.grad, 1e-3, 1e-2, "dbeta")
    close(outs[2], (g0 + xr.grad).sum(0), 1e-3, 1e-2, "colsum g")
    o3 = [torch.ones(D, device=dev) for _ in range(3)]
    o.ln_finalize(partials, L().LN_PARTIAL_BLOCKS, D, o3[0], None, o3[2])
    close(o3[0], 1 + gr.grad, 1e-3, 1e-2, "ln_finalize dgamma")
    close(o3[2], 1 + (g0 + xr.grad).sum(0), 1e-3, 1e-2, "ln_finalize colsum")
    assert float((o3[1] - 1).abs().max()) == 0.0
    # strided rows + g_init (final norm on CLS rows)
    N = 5; n_img = 40
    xs = torch.randn(n_img * N, D, generator=g).to(dev)
    y2, m2, r2 = o.layernorm_fwd(xs, gamma, beta, n_img, D, x_stride=N * D)
    close(y2, torch.nn.functional.layer_norm(xs.view(n_img, N, D)[:, 0], (D,), gamma, beta, 1e-6), 8e-3, 8e-3, "ln strided")
    gfull = torch.full((n_img * N, D), 3.0, device=dev)
    dy2 = torch.randn(n_img, D, generator=g).to(dev).to(bf16)
    o.layernorm_bwd(dy2, xs, m2, r2, gamma, gfull, None, partials, n_img, D, x_stride=N * D, g_stride=N * D, g_init=True)
    xr2 = xs.view(n_img, N, D)[:, 0].clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr2, (D,), gamma, beta, 1e-6).backward(dy2.float())
    gv = gfull.view(n_img, N, D)
    close(gv[:, 0], xr2.grad, 1e-4, 1e-4, "ln dx strided")
    assert float((gv[:, 1:] - 3.0).abs().max()) == 0.0


def test_colsum(dev):
    o = ops()
    rows, C = 2051, 1152
    x = torch.randn(rows, C, generator=torch.Generator().manual_seed(1)).to(dev)
    ws = torch.empty(64 * C, device=dev); out = torch.ones(C, device=dev)
    o.colsum(x.to(bf16), rows, C, ws, out, accumulate=True)
    close(out, 1 + x.to(bf16).float().sum(0), 1e-4, 1e-2, "colsum bf16")
    o.colsum(x, rows, C, ws, out, accumulate=False)
    close(out, x.sum(0), 1e-4, 1e-3, "colsum f32")


# ------------------------------------------------------------------------ attention
def _attn_ref(qkv, n_img, N, H, scale):
    q, k, v = qkv.float().view(n_img, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * scale
    p = s.softmax(-1)
    o = (p @ v).transpose(1, 2).reshape(n_img * N, H * 64)
    return o, torch.logsumexp(s, -1)


@pytest.mark.parametrize("N,H,n_img", [(197, 6, 5), (37, 6, 7), (17, 3, 9), (257, 6, 2), (65, 12, 3), (1, 3, 2), (224, 2, 1)])
def test_attention_fwd_bwd(dev, N, H, n_img):
    o = ops()
    g = torch.Generator().manual_seed(N * 7 + H)
    qkv = torch.randn(n_img * N, 3 * H * 64, generator=g).to(dev).to(bf16)
    scale = 64 ** -0.5
    out, lse = o.attention_fwd(qkv, n_img, N, H, scale)
    x = qkv.float().requires_grad_(True)
    ref, ref_lse = _attn_ref(x, n_img, N, H, scale)
    close(out, ref, 2e-2, 2e-2, "attn o")        # bf16 P and bf16 output
    close(lse, ref_lse, 1e-3, 1e-3, "lse")
    d_o = torch.randn(n_img * N, H * 64, generator=g).to(dev).to(bf16)
    ref.backward(d_o.float())
    dqkv = o.attention_bwd(qkv, out, d_o, lse, n_img, N, H, scale)
    scale_ref = float(x.grad.abs().max())
    close(dqkv, x.grad, 3e-2, 2e-2 * max(scale_ref, 1.0), "dqkv")


def test_attention_softmax_spike(dev):
    """one key dominating one query row: exercises the max-subtraction path (rule 26)."""
    o = ops()
    N, H, n_img = 197, 6, 2
    qkv = (torch.randn(n_img * N, 3 * H * 64, generator=torch.Generator().manual_seed(3)) * 0.1).to(dev).to(bf16)
    v = qkv.view(n_img, N, 3, H, 64)
    v[0, 5, 0, 2] = 8.0; v[0, 100, 1, 2] = 8.0       # q row 5 . k row 100 = 64*64*scale = 512
    out, lse = o.attention_fwd(qkv, n_img, N, H, 0.125)
    ref, ref_lse = _attn_ref(qkv, n_img, N, H, 0.125)
    assert bool(torch.isfinite(out.float()).all())
    close(out, ref, 2e-2, 2e-2, "spike o")
    close(lse, ref_lse, 1e-3, 1e-2, "spike lse")


# ------------------------------------------------------------------------- patchify
def test_patchify(dev):
    o = ops()
    g = torch.Generator().manual_seed(11)
    tiles = torch.randint(0, 256, (3, 256, 256, 3), generator=g, dtype=torch.uint8).to(dev)
    mean, std = (0.8998, 0.8253, 0.9357), (0.1125, 0.1751, 0.0787)
    for crop, wins in ((224, [(0, 0), (16, 16)]), (96, [(20 * l, 160 - 20 * l) for l in range(8)]), (96, [(1, 3), (7, 157)])):
        p = o.patchify(tiles, wins, crop, mean, std)
        side = crop // 16
        refs = []
        for (y0, x0) in wins:
            x = tiles[:, y0:y0 + crop, x0:x0 + crop, :].float() / 255.0
            x = (x - torch.tensor(mean, device=dev)) / torch.tensor(std, device=dev)
            x = x.permute(0, 3, 1, 2)                                   # [n,3,crop,crop]
            x = x.reshape(3, 3, side, 16, side, 16).permute(0, 2, 4, 1, 3, 5)  # n, prow, pcol, c, py, px
            refs.append(x.reshape(3 * side * side, 768))
        close(p, torch.cat(refs), 4e-3, 4e-3, f"patchify {crop}")     # bf16 rounding only


# -------------------------------------------------------------------- DINO head tail
def test_l2norm_weightnorm(dev):
    o = ops()
    g = torch.Generator().manual_seed(5)
    rows, C = 77, 256
    x = torch.randn(rows, C, generator=g).to(dev)
    y = torch.empty(rows, C, dtype=bf16, device=dev); inv = torch.empty(rows, device=dev)
    o.l2norm_fwd(x, y, inv, rows, C)
    close(y, torch.nn.functional.normalize(x, dim=-1), 8e-3, 1e-3, "l2norm")
    dy = torch.randn(rows, C, generator=g).to(dev)
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.normalize(xr, dim=-1).backward(dy)
    dx = torch.empty(rows, C, dtype=bf16, device=dev)
    o.l2norm_bwd(dy, y, inv, dx, rows, C)
    close(dx, xr.grad, 2e-2, 2e-3, "l2norm bwd")
    # C = 256 takes the two-rows-per-wave kernels (ragged last block at 1003 rows), any other width the generic ones
    for K, C in ((1000, 256), (1003, 256), (37, 128)):
        v = torch.randn(K, C, generator=g).to(dev); gg = (1 + 0.1 * torch.randn(K, generator=g)).to(dev)
        w = torch.empty(K, C, dtype=bf16, device=dev)
        o.weightnorm_fwd(v, gg, w, K, C)
        close(w, gg[:, None] * v / v.norm(dim=1, keepdim=True), 8e-3, 1e-3, f"weightnorm {K}x{C}")
        dw = torch.randn(K, C, generator=g).to(dev)
        vr = v.clone().requires_grad_(True); gr = gg.clone().requires_grad_(True)
        (gr[:, None] * vr / vr.norm(dim=1, keepdim=True)).backward(dw)
        dv = torch.ones(K, C, device=dev); dg = torch.zeros(K, device=dev)
        o.weightnorm_bwd(dw, v, gg, dv, dg, K, C, accumulate=True)
        close(dv, 1 + vr.grad, 1e-4, 1e-4, f"dv {K}x{C}")
        close(dg, gr.grad, 1e-4, 1e-4, f"dg {K}x{C}")
        dv2 = torch.full((K, C), 7.0, device=dev); dg2 = torch.full((K,), 7.0, device=dev)
        o.weightnorm_bwd(dw, v, gg, dv2, dg2, K, C, accumulate=False)
        close(dv2, vr.grad, 1e-4, 1e-4, f"dv (overwrite) {K}x{C}")
        close(dg2, gr.grad, 1e-4, 1e-4, f"dg (overwrite) {K}x{C}")


# ------------------------------------------------------------------------ DINO loss
@pytest.mark.parametrize("B,V,G,K", [(4, 10, 2, 4096), (3, 4, 2, 256), (2, 2, 2, 65536), (5, 6, 1, 1024)])
def test_dino_loss(dev, B, V, G, K):
    from oracle import vit_oracle as vo
    o = ops()
    g = torch.Generator().manual_seed(B * 100 + V)
    s = torch.randn(V * B, K, generator=g).to(dev); t = torch.randn(G * B, K, generator=g).to(dev)
    center = (0.1 * torch.randn(1, K, generator=g)).to(dev)
    sr = s.clone().requires_grad_(True)
    loss_ref, csum_ref = vo.dino_loss(sr, t, center, V, G, 0.1, 0.04)
    loss_ref.backward()
    ds = torch.empty(V * B, K, dtype=bf16, device=dev); loss = torch.empty(1, device=dev)
    csum = torch.empty(K, device=dev); ws = torch.empty(2 * (V + G) * B, device=dev)
    o.dino_loss(s, t, center, ds, loss, csum, ws, B, V, G, K, 0.1, 0.04)
    assert abs(float(loss) - float(loss_ref)) < 1e-4 * max(1.0, abs(float(loss_ref))), (float(loss), float(loss_ref))
    close(csum, csum_ref[0], 1e-4, 1e-4, "center sum")
    gmax = float(sr.grad.abs().max())
    close(ds, sr.grad, 1e-2, 1e-2 * gmax, "dstudent")
    c2 = center[0].clone()
    o.center_update(c2, csum, K, 0.9, 1.0 / (G * B))
    close(c2, vo.update_center(center, csum_ref, G * B, 0.9)[0], 1e-5, 1e-6, "center")


def test_softmax_lsce(dev):
    from oracle import vit_oracle as vo
    o = ops()
    for B, C in ((8, 2), (300, 5)):
        g = torch.Generator().manual_seed(B)
        z = torch.randn(B, C, generator=g).to(dev); tgt = torch.randint(0, C, (B, 1), generator=g).to(dev)
        zr = z.clone().requires_grad_(True)
        ref = vo.softmax_lsce(zr, tgt, 0.1); ref.backward()
        loss = torch.empty(1, device=dev); dz = torch.empty(B, C, device=dev); prob = torch.empty(B, C, device=dev)
        o.softmax_lsce(z, tgt.view(-1), loss, dz, prob, B, C, 0.1)
        assert abs(float(loss) - float(ref)) < 1e-5
        close(dz, zr.grad, 1e-4, 1e-6, "dlogits")
        close(prob, torch.softmax(z, 1), 1e-5, 1e-6, "prob")


# ---------------------------------------------------------------- tokens, misc, optim
def test_tokens_and_misc(dev):
    o = ops()
    g = torch.Generator().manual_seed(9)
    n_img, N, D = 6, 37, 384
    cls = torch.randn(D, generator=g).to(dev); pos = torch.randn(N, D, generator=g).to(dev)
    x = torch.zeros(n_img * N, D, device=dev)
    o.cls_rows(x, cls, pos, n_img, N, D)
    close(x.view(n_img, N, D)[:, 0], (cls + pos[0]).expand(n_img, D), 0, 1e-6, "cls rows")
    gg = torch.randn(n_img * N, D, generator=g).to(dev)
    gp = torch.empty(n_img * (N - 1), D, dtype=bf16, device=dev); dpos = torch.ones(N, D, device=dev); dcls = torch.ones(D, device=dev)
    o.tokens_bwd(gg, gp, dpos, dcls, n_img, N, D, accumulate=True)
    gv = gg.view(n_img, N, D)
    close(dpos, 1 + gv.sum(0), 1e-5, 1e-5, "dpos"); close(dcls, 1 + gv[:, 0].sum(0), 1e-5, 1e-5, "dcls")
    close(gp, gv[:, 1:].reshape(-1, D), 8e-3, 1e-3, "gpatch")
    y = torch.empty(n_img, D, dtype=bf16, device=dev)
    o.gather_cls(gg, y, n_img, N, D)
    close(y, gv[:, 0], 8e-3, 1e-3, "gather cls")
    A = torch.randn(36, 196, generator=g).to(dev); Bm = torch.randn(196, D, generator=g).to(dev)
    C = torch.empty(36, D, device=dev)
    o.small_matmul(A, Bm, C, 36, D, 196, sam=196, sak=1, sbk=D, sbn=1)
    close(C, A @ Bm, 1e-4, 1e-4, "small matmul")
    C2 = torch.ones(196, D, device=dev); Bt = torch.randn(36, D, generator=g).to(dev)
    o.small_matmul(A, Bt, C2, 196, D, 36, sam=1, sak=196, sbk=D, sbn=1, accumulate=True)
    close(C2, 1 + A.t() @ Bt, 1e-4, 1e-4, "small matmul T")
    n = 1_000_003
    src = torch.randn(n, generator=g).to(dev); dst = torch.empty(n, dtype=bf16, device=dev)
    o.cast_bf16(src, dst)
    assert torch.equal(dst, src.to(bf16))
    ws = torch.empty(1024, device=dev); out = torch.empty(1, device=dev)
    o.sumsq(src, ws, out)
    assert abs(float(out) - float((src.double() ** 2).sum())) < 1e-3 * n


def test_adamw_ema(dev):
    o = ops()
    n = 4096 * 33
    g = torch.Generator().manual_seed(2)
    p0 = torch.randn(n, generator=g).to(dev); grad = torch.randn(n, generator=g).to(dev)
    t0 = torch.randn(n, generator=g).to(dev)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.04)
    p = p0.clone(); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    pb = torch.empty(n, dtype=bf16, device=dev); t = t0.clone(); tb = torch.empty(n, dtype=bf16, device=dev)
    tref = t0.clone()
    for step in (1, 2, 3):
        pr.grad = grad.clone() * step
        opt.step()
        tref = 0.99 * tref + 0.01 * pr.detach()
        o.adamw_ema(p, grad * step, m, v, pb, t, tb, n, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.04,
                    step=step, teacher_momentum=0.99)
    close(p, pr.detach(), 1e-5, 1e-6, "adamw p")
    close(t, tref, 1e-5, 1e-6, "ema teacher")
    assert torch.equal(pb, p.to(bf16)) and torch.equal(tb, t.to(bf16))
    # clipping: scale = clip / (||g|| + 1e-6)
    p2 = p0.clone(); m.zero_(); v.zero_()
    gn = torch.tensor([float((grad.double() ** 2).sum())], device=dev)
    o.adamw_ema(p2, grad, m, v, None, None, None, n, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step=1,
                clip_norm=3.0, gnorm_sq=gn)
    c = 3.0 / (math.sqrt(float(gn)) + 1e-6)
    close(m, 0.1 * grad * c, 1e-4, 1e-7, "clipped m")
    # --clip-mode value (torch.nn.utils.clip_grad_value_): element-wise clamp of the scaled gradient
    m.zero_(); v.zero_(); p2 = p0.clone()
    cv = float(grad.abs().median())
    o.adamw_ema(p2, grad, m, v, None, None, None, n, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step=1, grad_scale=0.5, clip_value=cv)
    close(m, 0.1 * (0.5 * grad).clamp(-cv, cv), 1e-5, 1e-8, "value-clipped m")
    with pytest.raises(L().GipvitError, match="exclude"):
        o.adamw_ema(p2, grad, m, v, None, None, None, n, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step=1, clip_norm=1.0, gnorm_sq=gn, clip_value=1.0)
    # mode 1: Adam with L2 decay (timm --opt adam); mode 2: SGD Nesterov (timm --opt sgd)
    for mode, mk in ((1, lambda q: torch.optim.Adam([q], lr=1e-3, weight_decay=0.01)),
                     (2, lambda q: torch.optim.SGD([q], lr=1e-2, momentum=0.9, nesterov=True, weight_decay=0.01))):
        pr = p0.clone().requires_grad_(True); opt = mk(pr)
        p3 = p0.clone(); m.zero_(); v.zero_()
        for step in (1, 2, 3):
            pr.grad = grad.clone() * step; opt.step()
            o.adamw_ema(p3, grad * step, m, v, None, None, None, n, lr=1e-3 if mode == 1 else 1e-2, beta1=0.9, beta2=0.999, eps=1e-8,
                        weight_decay=0.01, step=step, mode=mode)
        close(p3, pr.detach(), 1e-5, 1e-6, f"optimizer mode {mode}")


@pytest.mark.parametrize("segs", [[(6, 197), (20, 37)], [(9, 37), (3, 197)], [(130, 197), (530, 37)], [(4, 100), (5, 37)], [(3, 197)], [(2, 224), (7, 64), (3, 17)]])
def test_attention_fwd_varlen(dev, segs):
    """gv_attention_fwd_varlen: the segments of a token-concatenated row space (multi-crop: 197- and 37-token crops) in one call.  A
    long + a short segment run as ONE launch (two short instances per workgroup fill the long launch's last round); every other
    mix runs per segment.  Either way the result equals the per-segment calls bit for bit, and rows outside are untouched."""
    o, H = ops(), 6
    g = torch.Generator().manual_seed(17)
    T = sum(n * N for n, N in segs)
    qkv = torch.randn(T + 3, 3 * H * 64, generator=g).to(dev).to(bf16)
    out = torch.full((T + 3, H * 64), 7.0, dtype=bf16, device=dev)
    lses = [torch.empty(n, H, N, dtype=f32, device=dev) for n, N in segs]
    o.attention_fwd_varlen(qkv, out, [(n, N, l) for (n, N), l in zip(segs, lses)], H, 0.125)
    row = 0
    for (n, N), l in zip(segs, lses):
        ref_o, ref_l = o.attention_fwd(qkv[row:row + n * N], n, N, H, 0.125)
        assert torch.equal(out[row:row + n * N], ref_o) and torch.equal(l, ref_l), (n, N)
        row += n * N
    assert float(out[T:].float().min()) == 7.0
    # ... and against fp32 torch on the first segment
    n, N = segs[0]
    q, k, v = (qkv[:n * N].float().view(n, N, 3, H, 64)[:, :, i].transpose(1, 2) for i in range(3))
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(n * N, H * 64)
    close(out[:n * N], ref, 2e-2, 2e-2, "varlen vs sdpa")


@pytest.mark.parametrize("segs", [[(6, 197), (20, 37)], [(9, 37), (3, 197)], [(2, 224), (7, 64), (3, 17)], [(3, 100)]])
def test_attention_bwd_varlen(dev, segs):
    """gv_attention_bwd_varlen: the backward of a token-concatenated multi-crop row space in one call (the forward's segment table).
    Checked against the fp32 torch expression (autograd through softmax(q k^T scale) v per segment) on the WHOLE mixed-length row
    space, against the per-segment calls bit for bit, and rows outside the segments stay untouched."""
    o, H, scale = ops(), 6, 0.125
    g = torch.Generator().manual_seed(23)
    T = sum(n * N for n, N in segs)
    qkv = torch.randn(T + 2, 3 * H * 64, generator=g).to(dev).to(bf16)
    d_o = torch.randn(T + 2, H * 64, generator=g).to(dev).to(bf16)
    out = torch.empty(T + 2, H * 64, dtype=bf16, device=dev)
    lses = [torch.empty(n, H, N, dtype=f32, device=dev) for n, N in segs]
    table = [(n, N, l) for (n, N), l in zip(segs, lses)]
    o.attention_fwd_varlen(qkv, out, table, H, scale)
    dqkv = torch.full((T + 2, 3 * H * 64), 5.0, dtype=bf16, device=dev)
    o.attention_bwd_varlen(qkv, out, d_o, dqkv, table, H, scale)
    assert float(dqkv[T:].float().min()) == 5.0 and float(dqkv[T:].float().max()) == 5.0
    row = 0
    for (n, N), l in zip(segs, lses):
        r = slice(row, row + n * N)
        ref_seg = o.attention_bwd(qkv[r], out[r], d_o[r], l, n, N, H, scale)
        assert torch.equal(dqkv[r], ref_seg), (n, N)
        x = qkv[r].float().view(n, N, 3, H, 64).requires_grad_(True)
        q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
        y = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(n * N, H * 64)
        y.backward(d_o[r].float())
        close(dqkv[r], x.grad.reshape(n * N, 3 * H * 64), 3e-2, 2e-2 * max(float(x.grad.abs().max()), 1.0), f"varlen backward vs fp32 torch, segment {(n, N)}")
        row += n * N


@pytest.mark.parametrize("N,H,n_img,ql", [(197, 6, 5, 1), (37, 6, 9, 1), (197, 3, 2, 40), (257, 6, 2, 1), (65, 12, 3, 33), (17, 3, 4, 1)])
def test_attention_q_limit(dev, N, H, n_img, ql):
    """q_limit (the last block of a ViT whose forward returns the CLS row): the forward computes only the first query rows (whole
    32-row groups), the backward skips every query behind them when dO is zero there.  Bit-equal to the full kernels on the rows that
    are computed (forward) and on every gradient (backward: dK, dV, the kept dQ rows; the skipped dQ rows are exactly zero)."""
    o, scale = ops(), 0.125
    g = torch.Generator().manual_seed(N + ql)
    qkv = torch.randn(n_img * N, 3 * H * 64, generator=g).to(dev).to(bf16)
    full_o, full_lse = o.attention_fwd(qkv, n_img, N, H, scale)
    out = torch.full_like(full_o, 7.0); lse = torch.full_like(full_lse, 7.0)
    o.attention_fwd(qkv, n_img, N, H, scale, o=out, lse=lse, q_limit=ql)
    qe = min(N, (ql + 31) // 32 * 32)
    ov, fv = out.view(n_img, N, H * 64), full_o.view(n_img, N, H * 64)
    assert torch.equal(ov[:, :qe], fv[:, :qe]) and torch.equal(lse[:, :, :qe], full_lse[:, :, :qe])
    if qe < N:
        assert float(ov[:, qe:].float().min()) == 7.0 and float(lse[:, :, qe:].min()) == 7.0       # untouched
    d_o = torch.zeros(n_img, N, H * 64, dtype=bf16, device=dev)
    d_o[:, :ql] = torch.randn(n_img, ql, H * 64, generator=g).to(dev).to(bf16)
    d_o = d_o.view(n_img * N, H * 64)
    ref = o.attention_bwd(qkv, full_o, d_o, full_lse, n_img, N, H, scale)
    got = torch.full_like(ref, 5.0)
    o.attention_bwd(qkv, out, d_o, lse, n_img, N, H, scale, dqkv=got, q_limit=ql)       # o / lse as the limited forward left them
    assert torch.equal(got, ref)
    if qe < N:
        assert float(got.view(n_img, N, 3, H * 64)[:, qe:, 0].float().abs().max()) == 0.0


@pytest.mark.parametrize("dt", [bf16, f32])
def test_dropout_kernels(dev, dt):
    """gv_dropout / gv_dropout_add: the counter-based keep mask of (seed, element index) equals the oracle's numpy restatement bit for
    bit, kept values are scaled by 1 / (1 - p); the per-site seed function matches; p near 0 and near 1."""
    from oracle import vit_oracle as vo
    o = ops()
    n = 257 * 384 + 3
    assert o.dropout_site_seed(123456, 7, 2) == vo.dropout_site_seed(123456, 7, 2) and o.dropout_site_seed(1, 0, 0) != o.dropout_site_seed(1, 0, 1)
    for p, seed in ((0.1, 5), (0.5, 0xDEADBEEF), (0.999, 77), (1e-6, 3)):
        x = torch.randn(n, generator=torch.Generator().manual_seed(1)).to(dev).to(dt)
        ref = (x.float().cpu() * vo.dropout_mask(seed, 0, n, p)).to(dt)
        o.dropout(x, seed, p)
        assert torch.equal(x.cpu(), ref), (p, seed)
    rows, cols = 301, 192
    t = torch.randn(rows, cols, generator=torch.Generator().manual_seed(2)).to(dev)
    resid = torch.randn(rows, cols, generator=torch.Generator().manual_seed(3)).to(dev)
    rs = (torch.rand(rows, generator=torch.Generator().manual_seed(4)) * 2).to(dev)
    out = torch.empty_like(t)
    o.dropout_add(t, resid, out, rows, cols, 31337, 0.2, row_scale=rs)
    m = vo.dropout_mask(31337, 0, rows * cols, 0.2).view(rows, cols)
    close(out.cpu(), resid.cpu() + rs.cpu()[:, None] * (t.cpu() * m), 1e-6, 1e-6, "dropout_add")


def test_agc_matches_oracle(dev):
    """gv_agc (--clip-mode agc, timm adaptive_clip_grad) against the oracle's restatement: rows of matrices and conv filters, whole 1-D
    tensors and dim-0-of-one tensors are units; only units whose gradient norm exceeds clip_factor * max(||p||, eps) are rescaled."""
    import math
    from oracle import vit_oracle as vo
    g = torch.Generator().manual_seed(5)
    shapes = {"w": (37, 50), "conv": (6, 3, 4, 4), "b": (45,), "pos": (1, 7, 12), "tiny": (3, 8)}
    params = {k: torch.randn(*sh, generator=g) * (1e-5 if k == "tiny" else 0.2) for k, sh in shapes.items()}         # 'tiny': ||p|| < eps
    grads = {k: torch.randn(*sh, generator=g) * torch.rand(sh[0] if len(sh) > 1 else 1, *([1] * (len(sh) - 1)), generator=g) * 0.5 for k, sh in shapes.items()}
    ref = vo.adaptive_clip_grad(params, {k: v * 0.5 for k, v in grads.items()}, 0.05)          # the oracle sees the scaled (mean) gradient
    offs, o_ = {}, 0
    for k, sh in shapes.items():
        offs[k] = o_; o_ += (math.prod(sh) + 63) // 64 * 64
    P, G = torch.zeros(o_), torch.zeros(o_)
    for k, sh in shapes.items():
        P[offs[k]:offs[k] + math.prod(sh)] = params[k].reshape(-1); G[offs[k]:offs[k] + math.prod(sh)] = grads[k].reshape(-1)
    P, G = P.to(dev), G.to(dev)
    units = ops().agc_units([(offs[k], sh) for k, sh in shapes.items()]).to(dev)
    assert units.shape[0] == 37 + 6 + 1 + 1 + 3
    ops().agc(P, G, units, 0.05, 1e-3, grad_scale=0.5)
    changed = 0
    for k, sh in shapes.items():
        got = G[offs[k]:offs[k] + math.prod(sh)].view(sh).cpu() * 0.5
        close(got, ref[k], 1e-5, 1e-7, f"agc {k}")
        changed += int((ref[k] != grads[k] * 0.5).any())
    assert changed >= 3                                             # the case does clip something


def test_lamb_matches_oracle(dev):
    """gv_lamb (timm --opt lamb: global-norm pre-clip, Adam moments, per-tensor trust ratio on the decayed tensors, EMA copy) against
    the oracle's restatement over an arena of three tensors, three steps."""
    import math
    from oracle import vit_oracle as vo
    g = torch.Generator().manual_seed(21)
    shapes = {"blocks.0.mlp.fc1.weight": (96, 64), "blocks.0.attn.qkv.weight": (40, 64), "blocks.0.mlp.fc1.bias": (96,)}     # two decayed, one not
    params = {k: torch.randn(*sh, generator=g) * 0.3 for k, sh in shapes.items()}
    orc = vo.Lamb({k: v.clone() for k, v in params.items()}, lr=2e-3, wd=0.05)
    spans, off = [], 0
    for k, sh in shapes.items():
        n = (math.prod(sh) + 63) // 64 * 64
        spans.append((off, off + n)); off += n
    flat = lambda d: torch.cat([torch.nn.functional.pad(d[k].reshape(-1), (0, (sp[1] - sp[0]) - d[k].numel())) for k, sp in zip(shapes, spans)])
    P = flat(params).to(dev); M, V, T = torch.zeros_like(P), torch.zeros_like(P), P.clone()
    Pb, Tb = torch.zeros(off, dtype=bf16, device=dev), torch.zeros(off, dtype=bf16, device=dev)
    tab = ops().lamb_block_table(spans, chunk=2048).to(dev)
    tabs = (tab[tab[:, 0] < 2].contiguous(), tab[tab[:, 0] >= 2].contiguous())
    stats, gsq, ws = torch.zeros(6, device=dev), torch.zeros(1, device=dev), torch.empty(1024, device=dev)
    t_ref = {k: v.clone() for k, v in params.items()}
    for step in range(1, 4):
        grads = {k: torch.randn(*sh, generator=g) * (3.0 if step == 1 else 0.05) for k, sh in shapes.items()}      # step 1 triggers the global clip
        orc.step(grads)
        for k in t_ref:
            t_ref[k] = 0.99 * t_ref[k] + 0.01 * orc.p[k]
        G = flat(grads).to(dev)
        ops().sumsq(G, ws, gsq); stats.zero_()
        for phase in (0, 1):
            for tb, wd in zip(tabs, (0.05, 0.0)):
                ops().lamb(P, G, M, V, Pb, T, Tb, tb, stats, gsq, phase=phase, lr=2e-3, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=wd, step=step,
                           teacher_momentum=0.99)
    torch.cuda.synchronize()
    for (k, sh), (lo, hi) in zip(shapes.items(), spans):
        n = math.prod(sh)
        close(P[lo:lo + n].view(sh).cpu(), orc.p[k], 1e-5, 1e-6, f"lamb p {k}")
        close(T[lo:lo + n].view(sh).cpu(), t_ref[k], 1e-5, 1e-6, f"lamb ema {k}")
        close(Pb[lo:lo + n].view(sh).cpu(), orc.p[k], 1e-2, 1e-2, f"lamb bf16 {k}")
        close(M[lo:lo + n].view(sh).cpu(), orc.m[k], 1e-5, 1e-7, f"lamb m {k}")


def test_crop_resize(dev):
    """gv_crop_resize vs the CPU restatement: same float32 arithmetic in the same association -> identical bytes."""
    import numpy as np
    from oracle import augment_oracle as ao
    rng = np.random.default_rng(0)
    tiles = rng.integers(0, 256, (4, 256, 256, 3), dtype=np.uint8)
    boxes = []
    for n in range(24):
        y0, x0, h, w = ao.sample_box(rng, 256, 256, (0.05, 1.0))
        boxes.append((n % 4, y0, x0, h, w, n % 2))
    boxes += [(1, 16, 16, 224, 224, 0), (2, 0, 0, 256, 256, 1), (3, 100, 50, 1, 1, 0), (0, 255, 0, 1, 256, 1)]
    boxes = np.array(boxes, np.int32)
    t_dev, b_dev = torch.from_numpy(tiles).to(dev), torch.from_numpy(boxes).to(dev)
    for out in (96, 224):
        got = ops().crop_resize(t_dev, b_dev, out).cpu().numpy()
        ref = ao.crop_resize(tiles, boxes, out)
        assert np.array_equal(got, ref), (out, int((got != ref).sum()), int(np.abs(got.astype(int) - ref.astype(int)).max()))
    with pytest.raises(L().GipvitError, match="multiple of 4"):
        ops().crop_resize(t_dev, b_dev, 98)


@pytest.mark.parametrize("out", [96, 224])
def test_crop_augment_byte_exact(dev, out):
    """gv_crop_augment (DINO views: random-resized crop + flip, then each crop's own ColorJitter order / grayscale / 3x3 blur /
    solarise, written once) against the PIL-pinned oracle on the same boxes and draws: identical bytes.  With every operation
    off it reproduces gv_crop_resize."""
    import numpy as np
    from gipvit.multicrop import MultiCropSampler, ViewAugmentSampler
    from oracle import augment_oracle as ao
    rng = np.random.default_rng(11)
    B = 6
    tiles = rng.integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)
    tiles[1] = 200                                                   # a flat tile: grey colours, S = 0 in the hue path
    bg, bl = MultiCropSampler(B, 256, 2, 3, seed=2).sample()
    boxes = (bg if out == 224 else bl).numpy()
    vs = ViewAugmentSampler(B, 2, 3, jitter_p=0.9, gray_p=0.3, blur_p=(1.0, 0.5, 0.5), solar_p=(0.3, 0.5, 0.3), seed=4)
    rec = vs.sample_host()[0 if out == 224 else 1]
    rec["n_color"][0] = 0; rec["gray"][0] = 0; rec["blur"][0] = 0; rec["solar"][0] = -1       # one crop with everything off
    t_dev, b_dev = torch.from_numpy(tiles).to(dev), torch.from_numpy(boxes).to(dev)
    stats = torch.zeros(len(boxes), dtype=torch.int64, device=dev)
    got = ops().crop_augment(t_dev, b_dev, ViewAugmentSampler.pack(rec).to(dev), stats, out).cpu().numpy()
    ref = ao.crop_views(tiles, boxes, out, ViewAugmentSampler.to_dicts(rec))
    bad = [(i, int((got[i] != ref[i]).sum()), int(np.abs(got[i].astype(int) - ref[i].astype(int)).max())) for i in range(len(boxes)) if not np.array_equal(got[i], ref[i])]
    assert not bad, (bad[:4], ViewAugmentSampler.to_dicts(rec)[bad[0][0]])
    assert np.array_equal(got[0], ops().crop_resize(t_dev, b_dev[:1].contiguous(), out).cpu().numpy()[0])
    assert any(r["gray"] for r in rec) and any(r["solar"] >= 0 for r in rec) and any(r["blur"] for r in rec) and any(r["n_color"] for r in rec)


@pytest.mark.parametrize("recipe", ["flip", "rvf", "cbnfrsc", "pcbnfrs", "cbnfr", "bnfrsc", "frs", "aug_receptornet"])
def test_augment_byte_exact(dev, recipe):
    """gv_augment against oracle/augment_oracle.py on the same draws: byte-identical tiles for every recipe of
    transformations.py:131-197 (colour jitter in a random order, noise, flips / rotations, zoom, cutout)."""
    import numpy as np
    from gipvit import augment as A
    from oracle import augment_oracle as ao
    n, size = 6, 64
    rng = np.random.default_rng(3)
    tiles = rng.integers(0, 256, (n, size, size, 3), dtype=np.uint8)
    tiles[0] = 200; tiles[1, :, :, :] = tiles[1, :, :, :1]               # a flat tile and a grey tile (hue / saturation corner cases)
    aug = A.TileAugmenter(recipe, size, color_param=0.15, seed=5)
    ps = [aug.sample_one() for _ in range(n)]
    out, fill = aug.apply(torch.from_numpy(tiles).to(dev), params=ps)
    torch.cuda.synchronize()
    z = A.normal_table().numpy()
    for i, p in enumerate(ps):
        ref = ao.augment_tile(tiles[i], p, z)
        got = out[i].cpu().numpy()
        assert np.array_equal(got, ref), (recipe, i, int((got != ref).sum()), p)
    assert (fill is not None) == any(p["fill"] for p in ps)


def test_augment_blur_and_fill(dev):
    """the 3x3 Gaussian blur with a real kernel (the recipes' sigma <= 0.1 makes it the identity) and the normalised fill
    boxes gv_patchify applies (Cutout after Normalize, MeanPixelRegularization)."""
    import numpy as np
    from gipvit import augment as A
    from oracle import augment_oracle as ao, vit_oracle as vo
    rng = np.random.default_rng(4)
    tiles = rng.integers(0, 256, (3, 64, 64, 3), dtype=np.uint8)
    aug = A.TileAugmenter("cbnfr", 64, seed=1)
    ps = []
    for sigma in (0.8, 2.0, 0.35):
        p = aug.sample_one(); p["blur"] = A.blur_weights(sigma); ps.append(p)
    out, _ = aug.apply(torch.from_numpy(tiles).to(dev), params=ps)
    z = A.normal_table().numpy()
    for i, p in enumerate(ps):
        assert np.array_equal(out[i].cpu().numpy(), ao.augment_tile(tiles[i], p, z)), i
    # fill boxes in patchify: box -> the given normalised value, elsewhere (u8 / 255 - mean) / std
    t = torch.from_numpy(tiles[:2]).to(dev)
    fill = torch.tensor([[10, 40, 5, 33, 0.0, 0.0, 0.0, 1.0], [0, 64, 0, 64, 0.25, -0.5, 1.5, 1.0]], dtype=f32, device=dev)
    got = ops().patchify(t, [(0, 0)], 64, vo.MEAN_RON, vo.STD_RON, fill=fill).float().cpu().view(2, 4, 4, 3, 16, 16)
    ref = vo.normalize_window(torch.from_numpy(tiles[:2]), (0, 0, 64))                      # [2, 3, 64, 64]
    ref[0, :, 10:40, 5:33] = 0.0
    ref[1] = torch.tensor([0.25, -0.5, 1.5]).view(3, 1, 1)
    ref_p = ref.view(2, 3, 4, 16, 4, 16).permute(0, 2, 4, 1, 3, 5)                          # [img, prow, pcol, c, py, px]
    close(got, ref_p, 8e-3, 8e-3, "patchify fill")


def test_stochastic_depth_row_scales(dev):
    """gv_expand_rows, the RESID row factor of gv_linear, gv_linear_ln_fwd's row_scale and the gb_scale of both LayerNorm
    backward forms."""
    o, l = ops(), L()
    g = torch.Generator().manual_seed(12)
    # expand: two segments (3 images x 5 tokens, 2 images x 3 tokens), 4 branches
    row_img = torch.tensor([0] * 5 + [1] * 5 + [2] * 5 + [3] * 3 + [4] * 3, dtype=torch.int32, device=dev)
    per = torch.rand(4, 5, generator=g).to(dev)
    rows = torch.empty(4, 21, device=dev)
    o.expand_rows(per, row_img, rows, 4, 5, 21)
    assert torch.equal(rows, per[:, row_img.long()])
    # gv_linear RESID with a row factor (exact on integers; factors 0 / 2)
    M, N, K = 300, 192, 128
    A, B = ints((M, K), dev, seed=41), ints((N, K), dev, seed=42)
    bias = (torch.arange(N, device=dev) % 3).float(); resid = torch.randint(-4, 5, (M, N), generator=g).float().to(dev)
    rs = (torch.randint(0, 2, (M,), generator=g).float() * 2).to(dev)
    C = torch.empty(M, N, device=dev)
    o.linear(A, B, C, M, N, K, epilogue=l.EPI_BIAS | l.EPI_RESID, bias=bias, resid=resid, row_scale=rs)
    assert torch.equal(C, (A.float() @ B.float().t() + bias) * rs[:, None] + resid)
    # fused Linear + LayerNorm forward
    M, K = 2000, 384
    A, W = ints((M, K), dev, seed=43), ints((384, K), dev, seed=44)
    resid = torch.randint(-4, 5, (M, 384), generator=g).float().to(dev)
    rs = (torch.randint(0, 2, (M,), generator=g).float() * 2).to(dev)
    out = torch.empty(M, 384, device=dev)
    o.linear_ln_fwd(A, W, out, M, K, resid=resid, row_scale=rs)
    assert torch.equal(out, (A.float() @ W.float().t()) * rs[:, None] + resid)
    # LayerNorm backward: g unscaled, gb = bf16(g * s), third column sum over the scaled rows -- stand-alone and fused forms
    rows_n, D = 1000, 384
    x = torch.randn(rows_n, D, generator=g).to(dev); gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev); beta = torch.zeros(D, device=dev)
    y, mean, rstd = o.layernorm_fwd(x, gamma, beta, rows_n, D)
    dy = torch.randn(rows_n, D, generator=g).to(dev).to(bf16)
    sc = (torch.randint(0, 2, (rows_n,), generator=g).float() * 1.25).to(dev)
    outs = []
    for s_ in (None, sc):
        gbuf = torch.zeros(rows_n, D, device=dev); gb = torch.empty(rows_n, D, dtype=bf16, device=dev)
        parts = torch.zeros(l.LN_PARTIAL_BLOCKS, 3, D, device=dev)
        o.layernorm_bwd(dy, x, mean, rstd, gamma, gbuf, gb, parts, rows_n, D, g_init=True, gb_scale=s_)
        d2 = torch.zeros(D, device=dev); o.ln_finalize(parts, l.LN_PARTIAL_BLOCKS, D, None, None, d2)
        outs.append((gbuf, gb, d2))
    (g0, gb0, s0), (g1, gb1, s1) = outs
    assert torch.equal(g0, g1) and torch.equal(gb1.float(), (g0 * sc[:, None]).to(bf16).float())
    close(s1, (g0 * sc[:, None]).sum(0), 1e-3, 1e-3, "scaled column sum")
    Kd = 256
    dY = ints((rows_n, Kd), dev, seed=45); Wb = ints((Kd, D), dev, seed=46)
    outs = []
    for s_ in (None, sc):
        gbuf = torch.zeros(rows_n, D, device=dev); gb = torch.empty(rows_n, D, dtype=bf16, device=dev)
        parts = torch.zeros(l.LN_PARTIAL_BLOCKS, 3, D, device=dev)
        nb = o.linear_ln_bwd(dY, Wb, x, mean, rstd, gamma, gbuf, gb, parts, rows_n, Kd, g_init=True, gb_scale=s_)
        d2 = torch.zeros(D, device=dev); o.ln_finalize(parts, nb, D, None, None, d2)
        outs.append((gbuf, gb, d2))
    (g0, gb0, s0), (g1, gb1, s1) = outs
    assert torch.equal(g0, g1) and torch.equal(gb1.float(), (g0 * sc[:, None]).to(bf16).float())
    close(s1, (g0 * sc[:, None]).sum(0), 1e-3, 1e-3, "scaled column sum (fused)")
