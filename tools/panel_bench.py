"""A/B timing of the fused full-row kernels (gv_linear_ln_fwd / _bwd) against the round-1 pair (128x128-tile gv_linear +
stand-alone LayerNorm) on the step's shapes, interleaved in one process.  python tools/panel_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit import ops as o, _lib as L

dev = torch.device("cuda:0")
bf16, f32 = torch.bfloat16, torch.float32
N = 384


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
    return ts[len(ts) // 2], ts[0]


for M in (44160, 25216):
    for K in (384, 1536, 1152):
        g = torch.Generator().manual_seed(1)
        A = torch.randn(M, K, generator=g).to(dev).to(bf16)
        W = (0.05 * torch.randn(N, K, generator=g)).to(dev).to(bf16)
        Wt = W.t().contiguous()
        bias = torch.randn(N, generator=g).to(dev); gamma = torch.ones(N, device=dev); beta = torch.zeros(N, device=dev)
        resid = torch.randn(M, N, generator=g).to(dev)
        out = torch.empty(M, N, device=dev); y = torch.empty(M, N, dtype=bf16, device=dev)
        mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
        fl = 2.0 * M * N * K
        if K != 1152:
            def old():
                o.linear(A, W, out, M, N, K, epilogue=L.EPI_BIAS | L.EPI_RESID, bias=bias, resid=resid)
                o.layernorm_fwd(out, gamma, beta, M, N, y=y, mean=mean, rstd=rstd)
            def old_gemm():
                o.linear(A, W, out, M, N, K, epilogue=L.EPI_BIAS | L.EPI_RESID, bias=bias, resid=resid)
            def new():
                o.linear_ln_fwd(A, W, out, M, K, bias=bias, resid=resid, gamma=gamma, beta=beta, y=y, mean=mean, rstd=rstd)
            t_old, t_g, t_new = timeit(old), timeit(old_gemm), timeit(new)
            print(f"fwd M={M} K={K}: old gemm+ln {t_old[0]:7.1f} us (gemm alone {t_g[0]:6.1f} = {fl / t_g[0] / 1e6:6.0f} TF)   fused {t_new[0]:7.1f} us "
                  f"(min {t_new[1]:6.1f}) = {fl / t_new[0] / 1e6:6.0f} TF   x{t_old[0] / t_new[0]:.2f}", flush=True)
        if K != 384 or True:
            dY = torch.randn(M, K, generator=g).to(dev).to(bf16)
            Wb = (0.05 * torch.randn(K, N, generator=g)).to(dev).to(bf16)
            x = torch.randn(M, N, generator=g).to(dev)
            gbuf = torch.zeros(M, N, device=dev); gb = torch.empty(M, N, dtype=bf16, device=dev); dxn = torch.empty(M, N, dtype=bf16, device=dev)
            parts = torch.empty(L.LN_PARTIAL_BLOCKS, 3, N, device=dev)
            o.layernorm_fwd(x, gamma, beta, M, N, y=y, mean=mean, rstd=rstd)
            def oldb():
                o.linear(dY, Wb, dxn, M, N, K, trans_b=True)
                o.layernorm_bwd(dxn, x, mean, rstd, gamma, gbuf, gb, parts, M, N)
            def oldb_g():
                o.linear(dY, Wb, dxn, M, N, K, trans_b=True)
            def newb():
                o.linear_ln_bwd(dY, Wb, x, mean, rstd, gamma, gbuf, gb, parts, M, K)
            t_old, t_g, t_new = timeit(oldb), timeit(oldb_g), timeit(newb)
            print(f"bwd M={M} K={K}: old gemm+ln {t_old[0]:7.1f} us (gemm alone {t_g[0]:6.1f} = {fl / t_g[0] / 1e6:6.0f} TF)   fused {t_new[0]:7.1f} us "
                  f"(min {t_new[1]:6.1f}) = {fl / t_new[0] / 1e6:6.0f} TF   x{t_old[0] / t_new[0]:.2f}", flush=True)
