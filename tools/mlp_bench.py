"""Timing of the fused MLP forward (gv_mlp_ln_fwd) against the unfused pair on the teacher's and the student's row counts.
python tools/mlp_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit import ops as o, _lib as L
dev = torch.device("cuda:0"); bf16 = torch.bfloat16; f32 = torch.float32


def timeit(fn, reps=30):
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


g = torch.Generator().manual_seed(1)
D, Hd = 384, 1536
for M in (25216, 44160, 12608):
    A = torch.randn(M, D, generator=g).to(dev).to(bf16)
    W1 = (0.05 * torch.randn(Hd, D, generator=g)).to(dev).to(bf16); W2 = (0.03 * torch.randn(D, Hd, generator=g)).to(dev).to(bf16)
    b1, b2 = torch.randn(Hd, generator=g).to(dev), torch.randn(D, generator=g).to(dev)
    resid = torch.randn(M, D, generator=g).to(dev)
    gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    out = torch.empty(M, D, dtype=f32, device=dev); y = torch.empty(M, D, dtype=bf16, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    h = torch.empty(M, Hd, dtype=bf16, device=dev)
    kw = dict(bias2=b2, resid=resid, gamma=gam, beta=bet, y=y, mean=mean, rstd=rstd)
    t_f = timeit(lambda: o.mlp_ln_fwd(A, W1, b1, W2, out, M, D, Hd, **kw))
    t_1 = timeit(lambda: o.linear(A, W1, h, M, Hd, D, epilogue=L.EPI_BIAS | L.EPI_GELU, bias=b1))
    t_2 = timeit(lambda: o.linear_ln_fwd(h, W2, out, M, Hd, bias=b2, resid=resid, gamma=gam, beta=bet, y=y, mean=mean, rstd=rstd))

    def pair():
        o.linear(A, W1, h, M, Hd, D, epilogue=L.EPI_BIAS | L.EPI_GELU, bias=b1)
        o.linear_ln_fwd(h, W2, out, M, Hd, bias=b2, resid=resid, gamma=gam, beta=bet, y=y, mean=mean, rstd=rstd)
    t_p = timeit(pair)
    fl = 4.0 * M * Hd * D
    print(f"M {M}: fused {t_f[0]:6.1f} us (min {t_f[1]:6.1f}) = {fl / t_f[0] / 1e6:5.0f} TF | fc1 + GELU {t_1[0]:6.1f}, fc2 + LN {t_2[0]:6.1f}, pair {t_p[0]:6.1f} us", flush=True)
