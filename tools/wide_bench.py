"""Timing of gv_linear's wide bf16 products (full-row kernel, csrc/panel.hip MODE_WIDE) on the step's shapes.
python tools/wide_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit import ops as o, _lib as L

dev = torch.device("cuda:0")
bf16 = torch.bfloat16


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
for M in (44160, 25216):
    A = torch.randn(M, 384, generator=g).to(dev).to(bf16)
    for name, N, tb, epi in (("qkv", 1152, False, L.EPI_BIAS), ("fc1", 1536, False, L.EPI_BIAS | L.EPI_GELU | L.EPI_SAVE_PRE),
                             ("fc1 teacher", 1536, False, L.EPI_BIAS | L.EPI_GELU), ("gelu' dX", 1536, True, L.EPI_DGELU), ("proj dX", 384, True, 0)):
        W = (0.05 * torch.randn(N, 384, generator=g)).to(dev).to(bf16)
        Wk = W.t().contiguous() if tb else W
        bias = torch.randn(N, generator=g).to(dev)
        C = torch.empty(M, N, dtype=bf16, device=dev); aux = torch.randn(M, N, generator=g).to(dev).to(bf16)
        kw = dict(trans_b=tb, epilogue=epi)
        if epi & L.EPI_BIAS: kw["bias"] = bias
        if epi & L.EPI_SAVE_PRE: kw["aux_out"] = aux
        if epi & L.EPI_DGELU: kw["aux_in"] = aux
        t = timeit(lambda: o.linear(A, Wk, C, M, N, 384, **kw))
        by = M * 384 * 2 + M * N * 2 * (2 if epi & (L.EPI_SAVE_PRE | L.EPI_DGELU) else 1)
        print(f"M {M} {name:12s} N {N}: {t[0]:6.1f} us (min {t[1]:6.1f}) = {by / t[0] / 1e6:5.2f} TB/s, {2.0 * M * N * 384 / t[0] / 1e6:5.0f} TF", flush=True)

# k-loop vs epilogue: the same output with growing K (slope = time per 64-deep k-step of all column blocks, intercept = epilogues)
M = 44160
for N, epi, name in ((1536, L.EPI_BIAS | L.EPI_GELU | L.EPI_SAVE_PRE, "fc1-like"), (1152, L.EPI_BIAS, "qkv-like")):
    for K in (128, 384, 768, 1536):
        A = torch.randn(M, K, generator=g).to(dev).to(bf16)
        W = (0.05 * torch.randn(N, K, generator=g)).to(dev).to(bf16)
        bias = torch.randn(N, generator=g).to(dev)
        C = torch.empty(M, N, dtype=bf16, device=dev); aux = torch.empty(M, N, dtype=bf16, device=dev)
        t = timeit(lambda: o.linear(A, W, C, M, N, K, epilogue=epi, bias=bias, aux_out=aux if epi & L.EPI_SAVE_PRE else None))
        print(f"{name} N {N} K {K:5d}: {t[0]:6.1f} us (min {t[1]:6.1f})  {2.0 * M * N * K / t[0] / 1e6:5.0f} TF", flush=True)
