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
