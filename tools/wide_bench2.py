"""K sweep of the wide products (lab): python tools/wide_bench2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit import ops as o, _lib as L
dev = torch.device("cuda:0"); bf16 = torch.bfloat16
def timeit(fn, reps=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
    return ts[len(ts) // 2], ts[0]
g = torch.Generator().manual_seed(1)
M = 44160
for N, epi, name in ((1536, L.EPI_BIAS | L.EPI_GELU | L.EPI_SAVE_PRE, "fc1"), (1536, L.EPI_BIAS | L.EPI_GELU, "fc1t"), (1152, L.EPI_BIAS, "qkv"), (384, L.EPI_BIAS, "n384")):
    out = []
    for K in (128, 256, 384, 768):
        A = torch.randn(M, K, generator=g).to(dev).to(bf16)
        W = (0.05 * torch.randn(N, K, generator=g)).to(dev).to(bf16)
        bias = torch.randn(N, generator=g).to(dev)
        C = torch.empty(M, N, dtype=bf16, device=dev); aux = torch.empty(M, N, dtype=bf16, device=dev)
        t = timeit(lambda: o.linear(A, W, C, M, N, K, epilogue=epi, bias=bias, aux_out=aux if epi & L.EPI_SAVE_PRE else None))
        out.append(f"K{K}: {t[0]:6.1f}/{t[1]:6.1f}")
    print(f"{name:5s} N {N:5d}  " + "  ".join(out), flush=True)
