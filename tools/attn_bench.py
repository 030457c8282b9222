"""Timing of the fused attention kernels on the step's shapes (global crops: 128 images x 197 tokens, local crops: 512 x 37;
6 heads).  python tools/attn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit import ops as o

dev = torch.device("cuda:0")
bf16 = torch.bfloat16
H = 6


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


for n_img, N in ((128, 197), (512, 37), (256, 197)):
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(n_img * N, 3 * H * 64, generator=g).to(dev).to(bf16)
    d_o = torch.randn(n_img * N, H * 64, generator=g).to(dev).to(bf16)
    out, lse = o.attention_fwd(qkv, n_img, N, H, 0.125)
    dqkv = torch.empty_like(qkv)
    tf = timeit(lambda: o.attention_fwd(qkv, n_img, N, H, 0.125, o=out, lse=lse))
    tb = timeit(lambda: o.attention_bwd(qkv, out, d_o, lse, n_img, N, H, 0.125, dqkv=dqkv))
    pairs = n_img * H
    fb = pairs * N * 64 * 2 * 4 + pairs * N * 4            # fwd bytes: q k v in, o out, lse
    bb = pairs * N * 64 * 2 * 8 + pairs * N * 4            # bwd: q k v o do in, dq dk dv out
    print(f"n_img {n_img} N {N}: fwd {tf[0]:6.1f} us (min {tf[1]:6.1f}) = {fb / tf[0] / 1e6:5.2f} TB/s, {4.0 * pairs * N * N * 64 / tf[0] / 1e6:5.0f} TF | "
          f"bwd {tb[0]:6.1f} us (min {tb[1]:6.1f}) = {bb / tb[0] / 1e6:5.2f} TB/s, {10.0 * pairs * N * N * 64 / tb[0] / 1e6:5.0f} TF", flush=True)
