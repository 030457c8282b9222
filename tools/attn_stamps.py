#!/usr/bin/env python3
"""s_memtime stamps of the attention backward (lab build: tools/lab.sh attn_stamps attention "-DGV_ATTN_STAMPS"), wave 0 of the first
64 workgroups, N = 197 x 128 images x 6 heads: where a (image, head) pair's time goes.
    GIPVIT_LIB=tools/lab_build/lib_attn_stamps.so python tools/attn_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit import ops as o, _lib as L
dev, bf16, H = torch.device("cuda:0"), torch.bfloat16, 6
for n_img, N in ((128, 197), (512, 37)):
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(n_img * N, 3 * H * 64, generator=g).to(dev).to(bf16)
    d_o = torch.randn(n_img * N, H * 64, generator=g).to(dev).to(bf16)
    out, lse = o.attention_fwd(qkv, n_img, N, H, 0.125)
    dqkv = torch.empty_like(qkv)
    for _ in range(3): o.attention_bwd(qkv, out, d_o, lse, n_img, N, H, 0.125, dqkv=dqkv)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (64 * 16))()
    rc = L.lib.gv_lab_attn_stamps(buf)
    assert rc == 0, rc
    st = torch.tensor(list(buf), dtype=torch.float64).view(64, 16)
    nstep = (N + 63) // 64 if N > 64 else (N + 31) // 32
    names = ["staged (Q, K, dO, V in)", "delta / lse in LDS"]
    idx = [1, 2]
    for q in range(nstep):
        names += [f"step {q}: phase A", f"step {q}: barrier wait", f"step {q}: phase B"]; idx += [3 + 3 * q, 4 + 3 * q, 5 + 3 * q]
    names.append("dQ zero rows + dK / dV stores retired"); idx.append(15)
    prev = st[:, 0]
    print(f"N = {N}, {n_img} images: wave 0 of workgroups 0..63, mean cycles (s_memtime ticks) per segment; total {float((st[:, 15] - st[:, 0]).mean()):.0f}")
    for nm, i in zip(names, idx):
        d = st[:, i] - prev
        print(f"  {nm:42s} {float(d.mean()):9.0f}  (min {float(d.min()):.0f}, max {float(d.max()):.0f})")
        prev = st[:, i]
