#!/usr/bin/env python3
"""gv_dino_loss at the headline shape (B = 64, V = 10, G = 2, K = 65536): time per call (two memsets + row_stats + loss_grad)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gipvit import ops as o
dev = torch.device("cuda:0")
B, V, G, K = 64, 10, 2, 65536
g = torch.Generator(device="cpu").manual_seed(0)
s = (torch.randn(V * B, K, generator=g) * 2).to(dev)
t = (torch.randn(G * B, K, generator=g) * 2).to(dev)
center = torch.zeros(K, device=dev)
ds = torch.empty(V * B, K, dtype=torch.bfloat16, device=dev)
loss = torch.zeros(1, device=dev); csum = torch.zeros(K, device=dev); ws = torch.zeros(2 * (V + G) * B, device=dev)
for _ in range(5): o.dino_loss(s, t, center, ds, loss, csum, ws, B, V, G, K, 0.1, 0.04)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): o.dino_loss(s, t, center, ds, loss, csum, ws, B, V, G, K, 0.1, 0.04)
e1.record(); torch.cuda.synchronize()
print(f"GIPVIT_RS_VAR={os.environ.get('GIPVIT_RS_VAR', '0')}: {e0.elapsed_time(e1) * 20:.1f} us per call, loss {float(loss):.5f}")
