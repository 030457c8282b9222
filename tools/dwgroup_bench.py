"""One transformer block's weight gradients as the engine issues them (gv_linear_dw_group, ViT-S shapes, K = 44160 tokens).
python tools/dwgroup_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit import ops as o, _lib as L

dev = torch.device("cuda:0")
bf16 = torch.bfloat16
K = int(sys.argv[1]) if len(sys.argv) > 1 else 44160
ws = torch.empty(L.lib.gv_linear_workspace_bytes() // 4, device=dev)
g = torch.Generator().manual_seed(1)
probs = []
for (M, N, cs) in ((384, 1536, False), (1536, 384, True), (384, 384, False), (1152, 384, True)):
    dY = torch.randn(K, M, generator=g).to(dev).to(bf16); X = torch.randn(K, N, generator=g).to(dev).to(bf16)
    probs.append((dY, X, torch.zeros(M, N, device=dev), torch.zeros(M, device=dev) if cs else None))
fl = sum(2.0 * p[0].shape[1] * p[1].shape[1] * K for p in probs)
for _ in range(3):
    o.linear_dw_group(probs, K, ws)
torch.cuda.synchronize()
reps = 20
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
ev[0].record()
for i in range(reps):
    o.linear_dw_group(probs, K, ws); ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
print(f"dW group K={K}: {ts[len(ts) // 2]:6.1f} us (min {ts[0]:6.1f}) = {fl / ts[len(ts) // 2] / 1e6:5.0f} TF incl. reduce", flush=True)
