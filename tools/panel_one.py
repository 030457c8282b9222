"""One shape of the fused full-row kernel, a few launches (for rocprofv3 --pmc).  python tools/panel_one.py [M K mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit import ops as o, _lib as L
M = int(sys.argv[1]) if len(sys.argv) > 1 else 44160
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
dev = torch.device("cuda:0"); bf16 = torch.bfloat16; N = 384
A = torch.randn(M, K).to(dev).to(bf16); W = (0.05 * torch.randn(N, K)).to(dev).to(bf16)
resid = torch.randn(M, N).to(dev); out = torch.empty(M, N, device=dev); y = torch.empty(M, N, dtype=bf16, device=dev)
mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev); gamma = torch.ones(N, device=dev); beta = torch.zeros(N, device=dev); bias = torch.zeros(N, device=dev)
for _ in range(6):
    o.linear_ln_fwd(A, W, out, M, K, bias=bias, resid=resid, gamma=gamma, beta=beta, y=y, mean=mean, rstd=rstd)
    o.linear(A, W, out, M, N, K, epilogue=L.EPI_BIAS | L.EPI_RESID, bias=bias, resid=resid)
torch.cuda.synchronize()
