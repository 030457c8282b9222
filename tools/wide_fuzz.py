"""Randomised exact check of the wide full-row products (integer operands: bf16 results must equal torch's, bit for bit) over row counts
around every panel / group boundary.   python tools/wide_fuzz.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gipvit import ops as o, _lib as L
dev = torch.device("cuda:0"); bf16 = torch.bfloat16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for it in range(n):
    N = 384 * int(rng.integers(1, 9)); K = 128 * int(rng.integers(1, 5))
    base = int(rng.choice([2048, 2048 + 64, 176 * 13, 192 * 40, 112 * 57, 44160, 25216, 60000, 3 * 12288]))
    M = max(2048, base + int(rng.integers(-70, 71)))
    tb = bool(rng.integers(0, 2))
    A = torch.randint(-2, 3, (M, K), generator=torch.Generator().manual_seed(it)).to(dev).to(bf16)
    B = torch.randint(-2, 3, (N, K), generator=torch.Generator().manual_seed(1000 + it)).to(dev).to(bf16)
    ref = (A.float() @ B.float().t()).to(bf16)
    C = torch.full((M + 2, N), 5.0, dtype=bf16, device=dev)
    if tb:
        o.linear(A, B.t().contiguous(), C, M, N, K, trans_b=True)
    else:
        o.linear(A, B, C, M, N, K, epilogue=L.EPI_BIAS, bias=torch.zeros(N, device=dev))
    ok = torch.equal(C[:M], ref) and float(C[M:].float().min()) == 5.0
    if not ok:
        bad += 1; print("MISMATCH", M, N, K, tb, int((C[:M] != ref).sum()), flush=True)
print(f"wide_fuzz: {n} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
