"""Race screen for the counted-wait kernels (ping-pong dW, ping-pong full-row k-loop): the exact integer checks of
tests/test_kernels_gpu.py repeated many times on fresh data, with other work on a second stream to perturb timing.
python tools/stress_exact.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit import ops as o, _lib as L

dev = torch.device("cuda:0")
bf16, f32 = torch.bfloat16, torch.float32
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ws = torch.empty(L.lib.gv_linear_workspace_bytes() // 4, dtype=f32, device=dev)
side = torch.cuda.Stream(dev)
noise = torch.empty(64 << 20, dtype=f32, device=dev)


def ints(shape, seed, lo=-2, hi=3):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).to(dev).to(bf16)


bad = 0
for it in range(rounds):
    with torch.cuda.stream(side):                   # HBM / L2 traffic beside the kernels under test
        noise.mul_(1.0001)
    # ---- wide full-row products (NT and NN), ragged M
    for (M, N, K) in ((2500 + 16 * it, 1152, 384), (44160, 1536, 384), (3000, 384, 1536)):
        A, B = ints((M, K), 1000 + it), ints((N, K), 2000 + it)
        ref = (A.float() @ B.float().t()).to(bf16)
        C = torch.empty(M, N, dtype=bf16, device=dev)
        o.linear(A, B, C, M, N, K)
        o.linear(A, B, C, M, N, K, epilogue=L.EPI_BIAS, bias=torch.zeros(N, device=dev))
        if not torch.equal(C, ref): bad += 1; print("wide NT mismatch", it, M, N, K, int((C != ref).sum()))
        o.linear(A, B.t().contiguous(), C, M, N, K, trans_b=True)
        if not torch.equal(C, ref): bad += 1; print("wide NN mismatch", it, M, N, K, int((C != ref).sum()))
    # ---- fused Linear + LayerNorm forward: the f32 row is exact on integer operands
    for (M, K) in ((44160, 1536), (25216, 384), (2000 + 8 * it, 1152)):
        A, W = ints((M, K), 3000 + it), ints((384, K), 4000 + it)
        resid = torch.randint(-8, 8, (M, 384), generator=torch.Generator().manual_seed(it)).float().to(dev)
        out = torch.empty(M, 384, device=dev)
        o.linear_ln_fwd(A, W, out, M, K, resid=resid)
        ref = A.float() @ W.float().t() + resid
        if not torch.equal(out, ref): bad += 1; print("ln_fwd mismatch", it, M, K, int((out != ref).sum()))
    # ---- grouped weight gradients
    T = 44160 if it % 4 == 0 else 3000 + 64 * it
    probs, refs = [], []
    for q, (Mq, Nq) in enumerate(((384, 1536), (1536, 384), (384, 384), (1152, 384))):
        dY, X = ints((T, Mq), 5000 + 7 * it + q, -1, 2), ints((T, Nq), 6000 + 7 * it + q, -1, 2)
        probs.append((dY, X, torch.zeros(Mq, Nq, device=dev), torch.zeros(Mq, device=dev)))
        refs.append((dY.float().t() @ X.float(), dY.float().sum(0)))
    o.linear_dw_group(probs, T, ws)
    for (dY, X, dW, cs), (rw, rc) in zip(probs, refs):
        if not (torch.equal(dW, rw) and torch.equal(cs, rc)): bad += 1; print("dW group mismatch", it, T, tuple(dW.shape), int((dW != rw).sum()))
    torch.cuda.synchronize()
    if it % 10 == 9: print(f"round {it + 1}: {bad} mismatches so far", flush=True)
print("stress_exact:", "CLEAN" if bad == 0 else f"{bad} MISMATCHES", f"({rounds} rounds)")
sys.exit(1 if bad else 0)
