"""Where does the full-row kernel's k-loop wait?  Same launch with the A rows (lda = 0: every row is row 0, L2-resident) or
the W rows (ldw = 0) collapsed: the arithmetic and instruction stream are unchanged, only where the bytes come from."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit import ops as o, _lib as L
dev = torch.device("cuda:0"); bf16 = torch.bfloat16; N = 384

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps): fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
    return ts[len(ts) // 2]

for M, K in ((44160, 1536), (44160, 384), (25216, 1536)):
    A = torch.randn(M, K).to(dev).to(bf16); W = (0.05 * torch.randn(N, K)).to(dev).to(bf16)
    resid = torch.randn(M, N).to(dev); out = torch.empty(M, N, device=dev); y = torch.empty(M, N, dtype=bf16, device=dev)
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev); gamma = torch.ones(N, device=dev); beta = torch.zeros(N, device=dev); bias = torch.zeros(N, device=dev)
    def run(lda, ldw, with_resid=True, with_ln=True):
        a = L.gv_linear_ln_fwd_args(A.data_ptr(), W.data_ptr(), M, N, K, lda, ldw, bias.data_ptr(), resid.data_ptr() if with_resid else None, N,
                                    out.data_ptr(), N, gamma.data_ptr() if with_ln else None, beta.data_ptr(), 1e-6, y.data_ptr(), mean.data_ptr(), rstd.data_ptr())
        L.call("gv_linear_ln_fwd", a, torch.cuda.current_stream().cuda_stream)
    print(f"M={M} K={K}: full {timeit(lambda: run(K, K)):6.1f}  A-from-L2 {timeit(lambda: run(0, K)):6.1f}  W-one-row {timeit(lambda: run(K, 0)):6.1f}  "
          f"both {timeit(lambda: run(0, 0)):6.1f}  no-resid {timeit(lambda: run(K, K, False)):6.1f}  no-ln {timeit(lambda: run(K, K, True, False)):6.1f} us", flush=True)
