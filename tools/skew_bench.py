"""Lab: does the relative placement of a kernel's output streams matter?  fc1 writes two [M, 1536] bf16 buffers (h and the saved
pre-activation); here they are carved out of one allocation at a chosen byte skew.   python tools/skew_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit import ops as o, _lib as L
dev = torch.device("cuda:0"); bf16 = torch.bfloat16
def timeit(fn, reps=60):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
    return ts[len(ts) // 2], ts[0], ts[-1], ts[len(ts) // 4], ts[3 * len(ts) // 4]
g = torch.Generator().manual_seed(1)
M, N, K = 44160, 1536, 384
A = torch.randn(M, K, generator=g).to(dev).to(bf16)
W = (0.05 * torch.randn(N, K, generator=g)).to(dev).to(bf16)
bias = torch.randn(N, generator=g).to(dev)
nb = M * N * 2
pool = torch.empty(2 * nb + (64 << 20), dtype=torch.uint8, device=dev)
base = (-pool.data_ptr()) % (2 << 20)                    # 2-MB aligned start
for skew in (0, 256, 4096, 65536, (1 << 20) + 4096, (2 << 20), (8 << 20) + 12288):
    C = pool[base: base + nb].view(bf16).view(M, N)
    off = base + ((nb + (2 << 20) - 1) // (2 << 20)) * (2 << 20) + skew
    aux = pool[off: off + nb].view(bf16).view(M, N)
    t = timeit(lambda: o.linear(A, W, C, M, N, K, epilogue=L.EPI_BIAS | L.EPI_GELU | L.EPI_SAVE_PRE, bias=bias, aux_out=aux))
    print(f"skew {skew:9d} B: median {t[0]:6.1f}  min {t[1]:6.1f}  q1 {t[3]:6.1f}  q3 {t[4]:6.1f}  max {t[2]:6.1f} us", flush=True)
