"""Timing of the weight-gradient products (gv_linear trans_a, trans_b, ACCUM + workspace) on the step's shapes:
per launch with the split-K reduce (events around the call) and the product kernel alone (gv_linear_timing rows).
python tools/dw_bench.py [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit import ops as o, _lib as L

dev = torch.device("cuda:0")
bf16 = torch.bfloat16
K = int(sys.argv[1]) if len(sys.argv) > 1 else 44160
ws = torch.empty(L.lib.gv_linear_workspace_bytes() // 4, device=dev)


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


SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]] or [(1152, 384), (384, 384), (1536, 384), (384, 1536)]
for (M, N) in SHAPES:
    g = torch.Generator().manual_seed(1)
    A = torch.randn(K, M, generator=g).to(dev).to(bf16)
    B = torch.randn(K, N, generator=g).to(dev).to(bf16)
    C = torch.zeros(M, N, device=dev)
    cs = torch.zeros(M, device=dev)
    def run():
        o.linear(A, B, C, M, N, K, trans_a=True, trans_b=True, epilogue=L.EPI_ACCUM, workspace=ws, colsum_a=cs)
    med, mn = timeit(run)
    o.linear_timing(True)
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    rows = o.linear_timing_read(); o.linear_timing(False)
    fl = 2.0 * M * N * K
    kern = "; ".join(f"{r['kernel']} {r['seconds'] / r['launches'] * 1e6:6.1f} us = {r['flops'] / r['seconds'] / 1e12:5.0f} TF" for r in rows)
    print(f"dW {M}x{N} K={K}: call {med:6.1f} us (min {mn:6.1f}) = {fl / med / 1e6:5.0f} TF | {kern}", flush=True)

if hasattr(L.lib, "gv_dw8_dbg_read"):     # library built with -DGV_DW8_STAMPS (tools/dw8_lab.sh)
    import ctypes, numpy as np
    buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
    L.lib.gv_dw8_dbg_read.argtypes = [ctypes.c_void_p]
    print("rc", L.lib.gv_dw8_dbg_read(buf.ctypes.data))
    b = buf.reshape(256, 8, 8).astype(np.float64)
    nt = b[:, :, 6]
    ok = nt[:, 0] > 0
    for grp, sl in (("wm=0", slice(0, 4)), ("wm=1", slice(4, 8))):
        x = b[ok][:, sl, :]
        ph = 3 * x[:, :, 6:7]
        per = (x[:, :, :5] / ph).mean(axis=(0, 1))
        print(grp, "cycles per phase: reads %.0f  issue+vmcnt %.0f  barrier1 %.0f  mfma %.0f  barrier2 %.0f | total/phase %.0f  nt %.0f" % (*per, (x[:, :, 5] / ph[:, :, 0]).mean(), x[:, :, 6].mean()))
