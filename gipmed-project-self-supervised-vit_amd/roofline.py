"""Live roofline measurement of the dominant kernel for bench.py's JSON line.

The step is MFMA-bound as a whole (98.7 % of FLOPs are dense contractions); the
dominant kernel is the NT GEMM with bf16 output (`gemm_kernel<false,false,bf16,false>`:
QKV and fc1 of every block and the two hidden DINOHead layers).  Each distinct shape of
that kernel in one step is launched back-to-back on the current stream between two HIP
events; achieved = algorithmic FLOPs of all its launches in a step / their summed time,
i.e. average FLOPs per launch / average launch duration -- the figure rocprofv3's
--kernel-trace --stats average for the same kernel name must agree with."""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops

PEAK_BF16_TFLOPS = 2500.0     # MI355X_MICROARCH.md, dense bf16 MFMA


def _time_linear(M, N, K, epilogue, dev, reps=20):
    A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    aux = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    bias = torch.zeros(N, device=dev)
    kw = dict(epilogue=epilogue, bias=bias, aux_out=aux if epilogue & L.EPI_SAVE_PRE else None)
    for _ in range(3):
        ops.linear(A, B, C, M, N, K, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.linear(A, B, C, M, N, K, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def dominant_kernel_roofline(eng, batch):
    dev = eng.dev
    D, depth = eng.D, eng.vit.depth
    groups = [(eng.g_teach.T, 1), (eng.g_stu.T, 1)]
    shapes = {}
    for T, _ in groups:
        for (N, epi) in ((3 * D, L.EPI_BIAS), (4 * D, L.EPI_BIAS | L.EPI_GELU | L.EPI_SAVE_PRE)):
            key = (T, N, D, epi)
            shapes[key] = shapes.get(key, 0) + depth
    for hb in (eng.hb_t, eng.hb_s):
        hid = eng.head.hidden
        for K in (D, hid):
            key = (hb.R, hid, K, L.EPI_BIAS | L.EPI_GELU | L.EPI_SAVE_PRE)
            shapes[key] = shapes.get(key, 0) + 1
    flops = secs = 0.0
    launches = 0
    detail = []
    for (M, N, K, epi), cnt in shapes.items():
        t = _time_linear(M, N, K, epi, dev)
        f = 2.0 * M * N * K
        flops += f * cnt; secs += t * cnt; launches += cnt
        detail.append({"M": M, "N": N, "K": K, "launches_per_step": cnt, "us": round(t * 1e6, 2), "tflops": round(f / t / 1e12, 1)})
    ach = flops / secs / 1e12
    return {"bound": "mfma", "kernel": "gemm_kernel<false,false,bf16,false> (NT, bf16 out: QKV / fc1 / DINOHead hidden)",
            "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
            "traffic": None, "launches_per_step": launches, "avg_launch_us": round(secs / launches * 1e6, 2),
            "avg_gflop_per_launch": round(flops / launches / 1e9, 3), "shapes": detail}
