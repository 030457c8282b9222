"""Lab: the one kernel pair that tests DESIGN.md section 5a's hypothesis directly.  The fused Linear + LayerNorm forward (K = 1536:
an MFMA-bound k-loop, then an epilogue at the chip's aggregate HBM rate) is run
  (a) as ONE stream of full launches (M = 44 160: 251 workgroups, every CU in the same phase), and
  (b) as TWO streams of half launches (M = 22 080 each: 126 workgroups, sized for 128 CUs by the lab library), the second stream
      started half a kernel late, so that one half of the chip is in its epilogue while the other is in its k-loop.
Same rows per CU and per unit of time in both; if (b) finishes its 2 x n launches sooner than (a) its n, the shifted epilogues ran
at the higher per-CU rate of profiles/r03_cu_stream_rate.txt.
    GIPVIT_LIB=tools/lab_build/lib_cubudget.so GIPVIT_CU_BUDGET=128 python tools/desync_probe.py     (tools/lab.sh cubudget panel -DGV_LAB_CU_BUDGET)
(the full launches are measured by a second process without GIPVIT_CU_BUDGET: python tools/desync_probe.py full)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit import ops as o

dev = torch.device("cuda:0")
bf16 = torch.bfloat16
N, REPS = 384, int(os.environ.get("PROBE_REPS", "200"))
mode = sys.argv[1] if len(sys.argv) > 1 else "halves"


def make(M, K, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g).to(dev).to(bf16)
    W = (0.05 * torch.randn(N, K, generator=g)).to(dev).to(bf16)
    bias = torch.randn(N, generator=g).to(dev); gamma = torch.ones(N, device=dev); beta = torch.zeros(N, device=dev)
    resid = torch.randn(M, N, generator=g).to(dev)
    out = torch.empty(M, N, device=dev); y = torch.empty(M, N, dtype=bf16, device=dev)
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    return lambda: o.linear_ln_fwd(A, W, out, M, K, bias=bias, resid=resid, gamma=gamma, beta=beta, y=y, mean=mean, rstd=rstd)


def wall(fns, streams, delay_us=0.0):
    """Device time of REPS launches per stream.  Every launch is enqueued while the streams are parked behind a 40-ms spin, so the
    host's enqueue rate (15 - 30 us per launch from Python) is not in the measurement."""
    for f, s in zip(fns, streams):
        with torch.cuda.stream(s):
            for _ in range(5): f()
    torch.cuda.synchronize()
    gate, start = torch.cuda.Event(), torch.cuda.Event(enable_timing=True)
    ends = [torch.cuda.Event(enable_timing=True) for _ in fns]
    with torch.cuda.stream(streams[0]):
        torch.cuda._sleep(int(40e3 * 100))                       # (the spin counts a 100-MHz clock)
        gate.record(); start.record()
    for i, (f, s) in enumerate(zip(fns, streams)):
        with torch.cuda.stream(s):
            if i > 0:
                s.wait_event(gate)
                if delay_us > 0:
                    torch.cuda._sleep(int(delay_us * 100))       # the second chain starts late
            for _ in range(REPS): f()
            ends[i].record()
    torch.cuda.synchronize()
    return max(start.elapsed_time(e) for e in ends) * 1e3 / REPS


for K in (1536, 384):
    if mode == "full":
        t = wall([make(44160, K, 1)], [torch.cuda.Stream(dev)])
        print(f"K={K}: one stream of full launches (251 workgroups): {t:6.1f} us per 44160 rows", flush=True)
    else:
        fa, fb = make(22080, K, 1), make(22080, K, 2)
        sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        t1 = wall([fa], [sa])
        print(f"K={K}: one stream of half launches alone (126 workgroups): {t1:6.1f} us per 22080 rows", flush=True)
        for d in (0.0, 0.25 * t1, 0.5 * t1):
            t = wall([fa, fb], [sa, sb], d)
            print(f"K={K}: two streams of half launches, second {d:5.1f} us late: {t:6.1f} us per 44160 rows", flush=True)
