"""Lab: do two half-batch DINO steps on two streams overlap each other's HBM-bound epilogues and MFMA-bound k-loops?
Two independent engines of B tiles each, stepped from one host thread on two high-priority streams, against one engine of 2B.
    python tools/concurrent_micro.py [B]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit.engine import DinoEngine
from gipvit.models import init_vit_state, init_dino_head_state
import bench

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def make(b):
    e = DinoEngine(arch="vit_small", img_size=224, out_dim=65536, batch=b, lr=1e-4, clip_grad=3.0, device=dev)
    e.load_state(init_vit_state("vit_small", 224, 0, seed=0), init_dino_head_state(384, 65536, seed=1))
    return e, bench.synth_tiles(b, 256, 1234, dev)


def run(engs, steps=20, warm=5):
    streams = [torch.cuda.Stream(dev, priority=-1) for _ in engs]
    def one():
        for (e, t), s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.step(t)
    for _ in range(warm): one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps

if not os.environ.get("GIPVIT_CU_BUDGET"):
    big = [make(N * B)]
    dt = run(big)
    print(f"one engine  B={N * B}: {dt * 1e3:.2f} ms/step  {N * B / dt:.0f} tiles/s", flush=True)
    del big; torch.cuda.empty_cache()
two = [make(B) for _ in range(N)]
dt = run(two)
print(f"{N} engines   B={B} each, concurrent: {dt * 1e3:.2f} ms/step  {N * B / dt:.0f} tiles/s", flush=True)
dt1 = run(two[:1])
print(f"one engine  B={B} alone: {dt1 * 1e3:.2f} ms/step  {B / dt1:.0f} tiles/s", flush=True)
