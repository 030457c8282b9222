"""Host-side enqueue time of one DINO step vs its GPU time (is the step launch-bound?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_tiles
from gipvit.engine import DinoEngine
from gipvit.models import init_vit_state, init_dino_head_state
dev = torch.device("cuda", 0)
eng = DinoEngine(arch="vit_small", img_size=224, out_dim=65536, batch=64, n_local=8, lr=1e-4, clip_grad=3.0, device=dev)
eng.load_state(init_vit_state("vit_small", 224, 0, seed=0), init_dino_head_state(eng.D, 65536, seed=1))
tiles = synth_tiles(64, 256, 1234, dev)
for _ in range(3): eng.step(tiles)
torch.cuda.synchronize()
for label in ("side on", "side off"):
    if label == "side off": eng.vit.side = None
    torch.cuda.synchronize()
    t0 = time.perf_counter(); enq = 0.0
    for _ in range(10):
        a = time.perf_counter(); eng.step(tiles); enq += time.perf_counter() - a
        torch.cuda.synchronize()       # so enqueue time is measured against an idle queue each step
    tot = time.perf_counter() - t0
    print(f"{label}: host enqueue {enq/10*1e3:.2f} ms/step, step (enqueue + drain) {tot/10*1e3:.2f} ms")
