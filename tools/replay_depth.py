"""How many replays of one hipGraph exec may be in flight before results go wrong? (debug aid)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gipvit.engine import DinoEngine
from gipvit.models import init_vit_state, init_dino_head_state
from bench import synth_tiles
dev = torch.device("cuda:0")
B, STEPS = 64, 45
tiles = synth_tiles(B, 256, 1234, dev)
bb, hd = init_vit_state("vit_small", 224, 0, seed=0), init_dino_head_state(384, 65536, seed=1)
for depth in (1, 2, 3, 4, 8, 16, 0):
    eng = DinoEngine(arch="vit_small", img_size=224, out_dim=65536, batch=B, lr=1.25e-4, clip_grad=3.0, device=dev)
    eng.load_state(bb, hd)
    eng.capture(tiles)
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    for i in range(STEPS):
        if depth and len(evs) >= depth:
            evs[-depth].synchronize()
        eng.set_hyper()
        eng.graph.replay()
        e = torch.cuda.Event(); e.record(); evs.append(e)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"in-flight {depth or 'unbounded'}: final loss {float(eng.loss):.5f}  {B * STEPS / dt:.1f} tiles/s", flush=True)
