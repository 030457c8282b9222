"""Which inter-replay ordering keeps back-to-back hipGraph replays correct? (debug aid)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gipvit.engine import DinoEngine
from gipvit.models import init_vit_state, init_dino_head_state
from bench import synth_tiles
dev = torch.device("cuda:0")
B = 64
tiles = synth_tiles(B, 256, 1234, dev)
bb, hd = init_vit_state("vit_small", 224, 0, seed=0), init_dino_head_state(384, 65536, seed=1)
for mode in (0, 1, 2, 3, 4, 0):
    eng = DinoEngine(arch="vit_small", img_size=224, out_dim=65536, batch=B, lr=1.25e-4, clip_grad=3.0, device=dev)
    eng.load_state(bb, hd)
    eng.capture(tiles)
    torch.cuda.synchronize()
    prev = None
    for i in range(13):
        if mode == 4:
            if i == 0: eng.set_hyper()
        else:
            eng.set_hyper()
        if mode == 2 and prev is not None:
            torch.cuda.current_stream().wait_event(prev)
        if mode == 3:
            eng.loss.add_(0.0)
        eng.graph.replay()
        if mode == 1:
            torch.cuda.current_stream().synchronize()
        if mode == 2:
            prev = torch.cuda.Event(); prev.record()
    torch.cuda.synchronize()
    print(f"mode {mode}: final loss {float(eng.loss):.5f}  nan params {int(torch.isnan(eng.arena.p).sum())}", flush=True)
