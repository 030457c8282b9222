"""burst -> idle -> burst patterns of graph replays (debug aid)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gipvit.engine import DinoEngine
from gipvit.models import init_vit_state, init_dino_head_state
from bench import synth_tiles
dev = torch.device("cuda:0")
B = 64
tiles = synth_tiles(B, 256, 1234, dev)
bb, hd = init_vit_state("vit_small", 224, 0, seed=0), init_dino_head_state(384, 65536, seed=1)
def run(name, mid, chain=False):
    eng = DinoEngine(arch="vit_small", img_size=224, out_dim=65536, batch=B, lr=1.25e-4, clip_grad=3.0, device=dev)
    eng.load_state(bb, hd)
    eng.capture(tiles)
    prev = [None]
    def step():
        if chain and prev[0] is not None:
            torch.cuda.current_stream().wait_event(prev[0])
        eng.step_graph()
        if chain:
            prev[0] = torch.cuda.Event(); prev[0].record()
    for i in range(5): step()
    mid(eng)
    for i in range(40): step()
    torch.cuda.synchronize()
    print(f"{name}: final loss {float(eng.loss):.5f}", flush=True)
run("device sync", lambda e: torch.cuda.synchronize())
run("device sync + event chain", lambda e: torch.cuda.synchronize(), chain=True)
run("device sync + event chain (again)", lambda e: torch.cuda.synchronize(), chain=True)
run("device sync, then stream sync", lambda e: (torch.cuda.synchronize(), torch.cuda.current_stream().synchronize()))
run("device sync, then tiny kernel + stream sync", lambda e: (torch.cuda.synchronize(), e.loss.add_(0.0), torch.cuda.current_stream().synchronize()))
