import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gipvit.engine import DinoEngine
from gipvit.models import init_vit_state, init_dino_head_state
from bench import synth_tiles
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sync = len(sys.argv) > 2
eng = DinoEngine(arch="vit_small", img_size=224, out_dim=65536, batch=B, device=dev)
eng.load_state(init_vit_state("vit_small", 224, 0, seed=0), init_dino_head_state(384, 65536, seed=1))
tiles = synth_tiles(B, 256, 1234, dev)
eng.capture(tiles)
torch.cuda.synchronize()
a = eng.arena
print("after capture: nan p", int(torch.isnan(a.p).sum()), "nan m", int(torch.isnan(a.m).sum()), "nan v", int(torch.isnan(a.v).sum()),
      "nan t", int(torch.isnan(a.t).sum()), "nan center", int(torch.isnan(eng.center).sum()), "hyper", eng.hyper.tolist())
for i in range(14):
    l = eng.step_graph()
    if sync:
        torch.cuda.synchronize()
        print(i, float(l), "nan p", int(torch.isnan(a.p).sum()), "hyper", [round(x, 5) for x in eng.hyper.tolist()], flush=True)
torch.cuda.synchronize()
print("final", float(eng.loss), "nan p", int(torch.isnan(a.p).sum()))
