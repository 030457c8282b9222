"""one replay-ordering configuration per process (debug aid): usage replay_cfg.py <cfg>"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gipvit.engine import DinoEngine
from gipvit.models import init_vit_state, init_dino_head_state
from bench import synth_tiles
cfg = sys.argv[1]
dev = torch.device("cuda:0")
B = 64
eng = DinoEngine(arch="vit_small", img_size=224, out_dim=65536, batch=B, lr=1.25e-4, clip_grad=3.0, device=dev)
eng.load_state(init_vit_state("vit_small", 224, 0, seed=0), init_dino_head_state(384, 65536, seed=1))
tiles = synth_tiles(B, 256, 1234, dev)
eng.capture(tiles)
evs = []
def step():
    if cfg == "E1" and evs: evs[-1].synchronize()
    if cfg == "E2" and len(evs) >= 2: evs[-2].synchronize()
    if cfg == "F": torch.cuda.synchronize()          # user code doing a device sync between steps
    eng.step_graph(sync=cfg in ("F", "G"))
    if cfg in ("C", "E1", "E2"):
        e = torch.cuda.Event(); e.record(); evs.append(e)
    if cfg == "D": torch.cuda.current_stream().synchronize()
for i in range(5): step()
if cfg != "A": torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(40): step()
torch.cuda.synchronize()
print(f"cfg {cfg}: final loss {float(eng.loss):.5f}   {B * 40 / (time.perf_counter() - t0):.0f} tiles/s", flush=True)
