"""Print the per-step loss / grad-norm of the bench configuration (debug aid)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gipvit.engine import DinoEngine
from gipvit.models import init_vit_state, init_dino_head_state
from gipvit import ops
from bench import synth_tiles
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = DinoEngine(arch="vit_small", img_size=224, out_dim=65536, batch=B, device=dev)
eng.load_state(init_vit_state("vit_small", 224, 0, seed=0), init_dino_head_state(384, 65536, seed=1))
tiles = synth_tiles(B, 256, 1234, dev)
for i in range(16):
    eng.set_hyper(); eng.forward_backward(tiles)
    ops.sumsq(eng.arena.g, eng.red_ws, eng.gnorm_sq)
    torch.cuda.synchronize()
    g = eng.arena.g
    print(i, "loss", float(eng.loss), "gnorm", float(eng.gnorm_sq) ** 0.5, "nan grads", int(torch.isnan(g).sum()),
          "max|logit|", float(eng.hb_s.logits.abs().max()), "center max", float(eng.center.abs().max()), flush=True)
    eng.optimizer_step()
