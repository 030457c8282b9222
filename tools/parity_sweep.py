"""Print GPU-vs-oracle errors of one DINO forward/backward over architectures / crop configs (debug aid for the parity gates)."""
import sys, math, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gipvit.engine import DinoEngine
from oracle import step_oracle as so, vit_oracle as vo
def rel(a,b): a=a.double().cpu(); b=b.double().cpu(); return float((a-b).norm()/b.norm())
for arch,n_local,B,K,seed in [("vit_tiny",8,2,2048,99),("vit_small",0,2,2048,99),("vit_small",0,2,2048,7),("vit_small",8,2,2048,99),("vit_small",8,2,4096,99),("vit_base",8,1,2048,99)]:
    orc = so.DinoOracle(arch=arch, img_size=224, out_dim=K, seed=0, n_local=n_local)
    eng = DinoEngine(arch=arch, img_size=224, out_dim=K, batch=B, n_local=n_local, device="cuda:0")
    eng.load_state(orc.p, orc.hp)
    tiles = vo.synth_tiles(B, 256, seed=seed)
    loss_r, grads_r, s_out, t_out, bsum = orc.forward_backward(tiles)
    eng.set_hyper(); eng.forward_backward(tiles.cuda()); torch.cuda.synchronize()
    g = eng.grads()
    worst = max((rel(g[k], r), k) for k, r in grads_r.items() if r is not None and float(r.abs().max()) > 1e-12 and k != "head.last_layer.weight_g")
    print(arch, n_local, B, K, seed, "loss", float(eng.loss), float(loss_r), "d", abs(float(eng.loss)-float(loss_r)),
          "t_err", float((eng.hb_t.logits.cpu()-t_out).abs().max()), "/", float(t_out.abs().max()),
          "s_err", float((eng.hb_s.logits.cpu()-s_out).abs().max()), "/", float(s_out.abs().max()), "worst grad", worst, flush=True)
    del eng
