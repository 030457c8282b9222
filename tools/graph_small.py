import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gipvit.engine import DinoEngine
from gipvit import ops
from oracle import step_oracle as so, vit_oracle as vo
dev = torch.device("cuda:0")
K, B = 2048, 2
orc = so.DinoOracle(arch="vit_tiny", img_size=224, out_dim=K, seed=0)
tiles = vo.synth_tiles(B, 256, seed=7).to(dev)
for mode in sys.argv[1:]:
    eng = DinoEngine(arch="vit_tiny", img_size=224, out_dim=K, batch=B, lr=2e-5, device=dev)
    eng.load_state(orc.p, orc.hp)
    if mode.startswith("graph"):
        eng.capture(tiles)
        if mode == "graph_reset":
            eng.load_state(orc.p, orc.hp); eng.arena.m.zero_(); eng.arena.v.zero_(); eng.center.zero_(); eng.t = 0
    out = []
    for i in range(6):
        l = eng.step_graph(tiles) if mode.startswith("graph") else eng.step(tiles)
        torch.cuda.synchronize()
        a = eng.arena
        out.append(f"{float(l):.4f}[gn {float(a.g.norm()):.3f} |p| {float(a.p.norm()):.2f} c {float(eng.center.abs().max()):.3f}]")
    print(mode, " ".join(out), flush=True)
