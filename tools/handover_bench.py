"""Tile hand-over cost: the DINO step on tiles resident in HBM vs tiles read from the reference's raw tile files
(<slide>/tile_<i>.data) through reader threads -> pinned staging -> copy stream (data.DevicePrefetcher), with and without the device
augmentation.  python tools/handover_bench.py [steps]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
from gipvit import data as D
from gipvit.engine import DinoEngine
from gipvit.models import init_vit_state, init_dino_head_state
from gipvit.augment import TileAugmenter

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
B = 64
root = tempfile.mkdtemp(prefix="tiles_")
rng = np.random.default_rng(0)
for s in range(8):
    os.makedirs(os.path.join(root, f"slide{s}"))
    for i in range(64):
        D.write_tile_file(os.path.join(root, f"slide{s}", f"tile_{i}.data"), rng.integers(0, 256, (256, 256, 3), dtype=np.uint8))
eng = DinoEngine(arch="vit_small", img_size=224, out_dim=65536, batch=B, lr=1e-4, clip_grad=3.0, device=dev)
eng.load_state(init_vit_state("vit_small", 224, 0, seed=0), init_dino_head_state(384, 65536, seed=1))
torch.cuda.set_stream(torch.cuda.Stream(dev, priority=-1))
resident = torch.from_numpy(rng.integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).to(dev)

def run(it, n, aug=None):
    torch.cuda.synchronize(); t0 = time.perf_counter(); k = 0
    for mb in it:
        tiles, fill = mb["Data"], None
        if aug is not None and "AugParams" in mb:
            tiles, fill = aug.run(tiles, mb["AugParams"]), mb["Fill"]
        elif aug is not None:
            tiles, fill = aug.apply(tiles)
        eng.step(tiles, fill=fill); k += 1
        if k == n: break
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k

class Resident:
    def __iter__(self):
        while True: yield {"Data": resident}

run(Resident(), 5)
t_res = run(Resident(), steps)
src = D.TileFolder(root, B, None, seed=0, tile_size=256, n_tiles=8 * steps, workers=8)
pf = D.DevicePrefetcher(src, dev, (B, 256, 256, 3))
run(pf, 5)                                               # first use: pinned allocations, reader pool
t_files = run(pf, steps)
aug = TileAugmenter("pcbnfrs", 256, 0.1, seed=0)
run(Resident(), 3, aug)                                  # first use: code-object load, normal table
pfa = D.DevicePrefetcher(src, dev, (B, 256, 256, 3), aug)
run(pfa, 5, aug)
t_files_aug = run(pfa, steps, aug)
t_res_aug = run(Resident(), steps, aug)
print(f"DINO ViT-S B=64 ms/step: resident {t_res:.2f} | tile files via pinned prefetcher {t_files:.2f} ({100 * (t_files / t_res - 1):+.1f} %) | "
      f"resident + device augmentation 'pcbnfrs' {t_res_aug:.2f} | files + augmentation {t_files_aug:.2f} ({100 * (t_files_aug / t_res - 1):+.1f} %)")
