"""Forward-only throughput of the encoder on 256-px tiles (slide-level feature extraction)."""
import os, sys, time, torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_tiles
from gipvit.engine import FeatureExtractor
from gipvit.models import init_vit_state
dev = torch.device("cuda", 0)
for arch, B in (("vit_small", 256), ("vit_base", 128)):
    fx = FeatureExtractor(arch=arch, img_size=256, batch=B, num_classes=2, device=dev)
    fx.load_state(init_vit_state(arch, 256, 2, seed=0))
    tiles = synth_tiles(B, 256, 7, dev)
    for _ in range(3): fx.forward(tiles)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fx.forward(tiles)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    gf = {"vit_small": 12.28, "vit_base": 46.8}[arch]          # forward GFLOP per 256-px tile (ViT-B scaled from 35.1 @224)
    print(f"{arch} 256px B={B}: {B / dt:.0f} tiles/s forward-only ({dt * 1e3:.2f} ms/batch, {B / dt * gf / 1e3:.0f} TFLOP/s)")
    del fx
