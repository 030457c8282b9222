"""Time gv_crop_resize on the bench's shapes (64 tiles of 256 px -> 2x224 + 8x96 crops) and print GB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_tiles
from gipvit import ops
from gipvit.multicrop import MultiCropSampler
dev = torch.device("cuda", 0)
B = 64
tiles = synth_tiles(B, 256, 1234, dev)
g, l = MultiCropSampler(B, 256, seed=0).sample(dev)
og = torch.empty(2 * B, 224, 224, 3, dtype=torch.uint8, device=dev); ol = torch.empty(8 * B, 96, 96, 3, dtype=torch.uint8, device=dev)
for name, boxes, size, out in (("global 2x224", g, 224, og), ("local 8x96", l, 96, ol)):
    for _ in range(3): ops.crop_resize(tiles, boxes, size, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.crop_resize(tiles, boxes, size, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    src = float((boxes[:, 3].double() * boxes[:, 4].double()).sum()) * 3      # bytes of the boxes (read once, algorithmic)
    dst = out.numel()
    print(f"{name}: {us:.1f} us, {src/1e6:.1f} MB box bytes in + {dst/1e6:.1f} MB out -> {(src+dst)/us/1e3:.0f} GB/s")
