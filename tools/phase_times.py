"""Wall time of the step's phases (HIP events on the main stream): teacher + student forward, heads + loss, head backward, encoder
backward (with the side stream's dW products overlapped), optimizer.  python tools/phase_times.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from gipvit.engine import DinoEngine
from gipvit.models import init_vit_state, init_dino_head_state
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
eng = DinoEngine(arch="vit_small", img_size=224, out_dim=65536, batch=64, lr=1e-4, clip_grad=3.0, device=dev)
eng.load_state(init_vit_state("vit_small", 224, 0, seed=0), init_dino_head_state(384, 65536, seed=1))
tiles = bench.synth_tiles(64, 256, 1234, dev)
torch.cuda.set_stream(torch.cuda.Stream(dev, priority=-1))
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
# wrap the engine's pieces
vf, hf, hb, vb = eng.vit.forward, eng.head.forward, eng.head.backward, eng.vit.backward
import gipvit.ops as ops
dl = ops.dino_loss
def vit_forward(W, *a, **k):
    r = vf(W, *a, **k)
    if W is eng.sW: mark("student_fwd")
    return r
def head_backward(*a, **k):
    r = hb(*a, **k); mark("head_bwd"); return r
def vit_backward(*a, **k):
    r = vb(*a, **k); mark("vit_bwd"); return r
def dino_loss(*a, **k):
    r = dl(*a, **k); mark("heads+loss"); return r
eng.vit.forward, eng.head.backward, eng.vit.backward, ops.dino_loss = vit_forward, head_backward, vit_backward, dino_loss
for _ in range(5): eng.step(tiles)
torch.cuda.synchronize()
acc = {}
for _ in range(steps):
    marks.clear(); mark("start")
    eng.step(tiles); mark("optimizer")
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
tot = sum(acc.values()) / steps
print("  ".join(f"{k} {v / steps:.2f}" for k, v in acc.items()), f"| total {tot:.2f} ms/step")
