#!/usr/bin/env python3
"""Every gv_linear call of one B = 64 DINO step with fewer than 4096 rows on either side of the product, timed one at a time
(side stream off, two events per call): shape, flags, microseconds.  The head, the patch embedding and the CLS-only tail."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GIPVIT_DW_STREAM", "0")
from gipvit import ops
from gipvit.engine import DinoEngine
dev = torch.device("cuda:0")
eng = DinoEngine(arch="vit_small", img_size=224, out_dim=65536, batch=64, device=dev)
eng.vit.side = None
eng.set_hyper()
tiles = torch.randint(0, 256, (64, 256, 256, 3), dtype=torch.uint8, device=dev)
rec, on = [], [False]
orig = ops.linear
def timed(A, B, C, M, N, K, **kw):
    if not on[0] or min(M, N) >= 4096 and K >= 4096:
        return orig(A, B, C, M, N, K, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(A, B, C, M, N, K, **kw); e1.record()
    rec.append(((M, N, K, int(kw.get("trans_a", False)), int(kw.get("trans_b", False)), kw.get("epilogue", 0), str(C.dtype)[6:]), e0, e1))
    return r
ops.linear = timed
import gipvit.engine as E
E.ops.linear = timed
for _ in range(3): eng.step(tiles)
torch.cuda.synchronize()
on[0] = True
N_STEPS = 4
for _ in range(N_STEPS): eng.step(tiles)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1 in rec:
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
tot = 0.0
print(f"{'M':>6} {'N':>6} {'K':>6} tA tB {'epi':>5} {'out':>9} {'n/step':>6} {'us':>7} {'us/step':>8}")
for (M, N, K, ta, tb, epi, dt), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if M >= 4096 and N >= 4096: continue
    print(f"{M:6d} {N:6d} {K:6d} {ta:2d} {tb:2d} {epi:5d} {dt:>9} {n / N_STEPS:6.1f} {t / n:7.1f} {t / N_STEPS:8.1f}")
    tot += t / N_STEPS
print(f"total {tot:.0f} us per step (event-to-event: includes the launch gap)")
