"""Run by tests/test_f16_gpu.py in a process of its own with GIPVIT_ACT_FORMAT=f16 (one process computes in one 16-bit format):
the float16 build of the library (libgipvit_hip_f16.so, --amp --amp-dtype float16) against the CPU oracle, and the device-side
GradScaler.  Prints one line per check and 'F16 OK' at the end; any failure is an exception (non-zero exit).

Parity: the oracle computes in f32; float16's 11-bit significand is held to a quarter of SURVEY 8d's bf16 column -- logits 5e-3 of
max |ref| (measured 1.0e-3 .. 1.3e-3), loss 1e-3 (1e-5 .. 6e-5), per-parameter gradients 1e-2 (1.4e-3 .. 2.5e-3), gradient norm 2e-3
(1e-4 .. 3e-4); the scaler's arithmetic (skip, back-off, growth, Adam's step count) is checked exactly."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert os.environ.get("GIPVIT_ACT_FORMAT") == "f16"

from gipvit import _lib, ops                                     # noqa: E402
from gipvit.engine import DinoEngine, SupervisedEngine           # noqa: E402
from oracle import step_oracle as so, vit_oracle as vo           # noqa: E402  (checker only)

assert _lib.lib.gv_act_format() == 1 and ops.bf16 is torch.float16
dev = torch.device("cuda:0")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check_grads(got, ref, tol=1e-2, skip=()):
    worst, gg, gr = [], 0.0, 0.0
    for k, r in ref.items():
        if r is None or k in skip:
            continue
        g = got[k].double().cpu()
        gg += float((g ** 2).sum()); gr += float((r.double() ** 2).sum())
        if float(r.abs().max()) >= 1e-12:
            worst.append((rel(g, r), k))
    worst.sort(reverse=True)
    assert worst[0][0] <= tol, worst[:6]
    gn = abs(math.sqrt(gg) - math.sqrt(gr)) / math.sqrt(gr)
    assert gn <= 2e-3, gn
    return worst[0], gn


def dino_parity(arch, n_local, B, K):
    orc = so.DinoOracle(arch=arch, img_size=224, out_dim=K, seed=0, n_local=n_local)
    eng = DinoEngine(arch=arch, img_size=224, out_dim=K, batch=B, n_local=n_local, device=dev)
    assert eng.scaler is not None and eng.hb_s.dlogits.dtype is torch.float16
    eng.load_state(orc.p, orc.hp)
    tiles = vo.synth_tiles(B, 256, seed=99)
    loss_r, grads_r, s_out, t_out, bsum = orc.forward_backward(tiles)
    eng.set_hyper()
    eng.forward_backward(tiles.to(dev))
    torch.cuda.synchronize()
    S = float(eng.scaler.state[0])
    errs = []
    for got, ref in ((eng.hb_t.logits, t_out), (eng.hb_s.logits, s_out)):
        e = float((got.float().cpu() - ref).abs().max()) / float(ref.abs().max())
        assert e <= 5e-3, e
        errs.append(e)
    dl = abs(float(eng.loss) - float(loss_r))
    assert dl <= 1e-3, (float(eng.loss), float(loss_r))
    worst, gn = check_grads(eng.grads(), grads_r, skip=("head.last_layer.weight_g",))      # grads(): the scale divided out
    print(f"dino {arch} L{n_local} B{B} K{K}: scale {S:g}  logits rel {max(errs):.2e}  |dloss| {dl:.2e}  worst grad {worst[0]:.2e} ({worst[1]})  "
          f"grad-norm rel {gn:.2e}", flush=True)


def supervised_parity_and_steps():
    orc = so.SupervisedOracle(arch="vit_tiny", img_size=64, num_classes=2, seed=0, lr=1e-3, wd=0.05)
    eng = SupervisedEngine(arch="vit_tiny", img_size=64, num_classes=2, batch=8, lr=1e-3, weight_decay=0.05, device=dev)
    eng.load_state(orc.p)
    tiles = vo.synth_tiles(8, 64, seed=1234)
    tgt = torch.randint(0, 2, (8, 1), generator=torch.Generator().manual_seed(5))
    loss_r, grads_r, logits_r = orc.forward_backward(tiles, tgt)
    eng.forward_backward(tiles.to(dev), tgt.to(dev))
    torch.cuda.synchronize()
    assert float(eng.scaler.state[0]) == 65536.0 and float(eng.arena.g.abs().max()) > 1.0      # the arena holds S x gradient
    assert float((eng.logits.cpu() - logits_r).abs().max()) <= 5e-3 * max(float(logits_r.abs().max()), 1.0)
    assert abs(float(eng.loss) - float(loss_r)) <= 1e-3
    worst, gn = check_grads(eng.grads(), grads_r)
    # three optimizer steps against the oracle's AdamW (the f32 reference trajectory): the unscale, the clip-free update and the
    # scaler's bookkeeping leave the parameters where the oracle's are
    dl = []
    for i in range(3):
        t = vo.synth_tiles(8, 64, seed=50 + i)
        lr_ = orc.step(t, tgt)["loss"]
        le = eng.step(t.to(dev), tgt.to(dev))
        torch.cuda.synchronize()
        dl.append(abs(float(le) - float(lr_)))
    assert max(dl) <= 1e-3, dl
    st = eng.scaler.state.tolist()
    assert st == [65536.0, 3.0, 0.0, 3.0], st
    sd = eng.state_dict()
    w = max(rel(sd[k], orc.p[k]) for k in ("blocks.0.attn.qkv.weight", "blocks.11.mlp.fc2.weight", "head.weight"))
    assert w < 2e-3, w
    print(f"supervised vit_tiny: worst grad {worst[0]:.2e} ({worst[1]})  grad-norm rel {gn:.2e}  3-step |dloss| {max(dl):.2e}  weights rel {w:.2e}", flush=True)


def scaler_mechanics():
    """GradScaler semantics on the device: an overflowing step is skipped (p, m, v untouched), halves the scale and does not count as
    an optimizer step; growth after growth_interval finite steps; a run whose first step overflowed continues exactly like a run that
    started at the backed-off scale."""
    mk = lambda: SupervisedEngine(arch="vit_tiny", img_size=64, num_classes=2, batch=8, lr=1e-3, weight_decay=0.05, clip_grad=1.0, device=dev)
    p0 = vo.init_vit("vit_tiny", 64, 2, seed=0)
    tiles = [vo.synth_tiles(8, 64, seed=70 + i).to(dev) for i in range(4)]
    tgt = torch.randint(0, 2, (8, 1), generator=torch.Generator().manual_seed(5)).to(dev)
    a, b = mk(), mk()
    a.load_state(p0); b.load_state(p0)
    a.scaler.growth_interval = b.scaler.growth_interval = 2
    # a: a scale no float16 gradient survives -> step 0 is skipped
    a.scaler.state[0] = 2.0 ** 40
    pa = a.arena.p.clone()
    a.step(tiles[0], tgt)
    torch.cuda.synchronize()
    assert torch.equal(a.arena.p, pa) and float(a.arena.m.abs().max()) == 0.0 and float(a.arena.v.abs().max()) == 0.0
    assert a.scaler.state.tolist() == [2.0 ** 39, 0.0, 1.0, 0.0], a.scaler.state.tolist()
    assert not math.isfinite(float(a.gnorm_sq))
    # from here a (scale reset to a working value) and b (never overflowed) must walk together: Adam's bias corrections count applied
    # steps, not calls
    a.scaler.state[0] = 1024.0
    b.scaler.state[0] = 1024.0
    for i in (1, 2, 3):
        la, lb = a.step(tiles[i], tgt), b.step(tiles[i], tgt)
        torch.cuda.synchronize()
        assert float(la) == float(lb)
    assert rel(a.arena.p, b.arena.p) < 1e-6 and rel(a.arena.m, b.arena.m) < 1e-6
    # 3 finite steps at growth_interval 2: one doubling after step 2, tracker back at 1 after step 3
    assert a.scaler.state.tolist() == [2048.0, 1.0, 1.0, 3.0] and b.scaler.state.tolist() == [2048.0, 1.0, 0.0, 3.0], (a.scaler.state.tolist(), b.scaler.state.tolist())
    # a power-of-two scale moves exponents only, so the scale is divided out exactly; what differs between two scales is which small
    # gradient values fall into float16's subnormal range (the reason the scale exists) -- two Adam steps stay within 2e-3
    c = mk(); c.load_state(p0); c.scaler.state[0] = 65536.0; c.scaler.growth_interval = 1000
    d = mk(); d.load_state(p0); d.scaler.state[0] = 1024.0; d.scaler.growth_interval = 1000
    for i in (1, 2):
        c.step(tiles[i], tgt); d.step(tiles[i], tgt)
    torch.cuda.synchronize()
    r = rel(c.arena.p, d.arena.p)
    assert 0 < r < 2e-3, r
    # state_dict round trip (timm's 'amp_scaler' checkpoint entry)
    sd = a.scaler.state_dict()
    assert sd["scale"] == 2048.0 and sd["_growth_tracker"] == 1 and sd["growth_interval"] == 2
    e = mk(); e.scaler.load_state_dict(sd)
    assert e.scaler.state.tolist() == a.scaler.state.tolist() and e.scaler.growth_interval == 2
    print(f"scaler: skip / back-off / growth / applied-step count exact; 65536 vs 1024 scale, 2 steps: rel {r:.1e}", flush=True)


def dino_overflow_recovers():
    """DINO at B = 1 per GPU-ish batch sizes overflows float16 at the initial scale (|dlogit| ~ S / (pairs B temp)); the scaler backs
    off until the step goes through, the teacher EMA and the centre keep running meanwhile."""
    K, B = 1024, 2
    eng = DinoEngine(arch="vit_tiny", img_size=224, out_dim=K, batch=B, lr=1e-4, clip_grad=3.0, device=dev)
    eng.load_state(vo.init_vit("vit_tiny", 224, 0, seed=0), vo.init_dino_head(192, K, seed=1))
    eng.scaler.state[0] = 2.0 ** 24           # dlogits ~ 2^24 / (18 * 2 * 0.1) >> 65504
    tiles = vo.synth_tiles(B, 256, seed=5).to(dev)
    p0 = eng.arena.p.clone()
    losses = []
    for _ in range(12):
        losses.append(float(eng.step(tiles)))
    torch.cuda.synchronize()
    st = eng.scaler.state.tolist()
    assert all(math.isfinite(l) for l in losses) and st[2] >= 1 and st[3] >= 1 and st[2] + st[3] == 12, (losses, st)
    assert not torch.equal(eng.arena.p, p0) and bool(torch.isfinite(eng.arena.p).all()) and bool(torch.isfinite(eng.arena.t).all())
    print(f"dino overflow: {int(st[2])} skipped, {int(st[3])} applied, scale {st[0]:g}, loss {losses[0]:.4f} -> {losses[-1]:.4f}", flush=True)


if __name__ == "__main__":
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    supervised_parity_and_steps()
    scaler_mechanics()
    dino_parity("vit_tiny", 8, 2, 4096)
    dino_parity("vit_small", 8, 4, 2048)          # the fused full-row kernels, varlen attention, grouped dW
    dino_parity("vit_base", 8, 2, 2048)
    dino_overflow_recovers()
    print("F16 OK")
