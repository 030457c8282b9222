"""GPU: the data-parallel path end to end with two ranks (gloo transport, both on the one GPU of
the test box): per-rank tile shards, ranged gradient all-reduce + centre all-reduce through
RcclReducer, 1/world folded into the optimizer.  Two ranks x B=2 must reproduce one process
with B=4 on the concatenated tiles (same mean loss, same update) up to bf16 / summation-order noise."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = 1024


def _make(B, reducer=None):
    from gipvit.engine import DinoEngine
    from gipvit.models import init_vit_state, init_dino_head_state
    eng = DinoEngine(arch="vit_tiny", img_size=224, out_dim=K, batch=B, lr=1e-4, weight_decay=0.04, clip_grad=3.0, device="cuda:0", reducer=reducer)
    eng.load_state(init_vit_state("vit_tiny", 224, 0, seed=0), init_dino_head_state(192, K, seed=1))
    return eng


def _tiles():
    sys.path.insert(0, ROOT)
    from bench import synth_tiles
    return synth_tiles(4, 256, 99, "cpu")


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gipvit.dist import RcclReducer, shard_range
    eng = _make(2, RcclReducer())
    lo, hi = shard_range(4, rank, world)
    tiles = _tiles()[lo:hi].to("cuda:0")
    # gradient level first (before any optimizer step: Adam is scale-invariant and would hide a wrong 1/world):
    # the reduced arena x the grad_scale the optimizer kernel will apply = the mean gradient over all four tiles
    eng.set_hyper()
    eng.forward_backward(tiles)
    torch.cuda.synchronize()
    scale = float(eng.hyper[5])                      # GV_HYP_GRAD_SCALE, what gv_adamw_ema multiplies the arena by
    names = ("backbone.blocks.0.attn.qkv.weight", "backbone.blocks.11.mlp.fc2.weight", "backbone.pos_embed", "backbone.norm.weight",
             "head.mlp.0.weight", "head.last_layer.weight_v")
    gr = eng.grads()
    gsel = {k: (gr[k] * scale).cpu().numpy() for k in names}
    gnorm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in gr.values()))) * scale
    eng.t = 0
    eng.arena.g.zero_()
    losses = [float(eng.step(tiles)) for _ in range(2)]
    torch.cuda.synchronize()
    sd = eng.backbone_state_dict()
    # numpy payloads: pickled by value (torch tensors would travel as fds the exiting child closes)
    q.put((rank, losses, sd["blocks.0.attn.qkv.weight"].cpu().numpy(), sd["pos_embed"].cpu().numpy(),
           eng.head_state_dict()["last_layer.weight_v"][:64].cpu().numpy(), eng.center.cpu().numpy(), gsel, gnorm, scale))
    dist.destroy_process_group()


def test_two_ranks_match_single_process(dev):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:              # a port the kernel reports free (a pid-derived one collided now and then)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=300) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    # ---- gradient level: reduced gradient x grad_scale == single-process mean gradient == the oracle's
    gsel, gnorm, scale = res[0][6], res[0][7], res[0][8]
    assert scale == 0.5, scale                       # 1 / world
    for k in gsel:                                   # both replicas hold the same reduced arena
        assert (gsel[k] == res[1][6][k]).all(), k
    one = _make(4)
    one.set_hyper(); one.forward_backward(_tiles().to(dev)); torch.cuda.synchronize()
    g1 = one.grads()
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    for k, g in gsel.items():
        assert rel(torch.from_numpy(g), g1[k].cpu()) < 2e-2, (k, rel(torch.from_numpy(g), g1[k].cpu()))      # bf16 noise only; a wrong 1/world is a factor 2
    gn1 = float(torch.sqrt(sum((g.double() ** 2).sum() for g in g1.values())))
    assert abs(gnorm - gn1) / gn1 < 1e-2, (gnorm, gn1)
    from oracle import step_oracle as so
    orc = so.DinoOracle(arch="vit_tiny", img_size=224, out_dim=K, seed=0)
    orc.p.update(one.backbone_state_dict()); orc.hp.update({k: v.cpu() for k, v in one.head_state_dict().items()})
    orc.p = {k: v.cpu() for k, v in orc.p.items()}
    _, g_or, _, _, _ = orc.forward_backward(_tiles())
    for k, g in gsel.items():
        assert rel(torch.from_numpy(g), g_or[k]) < 5e-2, (k, rel(torch.from_numpy(g), g_or[k]))
    gno = so.grad_norm({k: v for k, v in g_or.items() if k != "head.last_layer.weight_g"})
    assert abs(gnorm - gno) / gno < 1e-2, (gnorm, gno)
    # replicas stay identical (same reduced gradients, same update)
    res = [(r[0], r[1]) + tuple(torch.from_numpy(x) for x in r[2:6]) for r in res]
    for a, b in zip(res[0][2:], res[1][2:]):
        assert torch.equal(a, b)
    # and equal to one process on all four tiles
    ref = _make(4)
    tiles = _tiles().to(dev)
    ref_losses = [float(ref.step(tiles)) for _ in range(2)]
    torch.cuda.synchronize()
    mean_losses = [(res[0][1][i] + res[1][1][i]) / 2 for i in range(2)]
    assert abs(mean_losses[0] - ref_losses[0]) < 1e-4, (mean_losses, ref_losses)      # same weights: only the batch split differs
    assert abs(mean_losses[1] - ref_losses[1]) < 5e-3, (mean_losses, ref_losses)      # after one (Adam) update
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    sd = ref.backbone_state_dict()
    assert rel(res[0][2], sd["blocks.0.attn.qkv.weight"].cpu()) < 5e-3
    assert rel(res[0][5], ref.center.cpu()) < 1e-3
