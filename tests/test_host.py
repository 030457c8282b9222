"""CPU: the C-ABI library loads and exports every symbol include/gipvit.h declares (no
compute calls without a GPU), the ctypes structs have the header's layout, host-side logic
(arena layout, model registry, init, checkpoint I/O, reducer over gloo)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gipvit import _lib
    hdr = open(os.path.join(ROOT, "include", "gipvit.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(gv_\w+)\s*\(", hdr, re.M))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(_lib.lib, name), f"{name} declared in gipvit.h but not exported"
    bound = set(_lib.ENTRY_POINTS) | set(_lib.PLAIN_SYMBOLS)
    assert declared == bound, (declared - bound, bound - declared)
    assert _lib.lib.gv_version() == 9 and _lib.lib.gv_target() == b"gfx950" and _lib.lib.gv_act_format() == 0
    # the float16 build (-DGV_ACT_F16, --amp --amp-dtype float16) is the same ABI (looked at from a process of its own: one process
    # loads one of the two libraries)
    code = ("import ctypes, sys; l = ctypes.CDLL(sys.argv[1]); missing = [n for n in sys.argv[2:] if not hasattr(l, n)]; "
            "assert not missing, missing; assert l.gv_version() == 9 and l.gv_act_format() == 1")
    r = subprocess.run([sys.executable, "-c", code, os.path.join(os.path.dirname(_lib.LIB_PATH), "libgipvit_hip_f16.so")] + sorted(declared),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-600:]


def test_act_format_selects_the_library_build():
    """GIPVIT_ACT_FORMAT=f16 loads libgipvit_hip_f16.so and makes float16 the 16-bit tensor dtype; a mismatch is an ImportError."""
    code = ("import sys; sys.path.insert(0, %r); from gipvit import _lib, ops; "
            "print(_lib.lib.gv_act_format(), ops.bf16, _lib.LIB_PATH.endswith('libgipvit_hip_f16.so'))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GIPVIT_ACT_FORMAT="f16"), capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split() == ["1", "torch.float16", "True"], (r.stdout, r.stderr[-400:])
    from gipvit import _lib
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GIPVIT_ACT_FORMAT="f16", GIPVIT_LIB=_lib.LIB_PATH), capture_output=True, text=True)
    assert r.returncode != 0 and "built for 16-bit format 0" in r.stderr


def test_loss_scaler_state_interchanges_with_torch_gradscaler():
    """The checkpoint entry 'amp_scaler' (timm CheckpointSaver(amp_scaler=...), reference train.py:585-602) is torch's
    GradScaler.state_dict(): LossScaler reads that layout and writes a superset of it."""
    from gipvit import ops
    torch_state = torch.amp.GradScaler("cpu", init_scale=4096.0, growth_interval=100).state_dict()
    assert set(torch_state) == {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}
    sc = ops.LossScaler("cpu")
    assert sc.state.tolist() == [65536.0, 0.0, 0.0, 0.0] and (sc.growth_factor, sc.backoff_factor, sc.growth_interval) == (2.0, 0.5, 2000)
    sc.load_state_dict(dict(torch_state, _growth_tracker=7))
    assert sc.state.tolist() == [4096.0, 7.0, 0.0, 0.0] and sc.growth_interval == 100
    out = sc.state_dict()
    assert all(out[k] == v for k, v in dict(torch_state, _growth_tracker=7).items()) and out["applied_steps"] == 0
    back = torch.amp.GradScaler("cpu")
    back.load_state_dict({k: out[k] for k in torch_state})          # and torch reads ours


def test_struct_layout_matches_header():
    """sizeof of every args struct as the C compiler sees it == ctypes' view."""
    from gipvit import _lib
    structs = list(_lib.ENTRY_POINTS.values()) + [_lib.gv_linear_timing_row]
    names = sorted(s.__name__ for s in structs)
    src = '#include <stdio.h>\n#include "gipvit.h"\nint main(){' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    exe = os.path.join(ROOT, "gpurun_out", "_sizeof_test")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    r = subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src, text=True, capture_output=True)
    assert r.returncode == 0, r.stderr
    out = dict(l.split() for l in subprocess.run([exe], capture_output=True, text=True).stdout.splitlines())
    for st in structs:
        assert int(out[st.__name__]) == ctypes.sizeof(st), st.__name__


def test_bad_arguments_fail_loudly_without_gpu():
    """Argument validation happens before any launch: errors come back through the ABI."""
    from gipvit import _lib
    a = _lib.gv_linear_args()          # all NULL
    rc = _lib.lib.gv_linear(ctypes.byref(a), None)
    assert rc == -3 and b"null" in _lib.lib.gv_last_error()
    b = _lib.gv_layernorm_fwd_args(1, 100, 1, 1, 1, 1, 1, 4, 100, 1e-6)
    assert _lib.lib.gv_layernorm_fwd(ctypes.byref(b), None) == -1 and b"192" in _lib.lib.gv_last_error()
    c = _lib.gv_crop_resize_args(1, 1, 1, 4, 1, 256, 256, 98)      # out_size not a multiple of 4
    assert _lib.lib.gv_crop_resize(ctypes.byref(c), None) == -1 and b"multiple of 4" in _lib.lib.gv_last_error()
    rows = (_lib.gv_linear_timing_row * 4)()
    assert _lib.lib.gv_linear_timing_read(rows, 4) == 0             # nothing recorded: zero rows, no GPU touched
    # epilogue bits outside the documented mask (the tuning lab's ablation switches lived there) are refused
    d = _lib.gv_linear_args()
    d.A = d.B = d.C = 256; d.M = d.N = d.K = 128; d.lda = d.ldb = d.ldc = 128
    for bad in (1 << 20, 1 << 21, 1 << 22, 128):
        d.epilogue = bad
        assert _lib.lib.gv_linear(ctypes.byref(d), None) == -4 and b"GV_EPI" in _lib.lib.gv_last_error(), bad
    e = _lib.gv_adamw_ema_args(); e.p = e.grad = e.m = e.v = 256; e.n = 64; e.mode = 4
    assert _lib.lib.gv_adamw_ema(ctypes.byref(e), None) == -4
    # fp32 operand mode: same structs, its own checks
    f = _lib.gv_linear_args(); f.A = f.B = f.C = 256; f.M = f.N = f.K = 100; f.lda = f.ldb = f.ldc = 100
    assert _lib.lib.gv_linear_f32(ctypes.byref(f), None) == -4 and b"c_is_f32" in _lib.lib.gv_last_error()
    f.c_is_f32 = 1; f.epilogue = 1      # BIAS without a bias
    assert _lib.lib.gv_linear_f32(ctypes.byref(f), None) == -3
    g = _lib.gv_attention_fwd_args(256, 256, 256, 2, 300, 3, 0.125)
    assert _lib.lib.gv_attention_fwd_f32(ctypes.byref(g), None) == -1 and b"260" in _lib.lib.gv_last_error()
    from gipvit.engine import SupervisedEngine, DinoEngine
    for cls in (SupervisedEngine, DinoEngine):
        with pytest.raises(ValueError, match="precision"):
            cls(precision="fp16", device="cpu")


def test_workspace_query_runs_the_entry_points_in_plan_mode():
    """gv_workspace_bytes(op, args): the scratch one call would use, answered by the entry point's own kernel selection
    and k-slice sizing in a plan mode -- nothing is launched, no GPU is touched."""
    from gipvit import _lib
    q = lambda op, a: _lib.lib.gv_workspace_bytes(op, ctypes.byref(a))
    P = 1 << 20                                                        # any 16-byte aligned non-null value: never dereferenced
    T, D = 44160, 384
    dw = _lib.gv_linear_args(); dw.A = dw.B = dw.C = P
    dw.M, dw.N, dw.K, dw.lda, dw.ldb, dw.ldc = 4 * D, D, T, 4 * D, D, D
    dw.trans_a = dw.trans_b = dw.c_is_f32 = 1; dw.epilogue = _lib.EPI_ACCUM
    b = q(_lib.OP_LINEAR, dw)
    assert 0 < b <= _lib.lib.gv_linear_workspace_bytes() and b % (4 * D * D * 4) == 0, b      # S slabs of M x N f32
    wide = _lib.gv_linear_args(); wide.A = wide.B = wide.C = wide.bias = P
    wide.M, wide.N, wide.K, wide.lda, wide.ldb, wide.ldc = T, 4 * D, D, D, D, 4 * D
    wide.epilogue = _lib.EPI_BIAS
    assert q(_lib.OP_LINEAR, wide) == 0                                # the full-row kernel takes no scratch
    grp = _lib.gv_linear_dw_group_args(); grp.n, grp.K = 4, T
    for pr, (m, n) in zip(grp.prob, ((D, 4 * D), (4 * D, D), (D, D), (3 * D, D))):
        pr.dY = pr.X = pr.dW = P; pr.M, pr.N, pr.ldy, pr.ldx, pr.ldw = m, n, m, n, n
    g = q(_lib.OP_LINEAR_DW_GROUP, grp)
    mn = 4 * (D * 4 * D + 4 * D * D + D * D + 3 * D * D)
    assert g == 7 * mn, (g, mn)                                        # 36 tiles x 7 k-slices = 252 workgroups: DESIGN section 4's 50-MB slab
    assert q(7, wide) == -1 and b"GV_OP_LINEAR" in _lib.lib.gv_last_error()
    bad = _lib.gv_linear_args()                                        # null operands: the call itself would be rejected
    assert q(_lib.OP_LINEAR, bad) == -1


def test_arena_layout_and_decay_split():
    from collections import OrderedDict
    from gipvit import engine as E
    specs = OrderedDict(("backbone." + k, v) for k, v in E.vit_param_specs("vit_tiny", 64, 0).items())
    specs.update(("head." + k, v) for k, v in E.dino_head_specs(192, 256).items())
    a = E.Arena(specs, "cpu", teacher=True)
    assert a.n % 64 == 0 and a.n_decay % 64 == 0
    for n in specs:
        assert a.off[n] % 64 == 0
        assert (a.off[n] < a.n_decay) == (not E.no_weight_decay(n, specs[n])), n
    # gradients complete head-first: the head's matrices open the arena, patch embed closes the decayed part
    assert a.order[0] == "head.last_layer.weight_v" and a.order.index("backbone.blocks.11.mlp.fc2.weight") < a.order.index("backbone.blocks.0.attn.qkv.weight")
    assert E.no_weight_decay("backbone.pos_embed", (1, 17, 192)) and E.no_weight_decay("head.last_layer.weight_g", (256, 1))
    assert not E.no_weight_decay("backbone.patch_embed.proj.weight", (192, 3, 16, 16))
    a.view(a.p, "backbone.norm.weight").fill_(3.0)
    assert float(a.state_dict(prefix="backbone.")["norm.weight"].mean()) == 3.0


def test_model_registry_init_and_checkpoint_roundtrip(tmp_path):
    from gipvit import models as M
    assert M.resolve_arch("vit_small_patch16_224_dino") == "vit_small"
    with pytest.raises(ValueError):
        M.resolve_arch("resnet50")
    sd = M.init_vit_state("vit_tiny", 64, 2, seed=0)
    assert sd["pos_embed"].shape == (1, 17, 192) and sd["head.weight"].shape == (2, 192)
    assert float(sd["blocks.3.norm1.weight"].min()) == 1.0 and float(sd["blocks.3.attn.qkv.bias"].abs().max()) == 0.0
    assert 0.015 < float(sd["blocks.0.mlp.fc1.weight"].std()) < 0.025 and float(sd["blocks.0.mlp.fc1.weight"].abs().max()) <= 2.0
    # timm-style file: {'state_dict': {'module.<name>': ...}}, 224-px pos-embed into a 64-px model
    big = M.init_vit_state("vit_tiny", 224, 0, seed=1)
    path = tmp_path / "model_best.pth.tar"
    torch.save({"epoch": 3, "arch": "vit_tiny_patch16_224", "state_dict": {"module." + k: v for k, v in big.items()}, "version": 2}, path)
    got = M.load_encoder_checkpoint(str(path), "vit_tiny", 64, num_classes=2)
    assert got["pos_embed"].shape == (1, 17, 192) and torch.equal(got["blocks.5.mlp.fc2.weight"], big["blocks.5.mlp.fc2.weight"])
    assert got["head.weight"].shape == (2, 192)            # missing head -> freshly initialised
    assert torch.equal(got["pos_embed"][:, 0], big["pos_embed"][:, 0])


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gipvit.dist import RcclReducer, shard_range
    red = RcclReducer()
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    c = torch.full((16,), float(rank + 1))
    red.reduce_tensor(c)
    red.reduce_range(g, 0, 384)          # head range first, as the engine does
    red.reduce_range(g, 384, 1000)
    red.finish()
    q.put((rank, g.clone(), c.clone(), shard_range(64, rank, world)))
    dist.destroy_process_group()


def test_reducer_world_size_2_gloo():
    """N > 1 path on CPU: ranged all-reduces of the flat gradient arena + center sum."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:              # a port the kernel reports free (a pid-derived one collided now and then)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ps = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    for rank, g, c, shard in res:
        assert torch.equal(g, torch.arange(1000, dtype=torch.float32) * 3)   # (1 + 2) x
        assert torch.equal(c, torch.full((16,), 3.0))
        assert shard == (32 * rank, 32 * rank + 32)


def test_reduction_plan_tiles_the_arena_in_backward_order():
    """The gradient all-reduce plan of a data-parallel step (dist.reduction_plan over the real ViT-S + DINOHead arena layout):
    ranges cover [0, n) exactly once, in the order backward completes them; block ranges are coalesced to >= 25 MB (the
    reference's DDP bucket, train.py:634), and the last message is block 0 + patch embed + the no-decay tail, released at the end."""
    from collections import OrderedDict
    import math
    from gipvit import engine as E
    from gipvit.dist import reduction_plan
    specs = OrderedDict(("backbone." + k, v) for k, v in E.vit_param_specs("vit_small", 224, 0).items())
    specs.update(("head." + k, v) for k, v in E.dino_head_specs(384, 65536).items())
    a = E.Arena(specs, "meta", teacher=True)
    blocks = {}
    for i in range(12):
        names = [n for n in a.order if n.startswith(f"backbone.blocks.{i}.") and a.off[n] < a.n_decay]
        blocks[i] = (min(a.off[n] for n in names), max(a.span(n)[1] for n in names))
    head = [n for n in a.order if n.startswith("head.") and a.off[n] < a.n_decay]
    head_span = (min(a.off[n] for n in head), max(a.span(n)[1] for n in head))
    plan = reduction_plan(head_span, blocks, a.n)
    assert plan[0] == ("head", 0, head_span[1]) and plan[-1][0] == "end" and plan[-1][2] == a.n
    pos = 0
    for trg, lo, hi in plan:                                   # exact tiling, ascending = backward-completion order of the arena
        assert lo == pos and hi > lo
        pos = hi
    assert pos == a.n
    triggers = [t for t, _, _ in plan]
    assert triggers == ["head", 8, 4, 1, "end"]                # 4 + 4 + 3 blocks, then block 0 with everything behind it
    for trg, lo, hi in plan[1:-2]:
        assert (hi - lo) * 4 >= 25 << 20                       # full buckets (the one in front of the last message may be shorter)
    for trg, lo, hi in plan[1:-1]:                             # a range released by block i holds only blocks >= i
        assert lo >= blocks[11][0] and hi == blocks[trg][1] and all(j >= trg for j in range(12) if lo <= blocks[j][0] and blocks[j][1] <= hi)
    last_lo = plan[-1][1]
    assert last_lo == blocks[0][0] and a.off["backbone.patch_embed.proj.weight"] >= last_lo and a.n_decay < a.n
    assert (a.n - last_lo) * 4 < 12 << 20                      # the un-overlapped message stays short (ViT-S: 7.8 MB)
    # a one-block model and a headless one still tile
    assert reduction_plan(None, {0: (0, 100)}, 130) == [("end", 0, 130)]
    assert reduction_plan((0, 50), {1: (50, 80), 0: (80, 100)}, 130, bucket_bytes=1 << 30) == [("head", 0, 50), (1, 50, 80), ("end", 80, 130)]


def test_comm_setup_environment_contract(monkeypatch):
    """dist.comm_setup: nothing is touched at world size 1, for the gloo rehearsal transport or without GIPVIT_COMM_CUS (the
    default: RCCL's own channel count, launches sized for the whole chip); GIPVIT_COMM_CUS = C sizes the library for 256 - C and
    caps RCCL to C channels, and values already in the environment win."""
    import os
    from gipvit.dist import comm_setup
    for k in ("GIPVIT_COMM_CUS", "GIPVIT_CU_BUDGET", "NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS"):
        monkeypatch.delenv(k, raising=False)
    assert comm_setup(1, "nccl") == {"comm_cus": 0, "cu_budget": 256}
    assert comm_setup(8, "gloo") == {"comm_cus": 0, "cu_budget": 256}
    info = comm_setup(8, "nccl")
    assert info["comm_cus"] == 0 and info["cu_budget"] == 256 and "NCCL_MAX_NCHANNELS" not in os.environ and "GIPVIT_CU_BUDGET" not in os.environ
    monkeypatch.setenv("GIPVIT_COMM_CUS", "16")
    monkeypatch.setenv("GIPVIT_CU_BUDGET", "248")              # set by the caller: kept (and the library may already be loaded then)
    info = comm_setup(8, "nccl")
    assert info["comm_cus"] == 16 and info["cu_budget"] == 248 and os.environ["NCCL_MAX_NCHANNELS"] == "16" and os.environ["NCCL_MIN_NCHANNELS"] == "4"
    monkeypatch.setenv("GIPVIT_COMM_CUS", "2")
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "12")
    monkeypatch.delenv("NCCL_MIN_NCHANNELS")
    info = comm_setup(2, "nccl")
    assert info["nccl_max_nchannels"] == "12" and os.environ["NCCL_MIN_NCHANNELS"] == "2"


def test_train_cli_keeps_reference_surface(tmp_path):
    """Every flag of the reference trainer parses (documented command lines of
    train_instruct.txt:16-34 included), YAML -c defaults work, and without a GPU the driver
    refuses to run instead of falling back to a CPU path."""
    import yaml
    sys.path.insert(0, ROOT)
    import train
    from gipvit.cli_spec import REFERENCE_FLAGS
    assert len(REFERENCE_FLAGS) == 148
    cmd = ("--model vit_small_patch16_224 --dataset TCGA --target Her2 --num-classes 2 --batch-size 256 --epochs 500 --workers 2 "
           "--opt adam --lr-base 0.001 --sched cosine --warmup-epochs 20 --supervised --log-wandb --experiment e --subexperiment s").split()
    args, text = train.parse_args(cmd)
    assert args.model == "vit_small_patch16_224" and args.batch_size == 256 and args.lr_base == 0.001 and args.supervised
    assert args.smoothing == 0.1 and args.opt == "adam" and args.test_fold == 1 and args.transform_type == "rvf"
    cfg = tmp_path / "c.yaml"
    cfg.write_text(yaml.safe_dump({"epochs": 7, "batch_size": 32}))
    args, _ = train.parse_args(["-c", str(cfg), "--model", "vit_tiny", "--epochs", "9"])
    assert args.batch_size == 32 and args.epochs == 9
    if not torch.cuda.is_available():
        with pytest.raises(SystemExit, match="MI355X is required"):
            train.main(["--model", "vit_tiny", "--dataset", "synthetic"])


def test_transformations_hook_and_tile_files(tmp_path):
    import numpy as np
    from gipvit import transformations as T, data as D
    t = T.define_transformations("rvf", True, 256, seed=0)
    a = (np.arange(256 * 256 * 3) % 251).astype(np.uint8).reshape(256, 256, 3)
    out = t(a)
    assert out.shape == (256, 256, 3) and out.dtype == np.uint8 and sorted(out.ravel()) == sorted(a.ravel())
    assert t.mean == (0.8998, 0.8253, 0.9357) and T.define_transformations("none", False, 256, norm_type="Imagenet").std == (0.229, 0.224, 0.225)
    hook = lambda img: np.asarray(img)[::-1]
    assert T.define_transformations(hook, True, 256) is hook          # non-string = the user's transform (transformations.py:199-200)
    dev_t = T.define_transformations("pcbnfrsc", True, 256, color_param=0.2)
    assert dev_t.device_recipe == "pcbnfrsc" and dev_t.color_param == 0.2      # runs on the device (gipvit.augment), no CPU path
    with pytest.raises(NotImplementedError):
        dev_t(a)
    assert T.define_transformations("pcbnfrsc", False, 256).device_recipe is None          # evaluation: no augmentation
    with pytest.raises(ValueError):
        T.define_transformations("wcfrs", True, 256)                                       # not a recipe the reference defines either
    # reference raw tile format (datasets.py:452-466): "dtype w h c\\n" + bytes
    os.makedirs(tmp_path / "slideA"); os.makedirs(tmp_path / "slideB")
    for s, n in (("slideA", 3), ("slideB", 2)):
        for i in range(n):
            D.write_tile_file(str(tmp_path / s / f"tile_{i}.data"), (a + i).astype(np.uint8))
    (tmp_path / "labels.csv").write_text("slide,label\nslideA,1\nslideB,0\n")
    assert np.array_equal(D.read_tile_file(str(tmp_path / "slideA" / "tile_2.data")), (a + 2).astype(np.uint8))
    # one epoch = n_slides x n_tiles draws, draw i = a random tile of slide i % n_slides (datasets.py:428-429, 445-450)
    src = D.TileFolder(str(tmp_path), batch=2, transform=None, seed=0, n_tiles=3)
    batches = list(src)
    assert len(src) == 3 and len(batches) == 3 and batches[0]["Data"].shape == (2, 256, 256, 3) and batches[0]["Data"].dtype == torch.uint8
    assert batches[0]["Target"].shape == (2, 1) and batches[0]["Target"].dtype == torch.int64
    tg = torch.cat([b["Target"] for b in batches]).view(-1)
    assert int(tg.sum()) == 3                                         # every slide drawn n_tiles times: 3 x label 1, 3 x label 0
    for b in batches:                                                 # every tile is one of its slide's files, read byte-exactly
        for t, y in zip(b["Data"], b["Target"].view(-1)):
            assert any(np.array_equal(t.numpy(), (a + i).astype(np.uint8)) for i in range(3 if y else 2))
    # ranks get the same number of batches whatever the remainder (5 slides... here 2 x 3 = 6 draws, world 2, B 2 -> 1 batch each)
    r0, r1 = (D.TileFolder(str(tmp_path), batch=2, rank=r, world=2, seed=0, n_tiles=3) for r in (0, 1))
    assert len(r0) == len(r1) == 1
    with pytest.raises(ValueError):
        D.TileFolder(str(tmp_path), batch=64, seed=0, n_tiles=3)


def test_infer_tiles_chunks_and_labels(tmp_path):
    """Infer_Dataset's contract (datasets.py:634-817): per slide min(num_tiles, available) tiles in chunks of
    tiles_per_iter, 'Is Last Batch' on the chunk that ends a slide; labels / folds / --target from labels.csv."""
    import numpy as np
    from gipvit import data as D
    a = (np.arange(64 * 64 * 3) % 251).astype(np.uint8).reshape(64, 64, 3)
    for s, n in (("s0", 5), ("s1", 2), ("s2", 7)):
        os.makedirs(tmp_path / s)
        for i in range(n):
            D.write_tile_file(str(tmp_path / s / f"tile_{i}.data"), (a + i).astype(np.uint8))
    (tmp_path / "labels.csv").write_text("slide,label,fold,ER\ns0,1,1,Negative\ns1,0,2,Positive\ns2,1,1,Positive\n")
    inf = D.InferTiles(str(tmp_path), tile_size=64, tiles_per_iter=3, num_tiles=6, seed=0)
    assert inf.num_tiles == [5, 2, 6] and inf.image_file_names == ["s0", "s1", "s2"] and len(inf) == 2 + 1 + 2
    got = list(inf)
    assert [g["Is Last Batch"] for g in got] == [False, True, True, False, True]
    assert [g["Data"].shape[0] for g in got] == [3, 2, 2, 3, 3] and [int(g["Label"]) for g in got] == [1, 1, 0, 1, 1]
    assert got[0]["Slide Filename"] == "s0" and sorted(got[0]["Patch Loc"] + got[1]["Patch Loc"]) == [0, 1, 2, 3, 4]
    for g in got[:2]:
        for t, i in zip(g["Data"], g["Patch Loc"]):
            assert np.array_equal(t.numpy(), (a + i).astype(np.uint8))
    inf.reset_counter(); assert inf.slide_num == -1
    # --target picks a named label column ('Positive' / 'Negative' strings as in the reference's slide tables)
    sl = D.scan_slides(str(tmp_path), "ER")
    assert [s[2] for s in sl] == [0, 1, 1] and [s[3] for s in sl] == ["1", "2", "1"]
    assert [s[0] for s in D.select_fold(sl, 1, train=True)] == ["s1"] and [s[0] for s in D.select_fold(sl, 1, train=False)] == ["s0", "s2"]
    # the reference's fold rule (datasets.py:274-287): 'test' and 'val' slides are never trained on; evaluation = [test_fold, 'val'];
    # test_fold -1 trains on every numbered fold; an empty selection raises instead of handing back all slides
    mk = lambda folds: [(f"s{i}", [], 0, f) for i, f in enumerate(folds)]
    sl5 = mk(["1", "2", "3", "test", "val", "1"])
    names = lambda xs: [x[0] for x in xs]
    assert names(D.select_fold(sl5, 1, True)) == ["s1", "s2"] and names(D.select_fold(sl5, 1, False)) == ["s0", "s4", "s5"]
    assert names(D.select_fold(sl5, -1, True)) == ["s0", "s1", "s2", "s5"] and names(D.select_fold(sl5, 0, False)) == ["s3", "s4"]
    assert names(D.select_fold(sl5, 0, True)) == ["s0", "s1", "s2", "s5"]
    with pytest.raises(ValueError):
        D.select_fold(mk(["1", "1"]), -1, False)            # the reference evaluates on nothing for -1
    with pytest.raises(ValueError):
        D.select_fold(mk(["test", "val"]), 2, True)
    with pytest.raises(ValueError):
        D.select_fold(mk(["1", "2"]), 3, False)
    assert names(D.select_fold(mk([None, None]), 1, True)) == ["s0", "s1"] == names(D.select_fold(mk([None, None]), 1, False))
    syn = D.SyntheticSlides(n_slides=3, tiles_per_slide=5, tile_size=32, tiles_per_iter=2)
    chunks = list(syn)
    assert len(chunks) == len(syn) == 9 and sum(c["Is Last Batch"] for c in chunks) == 3 and chunks[2]["Data"].shape == (1, 32, 32, 3)


def test_validate_metrics_host_logic():
    from gipvit.validate import accuracy_topk
    import numpy as np
    prob = np.array([[0.9, 0.1], [0.2, 0.8], [0.6, 0.4], [0.3, 0.7]])
    a1, a5 = accuracy_topk(prob, np.array([0, 1, 1, 1]))
    assert a1 == 75.0 and a5 == 100.0                                  # top-5 of 2 classes always hits (timm semantics)


def test_every_used_flag_is_read_by_the_driver():
    """CLI honesty: a reference flag marked used=True in cli_spec must be read (``args.<dest>``) by train.py; a flag the
    driver never looks at must be marked used=False so it lands in the 'accepted, ignored' warning."""
    sys.path.insert(0, ROOT)
    import train
    from gipvit.cli_spec import REFERENCE_FLAGS
    src = open(os.path.join(ROOT, "train.py")).read()
    for e in REFERENCE_FLAGS:
        dest = train.flag_dest(e)
        reads = len(re.findall(r"args\." + re.escape(dest) + r"\b", src))
        if e["used"]:
            assert reads > 0, f"{e['flags']} is marked used but train.py never reads args.{dest}"
        elif e["flags"][0] != "data":
            assert reads == 0, f"{e['flags']} is read by train.py but marked used=False"
    # values the build cannot honour are refused before any GPU work
    for bad in (["--drop", "1.0"], ["--drop-path", "1.0"], ["--drop-connect", "0.1"], ["--pretrained"], ["--clip-mode", "adaptive"], ["--clip-mode", "agc", "--dino"], ["--clip-mode", "value", "--opt", "lamb"], ["--in-chans", "1"],
                ["--input-size", "3", "224", "200"], ["--dino", "--supervised"], ["--amp", "--amp-dtype", "bfloat16", "--precision", "fp32"],
                ["--opt", "lars"], ["--opt", "rmsprop"], ["--opt", "nadam"], ["--opt", "momentum"], ["--sched", "tanh"], ["--sched", "plateau"],
                ["--sched", "multistep"], ["--sched", "poly"], ["--dino", "--opt", "sgd"], ["--dino", "--opt", "adam"], ["--dino", "--opt", "lamb"]):
        args, _ = train.parse_args(["--model", "vit_tiny"] + bad)
        with pytest.raises(SystemExit):
            train.check_supported(args, lambda m: None)
    args, _ = train.parse_args(["--model", "vit_tiny", "--input-size", "3", "64", "64", "--amp", "--amp-dtype", "bfloat16", "--drop-path", "0.1"])
    assert train.check_supported(args, lambda m: None) == 64
    # every accepted VALUE of --opt / --sched selects its own arithmetic (nothing is mapped onto a neighbour): the optimizer modes are
    # distinct kernel modes, the schedules distinct curves; --dino moves the default of --opt to the recipe's adamw
    from gipvit import sched as S
    for o in train.SUPPORTED_OPTS:
        a, _ = train.parse_args(["--model", "vit_tiny", "--opt", o]); train.check_supported(a, lambda m: None)
    from gipvit.engine import SupervisedEngine
    import inspect
    assert all(f'"{o}"' in inspect.getsource(SupervisedEngine.__init__) for o in train.SUPPORTED_OPTS)
    curves = {sc: [S.LrSchedule(1.0, sc, epochs=10, warmup_epochs=0, decay_epochs=3).at(e) for e in range(10)] for sc in train.SUPPORTED_SCHEDS}
    assert curves["cosine"] != curves["step"] and len(set(curves["step"])) == 4
    with pytest.raises(ValueError):
        S.LrSchedule(1.0, "tanh")
    a, _ = train.parse_args(["--model", "vit_tiny", "--dino"])
    assert a.opt == "adamw" and train.check_supported(a, lambda m: None) is None
    a, _ = train.parse_args(["--model", "vit_tiny"])
    assert a.opt == "sgd"
    # every accepted value of --amp-dtype selects its own arithmetic: float16 (the reference default) = the float16 build of the
    # library + dynamic loss scaling, bfloat16 = the default build; anything else, and the pairings the scaler is not built for, raise
    a, _ = train.parse_args(["--model", "vit_tiny", "--amp"])
    assert a.amp_dtype == "float16" and train.amp_is_f16(a) and train.check_supported(a, lambda m: None) is None
    a, _ = train.parse_args(["--model", "vit_tiny", "--amp", "--amp-dtype", "bfloat16"])
    assert not train.amp_is_f16(a)
    a, _ = train.parse_args(["--model", "vit_tiny", "--amp-dtype", "float16"])
    assert not train.amp_is_f16(a)                                                     # without --amp the dtype is not consulted (train.py:452-465)
    for bad in (["--amp", "--amp-dtype", "float8"], ["--amp", "--opt", "lamb"], ["--amp", "--clip-mode", "agc", "--clip-grad", "0.1"]):
        a, _ = train.parse_args(["--model", "vit_tiny"] + bad)
        with pytest.raises(SystemExit):
            train.check_supported(a, lambda m: None)
    # --drop-path draws (gipvit.droppath): block 0 never drops, factors are 0 or 1 / keep, expectation 1
    import numpy as np
    from gipvit.droppath import DropPathSampler
    sp = DropPathSampler(12, 4000, 0.2, seed=3, device="cpu")
    f = sp.sample_host()
    assert f.shape == (12, 2, 4000) and float(f[0].min()) == 1.0 and set(np.unique(f[11]).tolist()) == {0.0, np.float32(1.0 / 0.8)}
    assert abs(float(f[11].mean()) - 1.0) < 0.04
    assert torch.equal(sp.sample().cpu(), sp.dev[sp.k].cpu())
    with pytest.raises(ValueError):
        DropPathSampler(12, 8, 1.0, 0, "cpu")
    ns = lambda **k: type("A", (), dict(warmup_teacher_temp=0.04, teacher_temp=0.07, **k))
    assert train.teacher_temp_at(ns(warmup_teacher_temp_epochs=0), 0) == 0.07          # no warm-up: the final temperature from step 0
    assert train.teacher_temp_at(ns(warmup_teacher_temp_epochs=3), 0) == 0.04 and abs(train.teacher_temp_at(ns(warmup_teacher_temp_epochs=3), 1) - 0.055) < 1e-12
    assert train.teacher_temp_at(ns(warmup_teacher_temp_epochs=3), 2) == 0.07 and train.teacher_temp_at(ns(warmup_teacher_temp_epochs=3), 9) == 0.07


def test_bench_launcher_spawns_ranks_with_torchrun_env(tmp_path):
    """`python bench.py --gpus N` without WORLD_SIZE is only a launcher: N children with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, rank 0's stdout relayed.  Exercised with a stub child (no GPU)."""
    sys.path.insert(0, ROOT)
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text("import os, sys, json\n"
                    "e = {k: os.environ.get(k) for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}\n"
                    "open(os.path.join(os.path.dirname(__file__), 'rank%s.json' % e['RANK']), 'w').write(json.dumps([e, sys.argv[1:]]))\n"
                    "print(json.dumps({'rank': e['RANK'], 'rccl_ranks': int(e['WORLD_SIZE'])}))\n"
                    "sys.exit(3 if e['RANK'] == '2' and '--fail' in sys.argv else 0)\n")
    import json
    rc, out = bench.launch_children(3, ["--gpus", "3", "--steps", "2"], child=[sys.executable, str(stub)], timeout=60)
    assert rc == 0 and json.loads(out) == {"rank": "0", "rccl_ranks": 3}
    envs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(3)]
    assert [e[0]["RANK"] for e in envs] == ["0", "1", "2"] and all(e[0]["WORLD_SIZE"] == "3" and e[0]["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert len({e[0]["MASTER_PORT"] for e in envs}) == 1 and envs[1][1] == ["--gpus", "3", "--steps", "2"]
    rc, _ = bench.launch_children(3, ["--fail"], child=[sys.executable, str(stub)], timeout=60)
    assert rc == 3                                                       # a failing rank fails the launch
    # a rank other than 0 dies at start-up while rank 0 sits in its rendezvous: the launcher must notice the dead child, kill the
    # rest and return that rank's code at once (it used to block on rank 0's pipe until the backend's timeout)
    hang = tmp_path / "hang.py"
    hang.write_text("import os, sys, time\n"
                    "if os.environ['RANK'] == '1':\n"
                    "    sys.stderr.write('rank 1: no such device\\n'); sys.exit(7)\n"
                    "time.sleep(600)\n")
    import time
    t0 = time.monotonic()
    rc, out = bench.launch_children(3, [], child=[sys.executable, str(hang)], timeout=120)
    assert rc == 7 and out == "" and time.monotonic() - t0 < 30, (rc, out, time.monotonic() - t0)
    t0 = time.monotonic()                                                # nobody fails, nobody finishes: the bound ends the job
    rc, _ = bench.launch_children(2, [], child=[sys.executable, "-c", "import time; time.sleep(600)"], timeout=2)
    assert rc == 124 and time.monotonic() - t0 < 30
    # end to end through main(): bare `--gpus 2` in a process without WORLD_SIZE goes to the launcher, not to the GPU
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", "import sys; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1']\n"
                        "import bench\n"
                        f"bench.launch_children = lambda n, argv, **k: (0, 'LAUNCHED %d %s\\n' % (n, ' '.join(argv)))\n"
                        "bench.main()"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "LAUNCHED 2 --gpus 2 --steps 1" in r.stdout, (r.stdout, r.stderr)


def test_lr_schedule_and_checkpoint_saver(tmp_path):
    from gipvit import sched as S
    from gipvit.checkpoint import CheckpointSaver, load_checkpoint_file
    assert abs(S.scaled_lr(None, 0.1, 128, 2) - 0.1) < 1e-12                      # linear: 256/256
    assert abs(S.scaled_lr(None, 0.001, 256, 4, opt="adam") - 0.002) < 1e-12     # sqrt for ada*
    sc = S.LrSchedule(1.0, "cosine", epochs=10, warmup_epochs=2, warmup_lr=0.0, min_lr=0.0)
    assert sc.at(0) == 0.0 and sc.at(1) == 0.5 and abs(sc.at(5) - 0.5) < 1e-12 and sc.at(10) < 1e-12
    assert S.cosine_between(0.996, 1.0, 0, 100) == 0.996 and abs(S.cosine_between(0.996, 1.0, 99, 100) - 1.0) < 1e-12
    sv = CheckpointSaver(str(tmp_path), "vit_tiny", {"lr": 0.1, "obj": object()}, decreasing=True, max_history=2)
    sd = {"w": torch.ones(3)}
    for ep, m in enumerate((3.0, 1.0, 2.0)):
        best, bep = sv.save_checkpoint(ep, sd, {"exp_avg": torch.zeros(3), "step": ep}, metric=m)
    assert (best, bep) == (1.0, 1)
    names = sorted(os.listdir(tmp_path))
    assert names == ["checkpoint-1.pth.tar", "checkpoint-2.pth.tar", "last.pth.tar", "model_best.pth.tar"]
    ck = load_checkpoint_file(str(tmp_path / "model_best.pth.tar"))          # weights_only loader
    assert ck["epoch"] == 1 and ck["version"] == 2 and ck["arch"] == "vit_tiny" and ck["args"] == {"lr": 0.1} and ck["metric"] == 1.0


def test_create_model_rejects_what_is_not_built():
    """models.create_model (the reference's create_model seam) refuses unsupported requests before touching a GPU."""
    from gipvit import models
    with pytest.raises(ValueError, match="unknown model"):
        models.create_model("resnet50")
    with pytest.raises(ValueError, match="in_chans"):
        models.create_model("vit_tiny", in_chans=1)
    with pytest.raises(ValueError, match="drop"):
        models.create_model("vit_small_patch16_224", drop_rate=1.0)
    with pytest.raises(ValueError, match="checkpoint_path"):
        models.create_model("vit_small_patch16_224", pretrained=True)
