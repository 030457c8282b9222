#!/usr/bin/env python3
"""Training driver for the MI355X hot path -- drop-in for the reference's ``train.py`` on the
ViT path only.

    python train.py --model vit_small_patch16_224 --dataset synthetic --num-classes 2 \
        --batch-size 256 --epochs 2 --opt adam --lr-base 0.001 --sched cosine --warmup-epochs 1
    torchrun --nproc_per_node=8 train.py --dino --model vit_small_patch16_224 --batch-size 64 ...

What it keeps from the reference (file:line = /root/reference/train.py):
  * the whole CLI surface (83-393, table in gipvit/cli_spec.py) + ``-c FILE`` YAML defaults (396-410);
  * step order (1044-1078): forward -> softmax -> LabelSmoothingCE -> backward -> clip -> optimizer;
  * lr = lr_base * global_batch / 256 (569-581), cosine/step schedule with warm-up (881-887);
  * the ``Train: ep [i/n] Loss .. Time .. rate/s LR .. Data ..`` log line (1096-1111);
  * output folder layout, ``args.yaml``, ``summary.csv``, checkpoint file names (854-879, 958-973);
  * batch dict keys 'Data' / 'Target' (1027-1028) and the ``define_transformations`` hook.
Deliberate deviations: the device is threaded through (the reference is
hard-wired to cuda and unrunnable on CPU -- here a GPU is REQUIRED, there is no CPU fallback);
W&B is optional; the LR is stepped every update with --sched-on-updates (the reference steps it only
inside the log-interval block, 1087/1130-1137); AUC is computed at log time from device-side
probabilities instead of a per-step D2H sync (1054); tiles are sharded over ranks.
``--dino`` adds the DINO multi-crop SSL step the north-star names (absent from the reference).
"""
from __future__ import annotations

import os as _os
# The step uses three streams with cross-stream waits (main, the side stream of engine.py, RCCL's
# communication stream).  HIP maps streams onto 4 hardware queues by default and a stream's wait
# blocks everything behind it in a shared queue: measured 20.2 vs 18.2 ms/step with RCCL in the
# picture.  Must be set before the HIP runtime initialises.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import argparse
import csv
import logging
import os
import sys
import time
from collections import OrderedDict

import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gipvit.cli_spec import REFERENCE_FLAGS  # noqa: E402

_logger = logging.getLogger("train")
_TYPES = {"int": int, "float": float, "str": str}


def build_parser():
    cfg = argparse.ArgumentParser(add_help=False)
    cfg.add_argument("-c", "--config", default="", type=str, metavar="FILE")
    p = argparse.ArgumentParser(description="MI355X ViT / DINO trainer (reference-compatible CLI)")
    for e in REFERENCE_FLAGS:
        kw = {k: e[k] for k in ("action", "default", "nargs", "const", "dest") if k in e}
        if "type" in e:
            kw["type"] = _TYPES[e["type"]]
        kw["help"] = ("" if e["used"] else "[accepted, ignored by this build] ") + f"reference train.py:{e['ref']}"
        p.add_argument(*e["flags"], **kw)
    g = p.add_argument_group("MI355X build additions")
    g.add_argument("--dino", action="store_true", help="DINO multi-crop self-supervised step (2 global + N local crops)")
    g.add_argument("--out-dim", type=int, default=65536)
    g.add_argument("--local-crops-number", type=int, default=8)
    g.add_argument("--global-crop-size", type=int, default=224)
    g.add_argument("--local-crop-size", type=int, default=96)
    g.add_argument("--random-crops", action="store_true", help="DINO: random-resized crops + flips cut on the device every step "
                   "(gv_crop_resize); default = the fixed parity windows")
    g.add_argument("--global-crops-scale", type=float, nargs=2, default=(0.4, 1.0))
    g.add_argument("--local-crops-scale", type=float, nargs=2, default=(0.05, 0.4))
    g.add_argument("--momentum-teacher", type=float, default=0.996)
    g.add_argument("--teacher-temp", type=float, default=0.04)
    g.add_argument("--warmup-teacher-temp", type=float, default=0.04)
    g.add_argument("--warmup-teacher-temp-epochs", type=int, default=0)
    g.add_argument("--weight-decay-end", type=float, default=0.4)
    g.add_argument("--freeze-last-layer", type=int, default=1, help="epochs during which the head's last layer is not updated")
    g.add_argument("--tile-size", type=int, default=256)
    g.add_argument("--batches-per-epoch", type=int, default=100, help="synthetic source only")
    g.add_argument("--graph", action="store_true", help="EXPERIMENTAL: capture the DINO step in a hipGraph (DESIGN.md section 7)")
    g.add_argument("--device", default="cuda", help="must be a GPU: the hot path has no CPU fallback")
    return cfg, p


def parse_args(argv=None):
    cfg, p = build_parser()
    known, remaining = cfg.parse_known_args(argv)
    if known.config:
        with open(known.config) as f:
            p.set_defaults(**yaml.safe_load(f))
    args = p.parse_args(remaining)
    return args, yaml.safe_dump(vars(args), default_flow_style=False)


class Meter:
    def __init__(self):
        self.val = self.sum = self.n = 0.0

    def update(self, v, k=1):
        self.val, self.sum, self.n = v, self.sum + v * k, self.n + k

    @property
    def avg(self):
        return self.sum / max(self.n, 1)


def main(argv=None):
    args, args_text = parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    if not torch.cuda.is_available() or not str(args.device).startswith("cuda"):
        raise SystemExit("train.py: an MI355X is required (device=%s, cuda available=%s); the HIP hot path has no CPU fallback"
                         % (args.device, torch.cuda.is_available()))
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", args.local_rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    reducer = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)         # RCCL
        from gipvit.dist import RcclReducer
        reducer = RcclReducer()
    primary = rank == 0
    torch.manual_seed(args.seed + rank)                        # utils.random_seed(seed, rank), train.py:467

    ignored = [e["flags"][-1] for e in REFERENCE_FLAGS if not e["used"] and e["flags"][-1].startswith("-")
               and getattr(args, (e.get("dest") or e["flags"][-1].lstrip("-")).replace("-", "_"), None) not in (e.get("default"), None, False)]
    if ignored and primary:
        _logger.warning("flags accepted for CLI compatibility but outside this build's hot path: %s", " ".join(ignored))

    from gipvit import models as M, sched as S, data as D, transformations as T
    from gipvit.engine import DinoEngine, SupervisedEngine
    from gipvit.checkpoint import CheckpointSaver, load_checkpoint_file
    arch = M.resolve_arch(args.model)
    B = args.batch_size
    mean = tuple(args.mean) if args.mean else T.MEAN["Ron"]
    std = tuple(args.std) if args.std else T.STD["Ron"]
    tile = args.tile_size
    opt = args.opt.lower()
    lr = S.scaled_lr(args.lr, args.lr_base, B, world, args.lr_base_size, args.lr_base_scale, opt)
    if primary:
        _logger.info(f"Learning rate ({lr}) calculated from base learning rate ({args.lr_base}) and global batch size ({B * world})")
    betas = tuple(args.opt_betas) if args.opt_betas else (0.9, 0.999)
    eps = args.opt_eps if args.opt_eps is not None else 1e-8

    # ---- data (batch dict contract of the reference: 'Data', 'Target')
    transform = T.define_transformations(args.transform_type if args.dataset not in ("", "synthetic") else "none", True, tile,
                                         args.c_param, "Ron")
    if args.dataset in ("", "synthetic"):
        source = D.SyntheticTiles(B, tile, args.batches_per_epoch, args.num_classes or 2, seed=args.seed + rank)
    elif args.dataset.startswith("tiles:") or args.data_dir:
        root = args.dataset[6:] if args.dataset.startswith("tiles:") else args.data_dir
        source = D.TileFolder(root, B, transform, rank, world, args.seed, tile)
    else:
        raise SystemExit(f"--dataset {args.dataset}: whole-slide datasets need openslide and are outside this build "
                         "(SURVEY section 2 #15); use 'synthetic' or 'tiles:<dir>' with pre-extracted tile_<i>.data files")
    updates_per_epoch = len(source)

    # ---- model + engine
    if args.dino:
        img = args.global_crop_size
        eng = DinoEngine(arch=arch, img_size=img, out_dim=args.out_dim, batch=B, tile=tile, n_local=args.local_crops_number,
                         gsize=args.global_crop_size, lsize=args.local_crop_size, lr=lr, weight_decay=args.weight_decay, betas=betas, eps=eps,
                         momentum_teacher=args.momentum_teacher, teacher_temp=args.teacher_temp, clip_grad=args.clip_grad or 0.0,
                         mean=mean, std=std, device=dev, reducer=reducer)
        bb = (M.load_encoder_checkpoint(args.initial_checkpoint, arch, img) if args.initial_checkpoint
              else M.init_vit_state(arch, img, 0, seed=args.seed))
        eng.load_state(bb, M.init_dino_head_state(eng.D, args.out_dim, seed=args.seed + 1))
    else:
        img = args.img_size or tile          # the reference patches timm's default cfg to 256 (train_instruct.txt:9-13)
        nc = args.num_classes or 2
        eng = SupervisedEngine(arch=arch, img_size=img, num_classes=nc, batch=B, lr=lr, weight_decay=args.weight_decay, betas=betas, eps=eps,
                               smoothing=args.smoothing, clip_grad=args.clip_grad or 0.0, mean=mean, std=std, device=dev, reducer=reducer,
                               opt=opt if opt in ("adam", "adamw", "sgd") else "adamw", momentum=args.momentum,
                               train_backbone=not args.no_grad)
        st = (M.load_encoder_checkpoint(args.initial_checkpoint, arch, img, nc) if args.initial_checkpoint
              else M.init_vit_state(arch, img, nc, seed=args.seed))
        eng.load_state(st)
    start_epoch = args.start_epoch or 0
    if args.resume:
        ck = load_checkpoint_file(args.resume)
        sd = {k[7:] if k.startswith("module.") else k: v for k, v in ck["state_dict"].items()}
        if args.dino:
            eng.load_state({k[9:]: v for k, v in sd.items() if k.startswith("backbone.")}, {k[5:]: v for k, v in sd.items() if k.startswith("head.")})
        else:
            eng.load_state(sd)
        if "optimizer" in ck and not args.no_resume_opt:
            eng.arena.m.copy_(ck["optimizer"]["exp_avg"]); eng.arena.v.copy_(ck["optimizer"]["exp_avg_sq"]); eng.t = int(ck["optimizer"]["step"])
        start_epoch = args.start_epoch if args.start_epoch is not None else ck.get("epoch", -1) + 1
    if primary:
        n_params = sum(int(torch.tensor(s).prod()) for s in eng.arena.specs.values())
        _logger.info(f"Model {args.model} ({arch}) created, param count:{n_params}")

    # ---- output dir, args.yaml, saver (train.py:854-879)
    saver = output_dir = None
    if primary:
        exp = args.experiment or "-".join([time.strftime("%Y%m%d-%H%M%S"), args.model.replace("/", "_"), str(img)])
        output_dir = os.path.join(args.output or "./output/train", exp, args.subexperiment or "")
        os.makedirs(output_dir, exist_ok=True)
        saver = CheckpointSaver(output_dir, args.model, vars(args), decreasing=True, max_history=args.checkpoint_hist)
        with open(os.path.join(output_dir, "args.yaml"), "w") as f:
            f.write(args_text)
    schedule = S.LrSchedule(lr, args.sched, args.epochs, args.warmup_epochs, args.warmup_lr, args.min_lr, updates_per_epoch,
                            args.decay_epochs, args.decay_rate, on_updates=args.sched_on_updates or args.dino)
    total_updates = args.epochs * updates_per_epoch
    use_graph = args.dino and args.graph
    feats_out = []
    sampler = None
    if args.dino and args.random_crops:       # DINO recipe: random-resized crops + flips, cut on the device
        from gipvit.multicrop import MultiCropSampler
        sampler = MultiCropSampler(B, tile, 2, args.local_crops_number, tuple(args.global_crops_scale), tuple(args.local_crops_scale),
                                   seed=args.seed + rank)

    # the step's critical path runs on a high-priority stream; the engine's side stream (teacher forward,
    # weight-gradient GEMMs) fills the CU slots it leaves
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.Stream(dev, priority=-1))
    # ---- epoch loop (train.py:905-977) / step loop (988-1143)
    for epoch in range(start_epoch, args.epochs):
        batch_time, data_time, losses = Meter(), Meter(), Meter()
        probs, targets = [], []
        end = time.time()
        last_idx = updates_per_epoch - 1
        if args.dino:
            eng.train_last_layer = epoch >= args.freeze_last_layer
            if use_graph and eng.graph is not None and (epoch == args.freeze_last_layer):
                eng.graph = None                      # the captured step changes when the last layer thaws
        for batch_idx, mb in enumerate(source):
            data = mb["Data"].to(dev, non_blocking=True)
            target = mb["Target"].to(dev, non_blocking=True)
            data_time.update(time.time() - end)
            cur_lr = schedule.at(epoch, batch_idx)
            if args.extract_features and not args.dino:
                logits, feats = eng.forward(data)
                feats_out.append(feats.float().cpu())
                loss_t = eng.loss
            elif args.dino:
                it = epoch * updates_per_epoch + batch_idx
                sch = dict(lr=cur_lr, wd=S.cosine_between(args.weight_decay, args.weight_decay_end, it, total_updates),
                           momentum_teacher=S.cosine_between(args.momentum_teacher, 1.0, it, total_updates),
                           teacher_temp=(args.warmup_teacher_temp + (args.teacher_temp - args.warmup_teacher_temp)
                                         * min(1.0, epoch / max(1, args.warmup_teacher_temp_epochs))))
                if use_graph:
                    if eng.graph is None:
                        eng.capture(data)
                    loss_t = eng.step_graph(data, **sch)
                else:
                    loss_t = eng.step(data, boxes=sampler.sample(dev) if sampler is not None else None, **sch)
            else:
                loss_t = eng.step(data, target, lr=cur_lr)
                probs.append(eng.prob[:, 1].clone() if eng.C > 1 else eng.prob[:, 0].clone()); targets.append(target.view(-1))
            torch.cuda.synchronize()                  # train.py:1083
            batch_time.update(time.time() - end)
            if batch_idx == last_idx or batch_idx % args.log_interval == 0:
                lv = float(loss_t)
                if world > 1:
                    t = torch.tensor([lv], device=dev)
                    torch.distributed.all_reduce(t)   # utils.reduce_tensor, train.py:1091-1093
                    lv = float(t) / world
                losses.update(lv, B)
                if primary:
                    _logger.info("Train: {} [{:>4d}/{} ({:>3.0f}%)]  Loss: {:#.4g} ({:#.3g})  Time: {:.3f}s, {:>7.2f}/s  ({:.3f}s, {:>7.2f}/s)  "
                                 "LR: {:.3e}  Data: {:.3f} ({:.3f})".format(
                                     epoch, batch_idx, updates_per_epoch, 100.0 * batch_idx / max(last_idx, 1), losses.val, losses.avg,
                                     batch_time.val, B * world / batch_time.val, batch_time.avg, B * world / batch_time.avg, cur_lr,
                                     data_time.val, data_time.avg))
                if saver is not None and args.recovery_interval and (batch_idx + 1) % args.recovery_interval == 0:
                    saver.save_recovery(epoch, batch_idx, eng.arena.state_dict())
            end = time.time()
        # ---- end of epoch: metrics, summary.csv, checkpoint (train.py:958-973)
        metrics = OrderedDict(loss=losses.avg)
        if probs:
            try:
                from sklearn.metrics import roc_auc_score
                metrics["auc"] = float(roc_auc_score(torch.cat(targets).cpu().numpy(), torch.cat(probs).float().cpu().numpy()))
            except Exception:                       # single-class epoch etc. (the reference would raise, train.py:1054)
                metrics["auc"] = float("nan")
        if primary:
            row = OrderedDict(epoch=epoch, **{"train_" + k: v for k, v in metrics.items()}, lr=cur_lr)
            fn = os.path.join(output_dir, "summary.csv")
            new = not os.path.exists(fn)
            with open(fn, "a") as f:
                w = csv.DictWriter(f, fieldnames=row.keys())
                if new:
                    w.writeheader()
                w.writerow(row)
            optim = {"exp_avg": eng.arena.m, "exp_avg_sq": eng.arena.v, "step": eng.t}
            best, best_ep = saver.save_checkpoint(epoch, eng.arena.state_dict(), optim, metric=metrics["loss"])
            _logger.info(f"*** epoch {epoch}: " + "  ".join(f"{k} {v:.4f}" for k, v in metrics.items()) + f"  (best loss {best:.4f} @ {best_ep})")
    if args.extract_features and primary and feats_out:
        os.makedirs("./TCGA_500", exist_ok=True)      # train.py:1281-1282 writes <slide>_features.pt here
        torch.save(torch.cat(feats_out), os.path.join("./TCGA_500", "synthetic_features.pt"))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
