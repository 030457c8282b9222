#!/usr/bin/env python3
"""Training driver for the MI355X hot path -- drop-in for the reference's ``train.py`` on the
ViT path only.

    python train.py --model vit_small_patch16_224 --dataset synthetic --num-classes 2 \
        --batch-size 256 --epochs 2 --opt adam --lr-base 0.001 --sched cosine --warmup-epochs 1
    torchrun --nproc_per_node=8 train.py --dino --model vit_small_patch16_224 --batch-size 64 ...

What it keeps from the reference (file:line = /root/reference/train.py):
  * the whole CLI surface (83-393, table in gipvit/cli_spec.py) + ``-c FILE`` YAML defaults (396-410);
  * step order (1044-1078): forward -> softmax -> LabelSmoothingCE -> backward -> clip -> optimizer -> model EMA;
  * lr = lr_base * global_batch / 256 (569-581), cosine/step schedule with warm-up (881-887);
  * the ``Train: ep [i/n] Loss .. Time .. rate/s LR .. Data ..`` log line (1096-1111);
  * the epoch loop (905-977): train -> slide-level ``validate`` (933, 1146-1345) -> EMA validate (943-955) ->
    summary.csv -> checkpoint on ``--eval-metric`` (970-973); ``--extract_features`` skips training and writes
    ``./TCGA_500/<slide>_features.pt`` (906, 1281-1282);
  * output folder layout, ``args.yaml``, checkpoint file names / keys incl. ``state_dict_ema`` (854-879);
  * batch dict keys 'Data' / 'Target' (1027-1028) and the ``define_transformations`` hook.
Deliberate deviations: the device is threaded through (the reference is
hard-wired to cuda and unrunnable on CPU -- here a GPU is REQUIRED, there is no CPU fallback);
W&B is optional; the LR is stepped every update with --sched-on-updates (the reference steps it only
inside the log-interval block, 1087/1130-1137); AUC is computed at log time from device-side
probabilities instead of a per-step D2H sync (1054); tiles are sharded over ranks.
``--dino`` adds the DINO multi-crop SSL step the north-star names (absent from the reference).
Every reference flag marked ``used`` in gipvit/cli_spec.py is read below (tests/test_host.py enforces it); a value
this build cannot honour is REJECTED, never silently ignored.
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
import json
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


def flag_dest(e) -> str:
    """argparse's dest for a cli_spec entry: explicit ``dest``, else the first long option."""
    if e.get("dest"):
        return e["dest"]
    longs = [f for f in e["flags"] if f.startswith("--")]
    return (longs[0] if longs else e["flags"][0]).lstrip("-").replace("-", "_")


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
    g.add_argument("--view-augment", action="store_true", help="DINO, with --random-crops: each crop gets its own ColorJitter / grayscale / "
                   "3x3 Gaussian blur / solarisation (DINO's DataAugmentationDINO) in the pass that cuts it (gv_crop_augment)")
    g.add_argument("--color-jitter-p", type=float, default=0.8, help="--view-augment: probability of the ColorJitter(0.4, 0.4, 0.2, 0.1)")
    g.add_argument("--gray-p", type=float, default=0.2, help="--view-augment: probability of a grayscale view")
    g.add_argument("--blur-p", type=float, nargs=3, default=(1.0, 0.1, 0.5), help="--view-augment: blur probability of global crop 1 / 2 / the local crops")
    g.add_argument("--solarize-p", type=float, default=0.2, help="--view-augment: solarisation probability of global crop 2")
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
    g.add_argument("--synthetic-slides", type=int, default=4, help="synthetic source only: slides of the inference set")
    g.add_argument("--no-validate", action="store_true", help="skip the per-epoch slide-level validation")
    g.add_argument("--features-dir", default="./TCGA_500", help="where --extract_features writes <slide>_features.pt (train.py:1282)")
    g.add_argument("--device", default="cuda", help="must be a GPU: the hot path has no CPU fallback")
    g.add_argument("--precision", default="bf16", choices=("bf16", "fp32"), help="bf16: bf16 GEMM / attention operands, f32 accumulation, f32 "
                   "residual stream and master weights (the training path).  fp32: every operand f32 -- the reference's arithmetic "
                   "without --amp; a verification mode, an order of magnitude slower")
    return cfg, p


def parse_args(argv=None):
    cfg, p = build_parser()
    known, remaining = cfg.parse_known_args(argv)
    from_file = {}
    if known.config:
        with open(known.config) as f:
            from_file = yaml.safe_load(f) or {}
            p.set_defaults(**from_file)
    args = p.parse_args(remaining)
    # --dino (this build's addition) changes the DEFAULT of --opt from the reference's 'sgd' to the recipe's 'adamw'; an --opt the
    # user did pass is never replaced (check_supported refuses what the DINO step cannot honour)
    if args.dino and "opt" not in from_file and not any(a == "--opt" or a.startswith("--opt=") for a in remaining):
        args.opt = "adamw"
    return args, yaml.safe_dump(vars(args), default_flow_style=False)


SUPPORTED_OPTS = ("sgd", "adam", "adamw", "lamb")  # modes of gv_adamw_ema (sgd = momentum + nesterov, timm's default for 'sgd') + gv_lamb
SUPPORTED_SCHEDS = ("cosine", "step")               # gipvit/sched.py


def check_supported(args, log=_logger.warning):
    """Reference flags whose value this build cannot honour are rejected loudly; the ones it maps onto its own
    arithmetic are explained once.  Returns the resolved image size (or None)."""
    if args.drop and not 0.0 <= args.drop < 1.0:
        raise SystemExit(f"--drop {args.drop}: dropout needs 0 <= rate < 1 (reference train.py:283-284, vit.pyc@L98-131)")
    if args.drop_path is not None and not 0.0 <= args.drop_path < 1.0:
        raise SystemExit(f"--drop-path {args.drop_path}: stochastic depth needs 0 <= rate < 1 (reference train.py:287-288, vit.pyc@L66-74)")
    if args.drop_connect:
        raise SystemExit("--drop-connect is the deprecated spelling of --drop-path (reference train.py:285-286): pass --drop-path")
    if args.pretrained and not args.initial_checkpoint:
        raise SystemExit("--pretrained downloads weights by URL (reference train.py:482-485): there is no network here -- "
                         "pass the file with --initial-checkpoint instead")
    if args.clip_mode not in ("norm", "value", "agc"):
        raise SystemExit(f"--clip-mode {args.clip_mode}: 'norm' (global norm, the reference default), 'value' (element-wise clamp) or 'agc' (adaptive, "
                         "per unit) -- the three modes of timm's dispatch_clip_grad (train.py:1072-1077)")
    if args.clip_mode == "agc" and args.dino:
        raise SystemExit("--clip-mode agc with --dino is not built (the reference applies it to a supervised model minus its classifier)")
    if args.clip_mode == "value" and args.opt.lower() == "lamb":
        raise SystemExit("--clip-mode value with --opt lamb is not built (Lamb's own global-norm clip needs the norm of the clamped gradient)")
    if args.opt.lower() not in SUPPORTED_OPTS:
        raise SystemExit(f"--opt {args.opt}: the fused optimizer kernel (csrc/optim.hip) implements {' / '.join(SUPPORTED_OPTS)}; timm's other "
                         "create_optimizer_v2 choices (reference train.py:161, 583) are not built and nothing is substituted for them")
    if args.dino and args.opt.lower() != "adamw":
        raise SystemExit(f"--dino --opt {args.opt}: the DINO step is AdamW with the recipe's weight-decay schedule (paper; the frozen-last-layer and "
                         "teacher-EMA ranges are fused into that pass); pass --opt adamw")
    if args.sched.lower() not in SUPPORTED_SCHEDS:
        raise SystemExit(f"--sched {args.sched}: this build steps the learning rate by {' / '.join(SUPPORTED_SCHEDS)} (+ linear warm-up); timm's other "
                         "create_scheduler_v2 choices (reference train.py:180, 881-887) are not built and a constant rate is not substituted")
    if args.in_chans not in (None, 3):
        raise SystemExit(f"--in-chans {args.in_chans}: the patch-embedding kernel reads 3-channel NHWC uint8 tiles")
    img = None
    if args.input_size:
        c, h, w = args.input_size
        if c != 3 or h != w or h % 16:
            raise SystemExit(f"--input-size {args.input_size}: need 3 x S x S with S a multiple of 16")
        img = h
    if args.amp and args.amp_dtype.lower() not in ("float16", "fp16", "bfloat16", "bf16"):
        raise SystemExit(f"--amp-dtype {args.amp_dtype}: 'float16' (the reference default, train.py:332) or 'bfloat16'")
    if amp_is_f16(args):
        # the same kernels built with IEEE half as the 16-bit format (libgipvit_hip_f16.so) + torch's GradScaler on the device
        if args.opt.lower() == "lamb" or args.clip_mode == "agc":
            raise SystemExit("--amp --amp-dtype float16: dynamic loss scaling is built for adamw / adam / sgd with --clip-mode norm | value; "
                             "run --opt lamb / --clip-mode agc with --amp-dtype bfloat16")
    elif not args.amp and args.precision != "fp32":
        log("no --amp: the reference would compute in fp32 (train.py:452-465); this build's GEMMs take bf16 operands with f32 "
            "accumulation, f32 residual stream and f32 master weights (the --amp --amp-dtype bfloat16 arithmetic) unless "
            "--precision fp32 is given")
    if args.amp and args.precision == "fp32":
        raise SystemExit("--amp asks for mixed precision, --precision fp32 for f32 operands throughout: pick one")
    if args.view_augment and not (args.dino and args.random_crops):
        raise SystemExit("--view-augment augments the crops that --dino --random-crops cuts on the device: pass both")
    if args.supervised and args.dino:
        raise SystemExit("--supervised (fine-tune with labels, train.py:715-717) and --dino (self-supervised) exclude each other")
    return img


def amp_is_f16(args) -> bool:
    """``--amp`` with the reference's default ``--amp-dtype float16`` (train.py:332, 452-465)."""
    return bool(args.amp) and args.amp_dtype.lower() in ("float16", "fp16")


class Meter:
    def __init__(self):
        self.val = self.sum = self.n = 0.0

    def update(self, v, k=1):
        self.val, self.sum, self.n = v, self.sum + v * k, self.n + k

    @property
    def avg(self):
        return self.sum / max(self.n, 1)


def teacher_temp_at(args, epoch: int) -> float:
    """DINO's teacher-temperature schedule: linear warm-up from --warmup-teacher-temp to --teacher-temp over the first
    --warmup-teacher-temp-epochs epochs (np.linspace over those epochs), --teacher-temp afterwards -- and from step 0
    when there is no warm-up."""
    n = args.warmup_teacher_temp_epochs
    if n <= 0 or epoch >= n:
        return args.teacher_temp
    if n == 1:
        return args.warmup_teacher_temp
    return args.warmup_teacher_temp + (args.teacher_temp - args.warmup_teacher_temp) * epoch / (n - 1)


def main(argv=None):
    args, args_text = parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    if not torch.cuda.is_available() or not str(args.device).startswith("cuda"):
        raise SystemExit("train.py: an MI355X is required (device=%s, cuda available=%s); the HIP hot path has no CPU fallback"
                         % (args.device, torch.cuda.is_available()))
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", args.local_rank))
    primary = rank == 0
    img_from_input_size = check_supported(args, _logger.warning if primary else (lambda m: None))
    if amp_is_f16(args):
        # one process computes in one 16-bit format: pick the float16 build of the library before it is first loaded
        assert "gipvit._lib" not in sys.modules, "the library was loaded before the --amp-dtype was known"
        os.environ["GIPVIT_ACT_FORMAT"] = "f16"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    reducer = None
    if world > 1:
        import torch.distributed as dist
        from gipvit.dist import RcclReducer, comm_setup
        comm = comm_setup(world)                               # CUs left to RCCL's channels; the library sizes its launches for the rest
        dist.init_process_group("nccl", device_id=dev)         # RCCL
        reducer = RcclReducer()
        if primary:
            _logger.info("data parallel over %d ranks (RCCL): launches sized for %d CUs (GIPVIT_COMM_CUS=%d)", world, comm["cu_budget"], comm["comm_cus"])
    torch.manual_seed(args.seed + rank)                        # utils.random_seed(seed, rank), train.py:467

    ignored = [e["flags"][-1] for e in REFERENCE_FLAGS if not e["used"] and e["flags"][-1].startswith("-")
               and getattr(args, flag_dest(e), None) not in (e.get("default"), None, False)]
    if ignored and primary:
        _logger.warning("flags accepted for CLI compatibility but outside this build's hot path: %s", " ".join(ignored))
    run = None
    if args.log_wandb and primary:                             # train.py:447-450 (optional here; credentials from the environment only)
        try:
            import wandb
            run = wandb.init(project=args.experiment or "gipvit", group=args.group or None, config=vars(args))
        except Exception as ex:                                # not installed / offline
            _logger.warning("--log-wandb: wandb unavailable (%s); metrics go to the log and summary.csv only", ex)

    from gipvit import models as M, sched as S, data as D, transformations as T
    from gipvit.engine import ARCHS, DinoEngine, SupervisedEngine, FeatureExtractor, Weights
    from gipvit.checkpoint import CheckpointSaver, load_checkpoint_file
    from gipvit.validate import validate
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
    eps = args.opt_eps if args.opt_eps is not None else (1e-6 if opt == "lamb" else 1e-8)       # timm defaults: Lamb 1e-6, Adam(W) 1e-8

    # ---- data (batch dict contract of the reference: 'Data', 'Target'; inference: Infer_Dataset's dict)
    synthetic = args.dataset in ("", "synthetic")
    transform = T.define_transformations(args.transform_type if not synthetic else "none", True, tile, args.c_param, "Ron")
    # the named recipes run on the DEVICE once the tiles are in HBM (gv_augment); the reader threads then deliver raw tiles
    augmenter = None
    if getattr(transform, "device_recipe", None):
        from gipvit.augment import TileAugmenter
        augmenter = TileAugmenter(transform.device_recipe, tile, transform.color_param, mean, std, seed=args.seed + 31 * rank, device=dev)
        transform = None
    inf_loader = None
    if synthetic:
        source = D.SyntheticTiles(B, tile, args.batches_per_epoch, args.num_classes or 2, seed=args.seed + rank)
        if not args.no_validate and (not args.dino or args.extract_features):
            inf_loader = D.SyntheticSlides(args.synthetic_slides, min(args.num_tiles, 2 * B), tile, args.tiles_per_iter, seed=args.seed + 99)
    elif args.dataset.startswith("tiles:") or args.data_dir:
        root = args.dataset[6:] if args.dataset.startswith("tiles:") else args.data_dir
        slides = D.scan_slides(root, args.target)
        # the evaluation selection is needed only where an inference loader (or --supervised's re-split) will use it; --test_fold -1
        # is the reference's documented "no validation" setting (train.py:367, datasets.py:284-287 give folds = []): such a run
        # trains on every fold but 'test' / 'val' and skips validate() instead of failing on the empty selection
        want_eval = not args.no_validate and (not args.dino or args.extract_features)
        no_val_fold = args.test_fold in (-1, "-1")
        train_slides = D.select_fold(slides, args.test_fold, True)
        if args.supervised or (want_eval and not no_val_fold):
            eval_slides = D.select_fold(slides, args.test_fold, False)       # raises when evaluation was asked for and no slide is in the fold
        else:
            eval_slides = []
            if want_eval and primary:
                _logger.info("--test_fold -1: no validation fold; validate() is skipped (reference train.py:367)")
            want_eval = False
        if args.supervised:
            # train.py:715-717: --supervised re-splits the TEST-fold set 80 / 20 into train / eval
            import numpy as np
            perm = np.random.default_rng(args.seed).permutation(len(eval_slides))
            k = max(1, int(len(eval_slides) * 0.8))
            train_slides, eval_slides = [eval_slides[i] for i in perm[:k]], [eval_slides[i] for i in perm[k:]] or eval_slides
        source = D.TileFolder(root, B, transform, rank, world, args.seed, tile, n_tiles=args.n_patches_train, workers=args.workers,
                              slides=train_slides)
        if want_eval:
            inf_loader = D.InferTiles(root, tile, args.tiles_per_iter, args.num_tiles, seed=args.seed, dataset_name=args.dataset,
                                      workers=args.workers, slides=eval_slides)
    else:
        raise SystemExit(f"--dataset {args.dataset}: whole-slide datasets need openslide and are outside this build "
                         "(SURVEY section 2 #15); use 'synthetic' or 'tiles:<dir>' with pre-extracted tile_<i>.data files")
    updates_per_epoch = len(source)

    # ---- model + engine
    ema_decay = args.model_ema_decay if args.model_ema else None                 # train.py:615-622
    if args.dino:
        if args.model_ema and primary:
            _logger.warning("--model-ema with --dino: the DINO teacher IS the EMA model (momentum schedule --momentum-teacher); flag has no further effect")
        img = args.global_crop_size
        eng = DinoEngine(arch=arch, img_size=img, out_dim=args.out_dim, batch=B, tile=tile, n_local=args.local_crops_number,
                         gsize=args.global_crop_size, lsize=args.local_crop_size, lr=lr, weight_decay=args.weight_decay, betas=betas, eps=eps,
                         momentum_teacher=args.momentum_teacher, teacher_temp=args.teacher_temp, clip_grad=args.clip_grad or 0.0,
                         mean=mean, std=std, device=dev, reducer=reducer, precision=args.precision, clip_mode=args.clip_mode)
        bb = (M.load_encoder_checkpoint(args.initial_checkpoint, arch, img) if args.initial_checkpoint
              else M.init_vit_state(arch, img, 0, seed=args.seed))
        eng.load_state(bb, M.init_dino_head_state(eng.D, args.out_dim, seed=args.seed + 1))
        nc = 0
    else:
        img = args.img_size or img_from_input_size or tile   # the reference patches timm's default cfg to 256 (train_instruct.txt:9-13)
        nc = args.num_classes or 2
        eng = SupervisedEngine(arch=arch, img_size=img, num_classes=nc, batch=B, lr=lr, weight_decay=args.weight_decay, betas=betas, eps=eps,
                               smoothing=args.smoothing, clip_grad=args.clip_grad or 0.0, mean=mean, std=std, device=dev, reducer=reducer,
                               opt=opt, momentum=args.momentum,
                               train_backbone=not args.no_grad, model_ema_decay=ema_decay, precision=args.precision, clip_mode=args.clip_mode)
        st = (M.load_encoder_checkpoint(args.initial_checkpoint, arch, img, nc) if args.initial_checkpoint
              else M.init_vit_state(arch, img, nc, seed=args.seed))
        eng.load_state(st)
    start_epoch = args.start_epoch or 0
    if args.resume:
        ck = load_checkpoint_file(args.resume)
        strip = lambda d: {k[7:] if k.startswith("module.") else k: v for k, v in d.items()}
        sd = strip(ck["state_dict"])
        sub = lambda d, pre: {k[len(pre):]: v for k, v in d.items() if k.startswith(pre)}
        if args.dino:
            eng.load_state(sub(sd, "backbone."), sub(sd, "head."))
            if "state_dict_ema" in ck:        # the teacher + centre: without them the run would restart the EMA from the student
                te = strip(ck["state_dict_ema"])
                eng.load_teacher_state(sub(te, "backbone."), sub(te, "head."), ck.get("dino_center"))
            elif primary:
                _logger.warning("--resume: %s holds no teacher ('state_dict_ema'); the teacher restarts as a copy of the student", args.resume)
        else:
            eng.load_state(sd, strip(ck["state_dict_ema"]) if (ema_decay is not None and "state_dict_ema" in ck) else None)
        if "optimizer" in ck and not args.no_resume_opt:
            eng.arena.m.copy_(ck["optimizer"]["exp_avg"]); eng.arena.v.copy_(ck["optimizer"]["exp_avg_sq"]); eng.t = int(ck["optimizer"]["step"])
        if eng.scaler is not None and isinstance(ck.get("amp_scaler"), dict):      # timm resume_checkpoint(..., loss_scaler=...)
            # (a GradScaler state written by torch itself has no count of applied steps: the optimizer's step count stands in)
            eng.scaler.load_state_dict(dict({"applied_steps": eng.t}, **ck["amp_scaler"]))
        start_epoch = args.start_epoch if args.start_epoch is not None else ck.get("epoch", -1) + 1
    if hasattr(source, "epoch"):
        source.epoch = start_epoch                 # the synthetic source seeds every epoch: a resumed run sees epoch k's tiles
    if primary:
        n_params = sum(int(torch.tensor(s).prod()) for s in eng.arena.specs.values())
        _logger.info(f"Model {args.model} ({arch}) created, param count:{n_params}")

    # ---- forward-only runners for validation / feature extraction: they evaluate the engine's LIVE weights
    eval_B = min(256, max(B, 32))
    runner = runner_ema = None
    if inf_loader is not None:
        if args.dino:       # features come from the teacher backbone (the model DINO users evaluate)
            runner = FeatureExtractor(arch, tile, eval_B, 0, mean, std, dev, weights=Weights(eng.arena, "backbone.", teacher=True, fp32=args.precision == "fp32"))
            if tile != img:
                raise SystemExit(f"--extract_features with --dino: tile size {tile} must equal the teacher's image size {img}")
        else:
            runner = FeatureExtractor(arch, img, eval_B, nc, mean, std, dev, weights=eng.W)
            if ema_decay is not None:
                runner_ema = FeatureExtractor(arch, img, eval_B, nc, mean, std, dev, weights=eng.Wema)

    # ---- output dir, args.yaml, saver (train.py:854-879)
    eval_metric = args.eval_metric                              # train.py:849
    decreasing = eval_metric == "loss"                          # train.py:866
    saver = output_dir = None
    if primary:
        exp = args.experiment or "-".join([time.strftime("%Y%m%d-%H%M%S"), args.model.replace("/", "_"), str(img)])
        output_dir = os.path.join(args.output or "./output/train", exp, args.subexperiment or "")
        os.makedirs(output_dir, exist_ok=True)
        saver = CheckpointSaver(output_dir, args.model, vars(args), decreasing=decreasing or inf_loader is None or args.dino,
                                max_history=args.checkpoint_hist)
        with open(os.path.join(output_dir, "args.yaml"), "w") as f:
            f.write(args_text)
    schedule = S.LrSchedule(lr, args.sched, args.epochs, args.warmup_epochs, args.warmup_lr, args.min_lr, updates_per_epoch,
                            args.decay_epochs, args.decay_rate, on_updates=args.sched_on_updates or args.dino)
    total_updates = args.epochs * updates_per_epoch
    sampler = None
    if args.dino and args.random_crops:       # DINO recipe: random-resized crops + flips, cut on the device
        from gipvit.multicrop import MultiCropSampler
        sampler = MultiCropSampler(B, tile, 2, args.local_crops_number, tuple(args.global_crops_scale), tuple(args.local_crops_scale),
                                   seed=args.seed + rank)
    view_sampler = None
    if args.view_augment:
        from gipvit.multicrop import ViewAugmentSampler
        view_sampler = ViewAugmentSampler(B, 2, args.local_crops_number, jitter_p=args.color_jitter_p, gray_p=args.gray_p, blur_p=tuple(args.blur_p),
                                          solar_p=(0.0, args.solarize_p, 0.0), seed=args.seed + 53 * rank + 5)

    def extra_state():
        ex = {}
        if args.dino:       # teacher (the EMA model) + centre: a DINO run cannot be resumed (or evaluated) without them
            ex["state_dict_ema"] = eng.arena.state_dict(eng.arena.t)
            ex["dino_center"] = eng.center.detach().cpu()
        elif ema_decay is not None:
            ex["state_dict_ema"] = eng.state_dict(ema=True)     # timm CheckpointSaver key
        if drop_sampler is not None:
            ex["drop_path_rng"] = drop_sampler.state_dict()     # a resumed run continues the mask stream
        # ... and the other host-side draw streams (--drop step seeds, random-resized-crop boxes, view augmentation), as JSON
        # strings of numpy's bit-generator state: plain str, loads with weights_only=True
        streams = {"drop": drop_rng, "crops": getattr(sampler, "rng", None), "views": getattr(view_sampler, "rng", None)}
        ex["host_rng"] = {k: json.dumps(g.bit_generator.state) for k, g in streams.items() if g is not None}
        if eng.scaler is not None:
            ex["amp_scaler"] = eng.scaler.state_dict()          # timm CheckpointSaver(amp_scaler=loss_scaler) key, train.py:585-602
        return ex

    # the step's critical path runs on a high-priority stream; the engine's side stream (teacher forward,
    # weight-gradient GEMMs) fills the CU slots it leaves
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.Stream(dev, priority=-1))
    # tiles reach HBM one batch ahead of the step: pinned staging + a copy stream (replaces pin_memory workers, train.py:732)
    loader = D.DevicePrefetcher(source, dev, (B, tile, tile, 3), augmenter)
    # --drop-path (train.py:287-288): stochastic depth on the training passes (DINO: the student's; every crop of every tile is
    # its own sample); validation runs on its own forward-only group and never sees it
    drop_sampler = None
    if args.drop_path:
        from gipvit.droppath import DropPathSampler
        drop_sampler = DropPathSampler(ARCHS[arch]["depth"], (eng.V * B) if args.dino else B, args.drop_path, args.seed + 17 * rank, dev)
        if args.resume and isinstance(ck.get("drop_path_rng"), dict):
            drop_sampler.load_state_dict(ck["drop_path_rng"])
    # --drop (train.py:283-284): nn.Dropout at the four sites of the encoder, a fresh 32-bit seed per step (counter-based masks)
    drop_rng = None
    if args.drop:
        import numpy as np
        drop_rng = np.random.default_rng(args.seed + 7919 * rank + 11)
    if args.resume and isinstance(ck.get("host_rng"), dict):     # continue the draw streams where the saved run left them
        for k, g in (("drop", drop_rng), ("crops", getattr(sampler, "rng", None)), ("views", getattr(view_sampler, "rng", None))):
            if g is not None and k in ck["host_rng"]:
                g.bit_generator.state = json.loads(ck["host_rng"][k])
    cur_lr = lr
    # ---- epoch loop (train.py:905-977) / step loop (988-1143)
    for epoch in range(start_epoch, args.epochs):
        batch_time, data_time, losses = Meter(), Meter(), Meter()
        probs, targets = [], []
        end = time.time()
        last_idx = updates_per_epoch - 1
        if args.dino:
            eng.train_last_layer = epoch >= args.freeze_last_layer
        for batch_idx, mb in (enumerate(loader) if not args.extract_features else ()):      # train.py:906
            data, target = mb["Data"], mb["Target"]
            fill = None
            if augmenter is not None:      # the draws came with the batch (pinned staging); the pixel work is one launch here
                data, fill = augmenter.run(data, mb["AugParams"]), mb["Fill"]
            data_time.update(time.time() - end)
            cur_lr = schedule.at(epoch, batch_idx)
            if drop_sampler is not None:
                eng.set_drop_path(drop_sampler.sample())
            if drop_rng is not None:
                eng.set_dropout(args.drop, int(drop_rng.integers(0, 1 << 32)))
            if args.dino:
                it = epoch * updates_per_epoch + batch_idx
                sch = dict(lr=cur_lr, wd=S.cosine_between(args.weight_decay, args.weight_decay_end, it, total_updates),
                           momentum_teacher=S.cosine_between(args.momentum_teacher, 1.0, it, total_updates),
                           teacher_temp=teacher_temp_at(args, epoch))
                loss_t = eng.step(data, boxes=sampler.sample(dev) if sampler is not None else None, fill=fill if sampler is None else None,
                                  views=view_sampler.sample(dev) if view_sampler is not None else None, **sch)
            else:
                loss_t = eng.step(data, target, lr=cur_lr, fill=fill)
                probs.append(eng.prob[:, 1].clone() if eng.C > 1 else eng.prob[:, 0].clone()); targets.append(target.view(-1).clone())
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
                    saver.save_recovery(epoch, batch_idx, eng.arena.state_dict(), extra=extra_state())
            end = time.time()
        # ---- end of epoch: train metrics, slide-level validation (train.py:932-955), summary.csv, checkpoint (958-973)
        train_metrics = OrderedDict(loss=losses.avg)
        if probs:
            try:
                from sklearn.metrics import roc_auc_score
                train_metrics["auc"] = float(roc_auc_score(torch.cat(targets).cpu().numpy(), torch.cat(probs).float().cpu().numpy()))
            except Exception:                       # single-class epoch etc. (the reference would raise, train.py:1054)
                train_metrics["auc"] = float("nan")
        eval_metrics = OrderedDict()
        if runner is not None:
            eval_metrics = validate(runner, inf_loader, extract_features=bool(args.extract_features), smoothing=args.smoothing,
                                    log_interval=args.log_interval, out_dir=args.features_dir, primary=primary)
            if runner_ema is not None and not args.extract_features:       # train.py:943-955: the EMA model's metrics win
                eval_metrics = validate(runner_ema, inf_loader, smoothing=args.smoothing, log_interval=args.log_interval, primary=primary,
                                        log_suffix=" (EMA)")
        if args.extract_features:
            if primary:
                _logger.info(f"*** features of {int(eval_metrics.get('slides', 0))} slides written to {args.features_dir}")
            break
        if primary:
            row = OrderedDict(epoch=epoch, **{"train_" + k: v for k, v in train_metrics.items()}, **{"eval_" + k: v for k, v in eval_metrics.items()},
                              lr=cur_lr)
            fn = os.path.join(output_dir, "summary.csv")
            new = not os.path.exists(fn)
            with open(fn, "a") as f:
                w = csv.DictWriter(f, fieldnames=row.keys())
                if new:
                    w.writeheader()
                w.writerow(row)
            if run is not None:
                run.log(dict(row))
            optim = {"exp_avg": eng.arena.m, "exp_avg_sq": eng.arena.v, "step": eng.t}
            if eval_metrics:
                if eval_metric not in eval_metrics:
                    raise SystemExit(f"--eval-metric {eval_metric}: validate() reports {list(eval_metrics)}")
                save_metric, name = eval_metrics[eval_metric], "eval " + eval_metric          # train.py:970-973
            else:
                save_metric, name = train_metrics["loss"], "train loss"
            best, best_ep = saver.save_checkpoint(epoch, eng.arena.state_dict(), optim, metric=save_metric, extra=extra_state())
            _logger.info(f"*** epoch {epoch}: " + "  ".join(f"{k} {v:.4f}" for k, v in list(train_metrics.items()) + [("eval_" + k, v) for k, v in eval_metrics.items()])
                         + f"  (best {name} {best:.4f} @ {best_ep})")
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
