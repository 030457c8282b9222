"""Slide-level inference: the counterpart of the reference's ``validate()`` (train.py:1146-1345) over the
chunked per-slide iterator of ``Infer_Dataset`` (datasets.py:634-817), on the forward-only encoder
(``engine.FeatureExtractor``).

Per chunk of a slide (``tiles_per_iter`` tiles): forward -> softmax (train.py:1233) -> the training loss on the
soft-maxed output (the reference hands ``train_loss_fn`` to ``validate``, train.py:935: LabelSmoothingCE, i.e. the
double softmax of train.py:1046/1053) -> top-1 / top-5 (train.py:1250).  When a slide's last chunk arrives
(``'Is Last Batch'``), its tile scores are appended to the per-patch pool and their mean becomes the slide score
(train.py:1288-1293); AUC per patch and per slide at the end (train.py:1334-1343).  With ``extract_features`` the
slide's CLS features are written to ``<out_dir>/<slide>_features.pt`` instead (train.py:1281-1282; 384-d for
ViT-S, train.py:1203).

Deviations from the reference, all deliberate: a slide is closed when its LAST chunk arrives --
the reference also closes it whenever ``batch_idx % log_interval == 0`` (the bookkeeping sits inside the logging
branch, train.py:1279), which splits slides at arbitrary chunks; the per-slide feature file holds exactly the
slide's [n_tiles, D] float32 tensor -- the reference saves a pickled numpy array whose first row is the all-zero
row its accumulator was initialised with (train.py:1204, 1282), ``reference_layout=True`` reproduces that."""
from __future__ import annotations

import logging
import os
import time
from collections import OrderedDict
from typing import Optional

import numpy as np
import torch

from . import ops

_logger = logging.getLogger("train")


def accuracy_topk(prob: np.ndarray, target: np.ndarray, topk=(1, 5)):
    """timm.utils.accuracy (train.py:1250): percentage of rows whose target is among the k largest outputs."""
    maxk = min(max(topk), prob.shape[1])
    order = np.argsort(-prob, axis=1, kind="stable")[:, :maxk]
    hit = order == target.reshape(-1, 1)
    return [100.0 * hit[:, : min(k, maxk)].any(axis=1).mean() for k in topk]


def _auc(y: np.ndarray, score: np.ndarray) -> float:
    from sklearn.metrics import roc_auc_score            # the reference's metric (train.py:28, 1334-1338)
    try:
        return float(roc_auc_score(y, score))
    except ValueError:                                  # one class only (the reference would raise here)
        return float("nan")


def validate(runner, loader, *, extract_features: bool = False, smoothing: float = 0.1, log_interval: int = 50,
             out_dir: str = "./TCGA_500", log_suffix: str = "", primary: bool = True, reference_layout: bool = False,
             log: Optional[logging.Logger] = None) -> "OrderedDict[str, float]":
    """runner: ``engine.FeatureExtractor`` (``run(tiles) -> (features, logits)``); loader: ``data.InferTiles`` /
    ``data.SyntheticSlides``.  Returns OrderedDict(loss, top1, top5, auc_per_patch, auc_per_slide) -- the first three
    are the reference's ``metrics`` (train.py:1340), the AUCs what it sends to W&B (train.py:1342-1343)."""
    log = log or _logger
    dev = runner.dev
    C = runner.C
    if not extract_features and not C:
        raise ValueError("validate: scoring needs a classifier head (num_classes > 0); use extract_features=True for an encoder without one")
    loss_sum = n_seen = 0.0
    top1_sum = top5_sum = 0.0
    all_out, all_tgt, slide_out, slide_tgt = [], [], [], []
    cur_out, cur_tgt, cur_feat = [], [], []
    loss_buf = torch.zeros(1, dtype=torch.float32, device=dev)
    slide_num = 0
    batch_time = 0.0
    end = time.time()
    if hasattr(loader, "reset_counter"):
        loader.reset_counter()                                              # train.py:932
    if extract_features and primary:
        os.makedirs(out_dir, exist_ok=True)
    for batch_idx, mb in enumerate(loader):
        data = mb["Data"]
        if data.dim() == 5:
            data = data.squeeze(0)                                          # batch_size=1 loader, train.py:1211
        data = data.to(dev, non_blocking=True)
        n = data.shape[0]
        label = int(mb["Label"].reshape(-1)[0])
        feats, logits = runner.run(data)
        if extract_features:
            cur_feat.append(feats.cpu())
        else:
            target = torch.full((n,), 1 if label == 1 else 0, dtype=torch.int64, device=dev)      # train.py:1214-1217
            dlog = torch.empty_like(logits)
            prob = torch.empty_like(logits)
            ops.softmax_lsce(logits, target, loss_buf, dlog, prob, n, C, smoothing)                # train.py:1233 + 1249
            torch.cuda.synchronize()                                                               # train.py:1259-1260
            p = prob.cpu().numpy().astype(np.float64)
            t = target.cpu().numpy()
            a1, a5 = accuracy_topk(p, t)
            loss_sum += float(loss_buf) * n; n_seen += n
            top1_sum += a1 * n; top5_sum += a5 * n
            cur_out.append(p); cur_tgt.append(t)
        batch_time = time.time() - end
        end = time.time()
        if mb["Is Last Batch"]:
            name = mb["Slide Filename"]
            name = name[0] if isinstance(name, (list, tuple)) else name
            if extract_features:
                f = torch.cat(cur_feat)
                if primary:
                    path = os.path.join(out_dir, f"{name}_features.pt")
                    if reference_layout:
                        torch.save(np.concatenate((np.zeros((1, f.shape[1])), f.numpy().astype(np.float64)), axis=0), path)
                    else:
                        torch.save(f, path)
                cur_feat = []
            else:
                so, st = np.concatenate(cur_out), np.concatenate(cur_tgt)
                all_out.append(so); all_tgt.append(st)
                slide_out.append(so.mean(0)); slide_tgt.append(st[0])
                cur_out, cur_tgt = [], []
            slide_num += 1
        if primary and not extract_features and (mb["Is Last Batch"] or batch_idx % log_interval == 0):
            log.info("Test{}: [{:>4d}]  Time: {:.3f}  Loss: {:>7.4f} ({:>6.4f})  Acc@1: {:>7.4f} ({:>7.4f})  Acc@5: {:>7.4f} ({:>7.4f})".format(
                log_suffix, batch_idx, batch_time, float(loss_buf), loss_sum / max(n_seen, 1), a1, top1_sum / max(n_seen, 1), a5,
                top5_sum / max(n_seen, 1)))
    metrics = OrderedDict()
    if extract_features:
        metrics["slides"] = float(slide_num)
        return metrics
    metrics["loss"] = loss_sum / max(n_seen, 1)
    metrics["top1"] = top1_sum / max(n_seen, 1)
    metrics["top5"] = top5_sum / max(n_seen, 1)
    if all_out and C > 1:
        po, pt = np.concatenate(all_out), np.concatenate(all_tgt)
        metrics["auc_per_patch"] = _auc(pt, po[:, 1])
        metrics["auc_per_slide"] = _auc(np.array(slide_tgt), np.stack(slide_out)[:, 1])
    return metrics
