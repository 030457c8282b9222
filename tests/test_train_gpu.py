"""GPU: the training driver end to end -- supervised (the reference's actual loss path, BASELINE config 1 shape) with
slide-level validation, --model-ema, --no-grad, --extract_features, the tiles:<dir> source behind the pinned
prefetcher, and --dino incl. checkpoint / resume of the teacher and the centre."""
import csv
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_train_supervised_and_resume(dev, tmp_path, caplog):
    sys.path.insert(0, ROOT)
    import train
    common = ["--model", "vit_tiny_patch16_224", "--dataset", "synthetic", "--num-classes", "2", "--img-size", "64", "--tile-size", "64",
              "-b", "8", "--batches-per-epoch", "6", "--opt", "adam", "--lr-base", "0.001", "--sched", "cosine", "--warmup-epochs", "1",
              "--log-interval", "2", "--output", str(tmp_path), "--experiment", "exp", "--subexperiment", "sub", "--seed", "1",
              "--synthetic-slides", "4", "--num_tiles", "12", "--tiles_per_iter", "5", "--model-ema", "--model-ema-decay", "0.9"]
    with caplog.at_level("INFO"):
        assert train.main(common + ["--epochs", "2"]) == 0
    out = tmp_path / "exp" / "sub"
    names = sorted(os.listdir(out))
    assert {"args.yaml", "summary.csv", "last.pth.tar", "model_best.pth.tar", "checkpoint-0.pth.tar", "checkpoint-1.pth.tar"} <= set(names)
    rows = list(csv.DictReader(open(out / "summary.csv")))
    assert [int(r["epoch"]) for r in rows] == [0, 1] and all(0.3 < float(r["train_loss"]) < 1.2 for r in rows)
    # slide-level validation ran every epoch (train.py:933): the reference's metrics + the AUCs
    for r in rows:
        assert 0.3 < float(r["eval_loss"]) < 1.2 and 0.0 <= float(r["eval_top1"]) <= 100.0 and float(r["eval_top5"]) == 100.0
        assert 0.0 <= float(r["eval_auc_per_patch"]) <= 1.0 and 0.0 <= float(r["eval_auc_per_slide"]) <= 1.0
    assert any(m.startswith("Train: 1 [") and "rate" not in m and "/s" in m and "LR:" in m for m in caplog.messages)
    assert any(m.startswith("Test: [") and "Acc@1" in m for m in caplog.messages) and any(m.startswith("Test (EMA): [") for m in caplog.messages)
    ck = torch.load(out / "last.pth.tar", weights_only=True)
    assert ck["epoch"] == 1 and ck["state_dict"]["pos_embed"].shape == (1, 17, 192) and ck["state_dict"]["head.weight"].shape == (2, 192)
    assert ck["optimizer"]["step"] == 12
    # --model-ema (train.py:615-622, 1080-1081): the EMA copy is saved under timm's key and trails the model
    ema, sd = ck["state_dict_ema"], ck["state_dict"]
    assert set(ema) == set(sd) and 1e-6 < _rel(ema["blocks.3.mlp.fc1.weight"], sd["blocks.3.mlp.fc1.weight"]) < 0.2
    # resume continues at epoch 2 with the optimizer state and the EMA
    assert train.main(common + ["--epochs", "3", "--resume", str(out / "last.pth.tar")]) == 0
    rows = list(csv.DictReader(open(out / "summary.csv")))
    assert [int(r["epoch"]) for r in rows] == [0, 1, 2]
    # encoder checkpoint round trip into a head-only fine-tune (--no-grad) of a 4-class head, with weight decay ON:
    # the frozen encoder must not move at all (requires_grad=False parameters never reach the optimizer, train.py:497-503)
    ft_args = common[:4] + ["--num-classes", "4"] + common[6:-3] + ["--epochs", "1", "--no-grad", "--experiment", "ft", "--weight-decay", "0.05",
                                                                     "--opt", "adamw", "--initial-checkpoint", str(out / "model_best.pth.tar")]
    assert train.main(ft_args) == 0
    ft = torch.load(tmp_path / "ft" / "sub" / "last.pth.tar", weights_only=True)["state_dict"]
    src = torch.load(out / "model_best.pth.tar", weights_only=True)["state_dict"]
    assert ft["head.weight"].shape == (4, 192)
    for k in ("blocks.3.mlp.fc1.weight", "blocks.0.attn.qkv.weight", "pos_embed", "norm.weight", "patch_embed.proj.weight", "blocks.11.mlp.fc2.bias"):
        assert torch.equal(ft[k], src[k]), k
    fresh_head = torch.load(tmp_path / "ft" / "sub" / "checkpoint-0.pth.tar", weights_only=True)["state_dict"]["head.weight"]
    assert float(fresh_head.abs().max()) > 0


def test_extract_features_writes_one_file_per_slide(dev, tmp_path):
    """--extract_features (train.py:906, 1281-1282): no training; <slide>_features.pt per slide, D columns."""
    sys.path.insert(0, ROOT)
    import train
    fd = tmp_path / "feats"
    rc = train.main(["--model", "vit_tiny_patch16_224", "--dataset", "synthetic", "--img-size", "64", "--tile-size", "64", "-b", "8", "--epochs", "1",
                     "--output", str(tmp_path), "--experiment", "fx", "--extract_features", "--synthetic-slides", "3", "--num_tiles", "11",
                     "--tiles_per_iter", "4", "--features-dir", str(fd)])
    assert rc == 0
    files = sorted(os.listdir(fd))
    assert files == [f"synthetic_{k}_features.pt" for k in range(3)]
    f0 = torch.load(fd / files[0], weights_only=True)
    assert f0.shape == (11, 192) and f0.dtype == torch.float32 and bool(torch.isfinite(f0).all()) and float(f0.std()) > 0.05
    assert not os.path.exists(tmp_path / "fx" / "summary.csv")              # nothing was trained


def test_tiles_dataset_through_pinned_prefetcher(dev, tmp_path):
    """tiles:<dir> (the reference's raw tile files) -> reader threads -> pinned staging -> copy stream -> step; the batches
    the engine sees are byte-identical to the files."""
    sys.path.insert(0, ROOT)
    import train
    from gipvit import data as D
    rng = np.random.default_rng(0)
    root = tmp_path / "tiles"
    for s in range(4):
        os.makedirs(root / f"slide{s}")
        for i in range(6):
            D.write_tile_file(str(root / f"slide{s}" / f"tile_{i}.data"), rng.integers(0, 256, (64, 64, 3), dtype=np.uint8))
    (root / "labels.csv").write_text("slide,label,fold\n" + "".join(f"slide{s},{s % 2},{1 + s // 2}\n" for s in range(4)))
    # the prefetcher alone: device batches equal the source's host batches, in order
    src = D.TileFolder(str(root), 4, None, seed=3, tile_size=64, n_tiles=4)
    ref = [b for b in D.TileFolder(str(root), 4, None, seed=3, tile_size=64, n_tiles=4)]
    got = []
    for mb in D.DevicePrefetcher(src, dev, (4, 64, 64, 3)):
        got.append((mb["Data"].cpu().clone(), mb["Target"].cpu().clone()))
    assert len(got) == len(ref) == 4
    for (d, t), r in zip(got, ref):
        assert torch.equal(d, r["Data"]) and torch.equal(t, r["Target"])
    rc = train.main(["--model", "vit_tiny_patch16_224", "--dataset", f"tiles:{root}", "--num-classes", "2", "--img-size", "64", "--tile-size", "64",
                     "-b", "4", "--epochs", "2", "--opt", "adamw", "--lr", "1e-4", "--warmup-epochs", "0", "--output", str(tmp_path), "--experiment", "t",
                     "--n_patches_train", "4", "--test_fold", "2", "--transform_type", "none", "--workers", "3", "--num_tiles", "5", "--tiles_per_iter", "3",
                     "--eval-metric", "loss", "--log-interval", "1"])
    assert rc == 0
    rows = list(csv.DictReader(open(tmp_path / "t" / "summary.csv")))
    assert len(rows) == 2 and all(np.isfinite(float(r["eval_loss"])) for r in rows)
    # a named reference recipe (colour jitter, noise, flips / rotations, zoom, cutout after Normalize): augmented on the device
    rc = train.main(["--model", "vit_tiny_patch16_224", "--dataset", f"tiles:{root}", "--num-classes", "2", "--img-size", "64", "--tile-size", "64",
                     "-b", "4", "--epochs", "1", "--opt", "adamw", "--lr", "1e-4", "--warmup-epochs", "0", "--output", str(tmp_path), "--experiment", "aug",
                     "--n_patches_train", "4", "--test_fold", "2", "--transform_type", "pcbnfrsc", "--c_param", "0.2", "--no-validate", "--log-interval", "1"])
    assert rc == 0
    r = list(csv.DictReader(open(tmp_path / "aug" / "summary.csv")))
    assert len(r) == 1 and 0.3 < float(r[0]["train_loss"]) < 1.2
    # --test_fold -1 is the reference's "no validation" setting (train.py:367; datasets.py:284-287 give folds = []): the driver
    # trains on every fold (no 'test' / 'val' rows here) and skips validate() instead of failing on the empty evaluation selection
    rc = train.main(["--model", "vit_tiny_patch16_224", "--dataset", f"tiles:{root}", "--num-classes", "2", "--img-size", "64", "--tile-size", "64",
                     "-b", "4", "--epochs", "1", "--opt", "adamw", "--lr", "1e-4", "--warmup-epochs", "0", "--output", str(tmp_path), "--experiment", "nofold",
                     "--n_patches_train", "4", "--test_fold", "-1", "--transform_type", "none", "--log-interval", "1"])
    assert rc == 0
    r = list(csv.DictReader(open(tmp_path / "nofold" / "summary.csv")))
    assert len(r) == 1 and np.isfinite(float(r[0]["train_loss"])) and not r[0].get("eval_loss")
    # ... but a run that asks for evaluation slides of a fold nobody is in still fails loudly
    with pytest.raises(ValueError, match="no evaluation slides"):
        train.main(["--model", "vit_tiny_patch16_224", "--dataset", f"tiles:{root}", "--num-classes", "2", "--img-size", "64", "--tile-size", "64",
                    "-b", "4", "--epochs", "1", "--opt", "adamw", "--output", str(tmp_path), "--experiment", "badfold", "--test_fold", "7"])


def test_train_dino_checkpoint_and_resume(dev, tmp_path):
    """DINO checkpoints carry the teacher ('state_dict_ema') and the centre; a run resumed from epoch 0's file continues
    exactly like the uninterrupted run (teacher, centre, schedules, frozen last layer in epoch 0)."""
    sys.path.insert(0, ROOT)
    import train
    base = ["--dino", "--model", "vit_tiny", "--dataset", "synthetic", "-b", "2", "--out-dim", "1024", "--batches-per-epoch", "3", "--lr", "1e-4",
            "--weight-decay", "0.04", "--clip-grad", "3.0", "--warmup-epochs", "1", "--freeze-last-layer", "1", "--log-interval", "1",
            "--output", str(tmp_path), "--seed", "7", "--warmup-teacher-temp-epochs", "2", "--teacher-temp", "0.07"]
    assert train.main(base + ["--epochs", "2", "--experiment", "full"]) == 0
    rows = list(csv.DictReader(open(tmp_path / "full" / "summary.csv")))
    assert len(rows) == 2 and all(5.0 < float(r["train_loss"]) < 8.0 for r in rows)
    ck0 = torch.load(tmp_path / "full" / "checkpoint-0.pth.tar", weights_only=True)
    ck1 = torch.load(tmp_path / "full" / "last.pth.tar", weights_only=True)
    sd, te = ck1["state_dict"], ck1["state_dict_ema"]
    assert "backbone.blocks.0.attn.qkv.weight" in sd and "head.last_layer.weight_v" in sd and set(te) == set(sd)
    assert ck1["dino_center"].shape == (1024,) and float(ck1["dino_center"].abs().max()) > 0
    assert 1e-7 < _rel(te["backbone.blocks.5.mlp.fc1.weight"], sd["backbone.blocks.5.mlp.fc1.weight"]) < 1e-1     # the teacher trails the student
    # epoch 0 froze the last layer: the student's copy is exactly its initial value there, the other head layers moved
    from gipvit.models import init_dino_head_state
    h0 = init_dino_head_state(192, 1024, seed=8)
    assert torch.equal(ck0["state_dict"]["head.last_layer.weight_v"], h0["last_layer.weight_v"])
    assert not torch.equal(ck0["state_dict"]["head.mlp.0.weight"], h0["mlp.0.weight"])
    assert not torch.equal(sd["head.last_layer.weight_v"], h0["last_layer.weight_v"])          # thawed in epoch 1
    # resume from epoch 0 and finish epoch 1: same teacher / centre / student as the uninterrupted run (atomics: not bitwise)
    assert train.main(base + ["--epochs", "2", "--experiment", "res", "--resume", str(tmp_path / "full" / "checkpoint-0.pth.tar")]) == 0
    r1 = torch.load(tmp_path / "res" / "last.pth.tar", weights_only=True)
    assert r1["epoch"] == 1 and r1["optimizer"]["step"] == 6
    for k in ("backbone.blocks.0.attn.qkv.weight", "backbone.blocks.11.mlp.fc2.weight", "head.mlp.2.weight", "head.last_layer.weight_v", "backbone.pos_embed"):
        assert _rel(r1["state_dict_ema"][k], te[k]) < 1e-5, k
        assert _rel(r1["state_dict"][k], sd[k]) < 2e-3, k
    assert _rel(r1["dino_center"], ck1["dino_center"]) < 1e-3


@pytest.mark.gpu
def test_train_dino_drop_path(dev, tmp_path):
    """--drop-path through the driver (DINO's own default is 0.1): the student trains with per-crop stochastic depth, the run
    is reproducible from its seed and differs from the run without it."""
    sys.path.insert(0, ROOT)
    import train
    base = ["--dino", "--model", "vit_small", "--dataset", "synthetic", "-b", "2", "--out-dim", "1024", "--batches-per-epoch", "3", "--lr", "1e-4",
            "--epochs", "1", "--log-interval", "1", "--output", str(tmp_path), "--seed", "7", "--no-validate"]
    for name, extra in (("dp_a", ["--drop-path", "0.3"]), ("dp_b", ["--drop-path", "0.3"]), ("plain", [])):
        assert train.main(base + ["--experiment", name] + extra) == 0
    w = {n: torch.load(tmp_path / n / "last.pth.tar", weights_only=True)["state_dict"]["backbone.blocks.11.mlp.fc2.weight"] for n in ("dp_a", "dp_b", "plain")}
    assert _rel(w["dp_a"], w["dp_b"]) < 1e-4                   # same seed, same draws (atomics: not bitwise)
    assert _rel(w["dp_a"], w["plain"]) > 1e-4
    rows = list(csv.DictReader(open(tmp_path / "dp_a" / "summary.csv")))
    assert 5.0 < float(rows[0]["train_loss"]) < 8.0


@pytest.mark.gpu
def test_train_opt_values_select_their_own_arithmetic(dev, tmp_path):
    """Every accepted --opt value (and --clip-mode value) through the driver: sgd / adam / adamw / lamb runs from one seed end at
    different weights (nothing is mapped onto a neighbour), lamb resumes with its moments, an unsupported value exits before any GPU work."""
    sys.path.insert(0, ROOT)
    import train
    base = ["--model", "vit_tiny", "--dataset", "synthetic", "--num-classes", "2", "--img-size", "64", "--tile-size", "64", "-b", "8",
            "--batches-per-epoch", "4", "--epochs", "1", "--lr", "1e-3", "--warmup-epochs", "0", "--weight-decay", "0.05", "--log-interval", "2",
            "--output", str(tmp_path), "--seed", "3", "--no-validate", "--clip-grad", "0.5"]
    ends = {}
    for name, extra in (("sgd", ["--opt", "sgd"]), ("adam", ["--opt", "adam"]), ("adamw", ["--opt", "adamw"]), ("lamb", ["--opt", "lamb"]),
                        ("adamw_val", ["--opt", "adamw", "--clip-mode", "value"]), ("adamw_drop", ["--opt", "adamw", "--drop", "0.1"]), ("adamw_agc", ["--opt", "adamw", "--clip-mode", "agc", "--clip-grad", "0.01"])):
        assert train.main(base + ["--experiment", name] + extra) == 0
        ck = torch.load(tmp_path / name / "last.pth.tar", weights_only=True)
        ends[name] = ck["state_dict"]["blocks.5.mlp.fc1.weight"]
        assert torch.isfinite(ends[name]).all() and ck["optimizer"]["step"] == 4
    names = list(ends)
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            assert _rel(ends[names[i]], ends[names[j]]) > 1e-5, (names[i], names[j])
    assert train.main(base + ["--experiment", "lamb", "--opt", "lamb", "--epochs", "2", "--resume", str(tmp_path / "lamb" / "last.pth.tar")]) == 0
    with pytest.raises(SystemExit):
        train.main(base + ["--experiment", "bad", "--opt", "nadam"])


@pytest.mark.gpu
def test_train_dino_view_augment(dev, tmp_path):
    """--random-crops --view-augment through the driver: every crop is cut and augmented (colour jitter / grayscale / blur /
    solarise with its own draws) in one device pass; the run trains, is reproducible from its seed and differs from the
    geometry-only views.  --view-augment without --random-crops is refused."""
    sys.path.insert(0, ROOT)
    import train
    base = ["--dino", "--model", "vit_tiny", "--dataset", "synthetic", "-b", "2", "--out-dim", "1024", "--batches-per-epoch", "3", "--lr", "1e-4",
            "--epochs", "1", "--log-interval", "1", "--output", str(tmp_path), "--seed", "7", "--no-validate", "--random-crops"]
    for name, extra in (("va_a", ["--view-augment"]), ("va_b", ["--view-augment"]), ("geo", [])):
        assert train.main(base + ["--experiment", name] + extra) == 0
    w = {n: torch.load(tmp_path / n / "last.pth.tar", weights_only=True)["state_dict"]["backbone.blocks.0.attn.qkv.weight"] for n in ("va_a", "va_b", "geo")}
    assert _rel(w["va_a"], w["va_b"]) < 1e-4 and _rel(w["va_a"], w["geo"]) > 1e-5
    rows = list(csv.DictReader(open(tmp_path / "va_a" / "summary.csv")))
    assert 5.0 < float(rows[0]["train_loss"]) < 8.0
    with pytest.raises(SystemExit):
        train.main([a for a in base if a != "--random-crops"] + ["--view-augment", "--experiment", "bad"])


def test_train_precision_fp32(dev, tmp_path):
    """--precision fp32 through the driver (the reference's arithmetic without --amp): training, the EMA copy and the
    per-epoch slide validation all run on the f32 kernels; the run stays close to the bf16 run of the same seed (same
    data, same draws) without being identical to it."""
    sys.path.insert(0, ROOT)
    import train
    common = ["--model", "vit_tiny_patch16_224", "--dataset", "synthetic", "--num-classes", "2", "--img-size", "64", "--tile-size", "64",
              "-b", "8", "--batches-per-epoch", "4", "--opt", "adamw", "--lr", "1e-4", "--sched", "cosine", "--epochs", "1",
              "--log-interval", "2", "--output", str(tmp_path), "--seed", "1", "--synthetic-slides", "4", "--num_tiles", "12",
              "--tiles_per_iter", "5", "--model-ema", "--model-ema-decay", "0.9"]
    assert train.main(common + ["--experiment", "f32", "--precision", "fp32"]) == 0
    assert train.main(common + ["--experiment", "b16"]) == 0
    rows = {n: list(csv.DictReader(open(tmp_path / n / "summary.csv")))[0] for n in ("f32", "b16")}
    assert abs(float(rows["f32"]["train_loss"]) - float(rows["b16"]["train_loss"])) < 5e-3
    assert abs(float(rows["f32"]["eval_loss"]) - float(rows["b16"]["eval_loss"])) < 5e-3
    w = {n: torch.load(tmp_path / n / "last.pth.tar", weights_only=True)["state_dict"]["blocks.11.mlp.fc2.weight"] for n in ("f32", "b16")}
    assert 0 < _rel(w["f32"], w["b16"]) < 2e-2
