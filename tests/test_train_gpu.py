"""GPU: the training driver end to end on synthetic tiles -- supervised (the reference's actual
loss path, BASELINE config 1 shape) and --dino; log-line / summary.csv / checkpoint layout, resume."""
import csv
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_supervised_and_resume(dev, tmp_path, caplog):
    sys.path.insert(0, ROOT)
    import train
    common = ["--model", "vit_tiny_patch16_224", "--dataset", "synthetic", "--num-classes", "2", "--img-size", "64", "--tile-size", "64",
              "-b", "8", "--batches-per-epoch", "6", "--opt", "adam", "--lr-base", "0.001", "--sched", "cosine", "--warmup-epochs", "1",
              "--log-interval", "2", "--output", str(tmp_path), "--experiment", "exp", "--subexperiment", "sub", "--seed", "1"]
    with caplog.at_level("INFO"):
        assert train.main(common + ["--epochs", "2"]) == 0
    out = tmp_path / "exp" / "sub"
    names = sorted(os.listdir(out))
    assert {"args.yaml", "summary.csv", "last.pth.tar", "model_best.pth.tar", "checkpoint-0.pth.tar", "checkpoint-1.pth.tar"} <= set(names)
    rows = list(csv.DictReader(open(out / "summary.csv")))
    assert [int(r["epoch"]) for r in rows] == [0, 1] and all(0.3 < float(r["train_loss"]) < 1.2 for r in rows)
    assert any(m.startswith("Train: 1 [") and "rate" not in m and "/s" in m and "LR:" in m for m in caplog.messages)
    ck = torch.load(out / "last.pth.tar", weights_only=True)
    assert ck["epoch"] == 1 and ck["state_dict"]["pos_embed"].shape == (1, 17, 192) and ck["state_dict"]["head.weight"].shape == (2, 192)
    assert ck["optimizer"]["step"] == 12
    # resume continues at epoch 2 with the optimizer state
    assert train.main(common + ["--epochs", "3", "--resume", str(out / "last.pth.tar")]) == 0
    rows = list(csv.DictReader(open(out / "summary.csv")))
    assert [int(r["epoch"]) for r in rows] == [0, 1, 2]
    # encoder checkpoint round trip into a head-only fine-tune (--no-grad) of a 4-class head
    assert train.main(common[:4] + ["--num-classes", "4"] + common[6:] + ["--epochs", "1", "--no-grad", "--experiment", "ft",
                      "--initial-checkpoint", str(out / "model_best.pth.tar")]) == 0
    ft = torch.load(tmp_path / "ft" / "sub" / "last.pth.tar", weights_only=True)["state_dict"]
    src = torch.load(out / "model_best.pth.tar", weights_only=True)["state_dict"]
    assert ft["head.weight"].shape == (4, 192)
    # --no-grad: backbone gradients are never produced, so Adam leaves it untouched (L2 decay is 2e-5 * lr ~ 0)
    assert float((ft["blocks.3.mlp.fc1.weight"] - src["blocks.3.mlp.fc1.weight"]).abs().max()) < 1e-3   # only lr * wd * w decay
    assert float((ft["head.weight"] - 0).abs().max()) > 0


def test_train_dino(dev, tmp_path):
    sys.path.insert(0, ROOT)
    import train
    rc = train.main(["--dino", "--model", "vit_tiny", "--dataset", "synthetic", "-b", "2", "--out-dim", "1024", "--epochs", "2",
                     "--batches-per-epoch", "3", "--lr", "1e-4", "--weight-decay", "0.04", "--clip-grad", "3.0", "--warmup-epochs", "1",
                     "--freeze-last-layer", "1", "--log-interval", "1", "--output", str(tmp_path), "--experiment", "dino"])
    assert rc == 0
    rows = list(csv.DictReader(open(tmp_path / "dino" / "summary.csv")))
    assert len(rows) == 2 and all(5.0 < float(r["train_loss"]) < 8.0 for r in rows)
    sd = torch.load(tmp_path / "dino" / "last.pth.tar", weights_only=True)["state_dict"]
    assert "backbone.blocks.0.attn.qkv.weight" in sd and "head.last_layer.weight_v" in sd
