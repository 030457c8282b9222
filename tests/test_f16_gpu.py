"""--amp --amp-dtype float16 (the reference's default AMP arithmetic, train.py:332, 452-465, 585-602): the float16 build of the
library + the device-side GradScaler.  One process computes in one 16-bit format, so these checks run in processes of their own
(GIPVIT_ACT_FORMAT=f16 / the driver's --amp), one at a time."""
import csv
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None, timeout=900):
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


@pytest.mark.gpu
def test_f16_library_parity_and_loss_scaler(dev):
    """tests/f16_worker.py: supervised + DINO (ViT-T, the fused ViT-S path, ViT-B) forward / backward against the oracle at the
    16-bit gates, three optimizer steps against the oracle's AdamW, and the GradScaler arithmetic (skip, back-off, growth, Adam's
    count of applied steps, checkpoint state) exactly."""
    out = _run([sys.executable, os.path.join(ROOT, "tests", "f16_worker.py")], env={"GIPVIT_ACT_FORMAT": "f16"})
    print(out)
    assert out.strip().endswith("F16 OK")


@pytest.mark.gpu
def test_train_amp_float16(dev, tmp_path):
    """`--amp` through the driver (default --amp-dtype float16): the run trains under loss scaling, the checkpoint carries timm's
    'amp_scaler' entry, a resumed run picks the scale up, and the result stays close to the bfloat16 run of the same seed."""
    common = [sys.executable, os.path.join(ROOT, "train.py"), "--model", "vit_tiny_patch16_224", "--dataset", "synthetic", "--num-classes", "2",
              "--img-size", "64", "--tile-size", "64", "-b", "8", "--batches-per-epoch", "6", "--opt", "adamw", "--lr", "1e-4", "--sched", "cosine",
              "--log-interval", "2", "--output", str(tmp_path), "--seed", "1", "--synthetic-slides", "4", "--num_tiles", "12", "--tiles_per_iter", "5",
              "--clip-grad", "1.0"]
    _run(common + ["--epochs", "1", "--experiment", "h16", "--amp"])
    _run(common + ["--epochs", "1", "--experiment", "b16", "--amp", "--amp-dtype", "bfloat16"])
    ck = torch.load(tmp_path / "h16" / "last.pth.tar", weights_only=True)
    sc = ck["amp_scaler"]
    assert sc["scale"] > 0 and sc["applied_steps"] + sc["skipped_steps"] == 6 and sc["growth_interval"] == 2000
    assert "amp_scaler" not in torch.load(tmp_path / "b16" / "last.pth.tar", weights_only=True)
    rows = {n: list(csv.DictReader(open(tmp_path / n / "summary.csv")))[0] for n in ("h16", "b16")}
    assert abs(float(rows["h16"]["train_loss"]) - float(rows["b16"]["train_loss"])) < 5e-3
    assert abs(float(rows["h16"]["eval_loss"]) - float(rows["b16"]["eval_loss"])) < 5e-3
    out = _run(common + ["--epochs", "2", "--experiment", "h16r", "--amp", "--resume", str(tmp_path / "h16" / "last.pth.tar")])
    ck2 = torch.load(tmp_path / "h16r" / "last.pth.tar", weights_only=True)
    assert ck2["amp_scaler"]["applied_steps"] + ck2["amp_scaler"]["skipped_steps"] == 12 and ck2["epoch"] == 1
    # the self-supervised step under the same flag: student + teacher + centre train, the scaler state rides in the checkpoint
    dino = [sys.executable, os.path.join(ROOT, "train.py"), "--dino", "--model", "vit_tiny", "--dataset", "synthetic", "-b", "2", "--out-dim", "1024",
            "--batches-per-epoch", "4", "--lr", "1e-4", "--epochs", "1", "--log-interval", "1", "--output", str(tmp_path), "--seed", "7", "--no-validate",
            "--clip-grad", "3.0", "--amp", "--experiment", "dino16"]
    _run(dino)
    ckd = torch.load(tmp_path / "dino16" / "last.pth.tar", weights_only=True)
    assert ckd["amp_scaler"]["applied_steps"] + ckd["amp_scaler"]["skipped_steps"] == 4 and ckd["amp_scaler"]["applied_steps"] >= 1
    assert bool(torch.isfinite(ckd["state_dict"]["backbone.blocks.0.attn.qkv.weight"]).all()) and "state_dict_ema" in ckd
    row = list(csv.DictReader(open(tmp_path / "dino16" / "summary.csv")))[0]
    assert 5.0 < float(row["train_loss"]) < 8.0
