"""Live roofline measurement of the dominant kernel for bench.py's JSON line.

98.7 % of the step's FLOPs are dense contractions and its dominant kernels are the bf16 MFMA
GEMM-class kernels behind gv_linear / gv_linear_ln_* / gv_linear_dw_group.  After the timed region bench.py
runs a few more steps of the SAME workload with gv_linear_timing enabled: the library brackets every such
launch with two HIP events on the launch stream and folds them into one row per kernel instantiation, with
the launch's algorithmic FLOPs (2*M*N*K) and algorithmic HBM bytes (every operand read once, every output
written once).  The dominant kernel is the row with the largest summed time.  Which roof bounds it follows
from its arithmetic intensity against the machine balance (2500 TFLOP/s / 8 TB/s = 312 FLOP/B): below it the
kernel is priced against HBM (achieved = algorithmic bytes / time), above it against the dense bf16 MFMA
peak; both fractions are reported.  The average launch duration must agree with what rocprofv3 --kernel-trace
--stats reports for the same kernel name (profiles/*_step_kernel_stats.csv for the headline, *_exclusive.csv for `exclusive`).  `traffic` is the measured HBM bytes per launch of that
kernel from the committed PMC passes, looked up in profiles/*_hbm_traffic_per_kernel.json."""
from __future__ import annotations

import glob
import json
import os

import torch

from . import ops

PEAK_BF16_TFLOPS = 2500.0     # MI355X_MICROARCH.md, dense bf16 MFMA
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md, HBM3E
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, dense f32 MFMA (v_mfma_f32_16x16x4_f32): the fp32 operand mode's kernel


def _traffic_for(kernel: str):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in sorted(glob.glob(os.path.join(root, "profiles", "*_hbm_traffic_per_kernel.json")), reverse=True):
        try:
            for row in json.load(open(path)):
                if row.get("kernel") == kernel:
                    mb = row["read_MB_per_launch_x2corrected"] + row["write_MB_per_launch"]
                    return round(mb * 1e6), os.path.basename(path)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def _timed_rows(step_fn, steps):
    torch.cuda.synchronize()
    ops.linear_timing(True)
    try:
        for _ in range(steps):
            step_fn()
        torch.cuda.synchronize()
    finally:
        ops.linear_timing(False)
    rows = sorted(ops.linear_timing_read(), key=lambda r: -r["seconds"])
    if not rows:
        raise RuntimeError("gv_linear_timing recorded no GEMM launches")
    return rows


def _figures(d, steps):
    tf, gbs = d["flops"] / d["seconds"] / 1e12, d["bytes"] / d["seconds"] / 1e9
    peak_tf = PEAK_F32_MFMA_TFLOPS if "f32_kernel" in d["kernel"] else PEAK_BF16_TFLOPS
    hbm_bound = d["bytes"] > 0 and d["flops"] / d["bytes"] < peak_tf * 1e3 / PEAK_HBM_GBS
    head = ({"bound": "hbm", "kernel": d["kernel"], "achieved": round(gbs, 0), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)}
            if hbm_bound else
            {"bound": "mfma", "kernel": d["kernel"], "achieved": round(tf, 1), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(tf / peak_tf, 4)})
    return {**head, "mfma_frac": round(tf / peak_tf, 4), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4),
            "flop_per_byte": round(d["flops"] / d["bytes"], 1) if d["bytes"] > 0 else None,
            "avg_mb_per_launch": round(d["bytes"] / d["launches"] / 1e6, 2),
            "launches_per_step": round(d["launches"] / steps, 2), "avg_launch_us": round(d["seconds"] / d["launches"] * 1e6, 2),
            "avg_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 3)}


def dominant_kernel_roofline(step_fn, steps: int = 3, vit=None):
    """`vit`: the engine's VitRunner.  The headline figures are those of the TIMED configuration: the step as bench.py
    times it, with the side stream's kernels (weight gradients, the teacher's forward) sharing the chip -- a kernel that
    shares CUs is stretched by its neighbours, and that is the duration the step pays.  `exclusive` holds the same kernel
    with the side stream switched off (one kernel on the chip at a time: what `GIPVIT_DW_STREAM=0 rocprofv3 --kernel-trace
    --stats` reports, profiles/*_step_kernel_stats_exclusive.csv) -- the figure that says how good the kernel itself is."""
    rows = _timed_rows(step_fn, steps)
    d = rows[0]
    exclusive, table_alone = None, None
    if vit is not None and vit.side is not None:
        keep, vit.side = vit.side, None
        try:
            alone = {r["kernel"]: r for r in _timed_rows(step_fn, steps)}
        finally:
            vit.side = keep
        if d["kernel"] in alone:
            exclusive = _figures(alone[d["kernel"]], steps)
            exclusive["note"] = "same kernel with the side stream off: one kernel on the chip at a time"
        table_alone = {k: round(r["seconds"] / r["launches"] * 1e6, 2) for k, r in alone.items()}
    table = [{"kernel": r["kernel"], "launches_per_step": round(r["launches"] / steps, 2), "avg_us": round(r["seconds"] / r["launches"] * 1e6, 2),
              "ms_per_step": round(r["seconds"] / steps * 1e3, 3), "tflops": round(r["flops"] / r["seconds"] / 1e12, 1),
              "gbs": round(r["bytes"] / r["seconds"] / 1e9, 0),
              **({"avg_us_exclusive": table_alone.get(r["kernel"])} if table_alone is not None else {})} for r in rows]
    traffic, src = _traffic_for(d["kernel"])
    return {**_figures(d, steps), "configuration": "in_step (the timed configuration: side stream on)" if exclusive is not None else "in_step",
            "traffic": traffic, "traffic_source": src, "traffic_measured_in": "profiles",
            "timed_steps": steps, "exclusive": exclusive, "gemm_kernels": table}
