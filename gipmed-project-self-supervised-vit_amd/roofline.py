"""Live roofline measurement of the dominant kernel for bench.py's JSON line.

The step is MFMA-bound as a whole (98.7 % of FLOPs are dense contractions) and its
dominant kernels are instantiations of the bf16 MFMA GEMM behind gv_linear.  After the timed
region bench.py runs a few more steps of the SAME workload with gv_linear_timing enabled: the
library brackets every GEMM launch with two HIP events on the launch stream and folds them
into one row per kernel instantiation.  The dominant kernel is the row with the largest summed
time; achieved = its algorithmic FLOPs (2*M*N*K of every launch) / its summed duration, i.e.
average FLOPs per launch / average launch duration -- the average rocprofv3 --kernel-trace
--stats reports for the same kernel name must agree (profiles/).  `traffic` is the HBM bytes
per launch of that kernel from the committed PMC passes (FETCH_SIZE doubled per the gfx950
correction + WRITE_SIZE), looked up in profiles/*_hbm_traffic_per_kernel.json."""
from __future__ import annotations

import glob
import json
import os

import torch

from . import ops

PEAK_BF16_TFLOPS = 2500.0     # MI355X_MICROARCH.md, dense bf16 MFMA


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


def dominant_kernel_roofline(step_fn, steps: int = 3, vit=None):
    """`vit`: the engine's VitRunner.  The step overlaps the weight-gradient GEMMs and the teacher
    forward with the main stream (engine.py); a kernel that shares the chip is stretched by its
    neighbours, so the roofline figure is taken from steps run with that side stream switched off
    (one kernel on the chip at a time, what `GIPVIT_DW_STREAM=0 rocprofv3 --kernel-trace --stats`
    reports); the same kernel's duration inside the overlapped step is given as `in_step`."""
    in_step = None
    if vit is not None and vit.side is not None:
        shared = {r["kernel"]: r for r in _timed_rows(step_fn, steps)}
        keep, vit.side = vit.side, None
        try:
            rows = _timed_rows(step_fn, steps)
        finally:
            vit.side = keep
        r = shared.get(rows[0]["kernel"])
        if r is not None:
            in_step = {"avg_launch_us": round(r["seconds"] / r["launches"] * 1e6, 2), "achieved": round(r["flops"] / r["seconds"] / 1e12, 1),
                       "note": "same kernel while the side stream's kernels share the chip"}
    else:
        rows = _timed_rows(step_fn, steps)
    table = [{"kernel": r["kernel"], "launches_per_step": round(r["launches"] / steps, 2), "avg_us": round(r["seconds"] / r["launches"] * 1e6, 2),
              "ms_per_step": round(r["seconds"] / steps * 1e3, 3), "tflops": round(r["flops"] / r["seconds"] / 1e12, 1)} for r in rows]
    d = rows[0]
    ach = d["flops"] / d["seconds"] / 1e12
    traffic, src = _traffic_for(d["kernel"])
    return {"bound": "mfma", "kernel": d["kernel"], "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": src,
            "launches_per_step": round(d["launches"] / steps, 2), "avg_launch_us": round(d["seconds"] / d["launches"] * 1e6, 2),
            "avg_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 3), "timed_steps": steps, "in_step": in_step, "gemm_kernels": table}
