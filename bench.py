#!/usr/bin/env python3
"""bench.py -- tiles/s of one full DINO multi-crop training step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config c3|c2]

One process per GPU.  Under torchrun (RANK/LOCAL_RANK/WORLD_SIZE in the environment) this process IS one
rank; started bare with --gpus N > 1 it is only a launcher: it touches no GPU, starts N copies of itself
with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_ADDR=127.0.0.1/MASTER_PORT set (the reference's
`torchrun --nproc_per_node=N train.py` pattern, sbatch-ssl.sh:55) and relays rank 0's JSON line.
Kernels are launched eagerly on the current HIP stream.  A step = teacher forward on
the 2 global crops + student forward/backward on 2x224 + 8x96 crops of B synthetic 256-px
NHWC uint8 tiles per GPU (resident in HBM before the timed region) + DINO loss + gradient /
center all-reduce (RCCL) + AdamW + teacher EMA.  ViT-S/16, K = 65536, bf16 MFMA with f32
accumulation / master weights.  Rank 0 prints ONE JSON line (contract in DESIGN.md).
"""
from __future__ import annotations

import os as _os
# The step uses three streams with cross-stream waits (main, the side stream of engine.py, RCCL's
# communication stream).  HIP maps streams onto 4 hardware queues by default and a stream's wait
# blocks everything behind it in a shared queue: measured 20.2 vs 18.2 ms/step with RCCL in the
# picture.  Must be set before the HIP runtime initialises.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_TILE = {("vit_small", "c3"): 113.83, ("vit_small", "c2"): 73.93, ("vit_base", "c3"): 435.56}   # BASELINE.md section 2 / SURVEY 8(a)
MFMA_BF16_PEAK_TFLOPS = 2500.0                      # MI355X_MICROARCH.md: dense bf16
MFMA_F32_PEAK_TFLOPS = 157.3                        # MI355X_MICROARCH.md: dense f32 (the fp32 operand mode's MFMA)


def synth_tiles(B, size, seed, device):
    g = torch.Generator().manual_seed(seed)
    mean = torch.tensor((0.8998, 0.8253, 0.9357)) * 255.0
    std = torch.tensor((0.1125, 0.1751, 0.0787)) * 255.0
    x = torch.randn(B, size, size, 3, generator=g) * std + mean
    return x.round().clamp(0, 255).to(torch.uint8).to(device)


def cpu_baseline(arch, n_local, warmup=2, steps=5):
    """The oracle (CPU restatement, 'port') timed on this box's host cores with BASELINE.md section 3's protocol:
    B = 8 tiles per step, K = 65536, 2 warm-up steps, then 5 timed steps (about 15-25 s on the box's 16 cores)."""
    from oracle import step_oracle as so, vit_oracle as vo
    # the GPU box exposes more logical CPUs than its share (16 per GPU); oversubscribing stalls torch
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    torch.set_num_threads(cores)
    B = 8
    orc = so.DinoOracle(arch=arch, img_size=224, out_dim=65536, n_local=n_local)
    tiles = vo.synth_tiles(B, 256, seed=1234)
    for _ in range(warmup):
        orc.step(tiles)
    t0 = time.time()
    for _ in range(steps):
        orc.step(tiles)
    dt = time.time() - t0
    return {"value": round(B * steps / dt, 4), "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": f"{steps} oracle steps of B={B} tiles (2x224+{n_local}x96 crops, {arch}, K=65536) after {warmup} warm-up steps (BASELINE.md section 3 protocol)"}


def launch_children(n: int, argv, child=None, timeout=None, poll_s: float = 0.2):
    """Start n ranks of this script (or of `child`, a command list: tests use a stub) as CHILD processes with the
    torchrun environment and return (exit code, rank 0's stdout).  The parent never initialises HIP -- nothing is
    exec'ed from a process that has touched the GPU.  Every child is polled: the first rank that exits non-zero (or the
    bound `timeout`, default BENCH_LAUNCH_TIMEOUT = 1500 s) ends the job -- the remaining ranks, which would sit in a
    collective until the backend's own timeout, are killed (the exact Popen objects), that rank's stderr tail is relayed
    and its code returned (the reference's torchrun does the same for `sbatch-ssl.sh:55`)."""
    import socket
    import subprocess
    import tempfile
    import threading
    if timeout is None:
        timeout = float(os.environ.get("BENCH_LAUNCH_TIMEOUT", "1500"))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = list(child) if child is not None else [sys.executable, os.path.abspath(__file__)]
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        errs.append(tempfile.TemporaryFile("w+"))
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=errs[r], text=True))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)     # drains rank 0's pipe while all ranks are polled
    reader.start()
    deadline = time.monotonic() + timeout
    rc, failed = 0, None
    try:
        while True:
            codes = [pr.poll() for pr in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed, rc = bad[0][0], bad[0][1]
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                failed, rc = -1, 124
                break
            time.sleep(poll_s)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()               # the exact children started above, never a pattern
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
    reader.join(timeout=10)

    def tail(r, lines=30):
        errs[r].seek(0)
        return "".join(errs[r].readlines()[-lines:])
    if failed is None:
        sys.stderr.write(tail(0, 200))                      # rank 0's diagnostics, as when its stderr was inherited
    elif failed < 0:
        sys.stderr.write(f"bench.py launcher: {n} ranks did not finish within {timeout:.0f} s; killed.  rank 0 stderr tail:\n{tail(0)}")
    else:
        sys.stderr.write(f"bench.py launcher: rank {failed} exited with code {rc}; the other ranks were killed.  Its stderr tail:\n{tail(failed)}")
        rc = rc if rc > 0 else 1                            # a signal (negative Popen code) still reads as failure
    sys.stderr.flush()
    for f in errs:
        f.close()
    return rc, (out0[0] if out0 else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="tiles per GPU per step")
    ap.add_argument("--config", default="c3", choices=["c3", "c2"])
    ap.add_argument("--arch", default="vit_small")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--micro", type=int, default=1, help="micro-batches of --batch tiles per optimizer step (gradient accumulation: BASELINE "
                    "config 5 is --arch vit_base --batch 64 --micro 8 = 512 tiles per GPU and step); a parity-case run, not the headline line")
    ap.add_argument("--random-crops", action="store_true", help="cut random-resized crops on the device every step (gv_crop_resize) instead of "
                    "the fixed parity windows of SURVEY 8(d); the default (and the quoted metric) uses the fixed windows")
    ap.add_argument("--view-augment", action="store_true", help="with --random-crops: per-crop colour jitter / grayscale / blur / solarisation in the "
                    "pass that cuts the crops (gv_crop_augment)")
    ap.add_argument("--precision", default="bf16", choices=("bf16", "f16", "fp32"), help="fp32: the verification mode (every operand f32, "
                    "csrc/f32path.hip); f16: the float16 build of the library under dynamic loss scaling (train.py --amp, the reference's default "
                    "--amp-dtype); the headline number is the bf16 training path")
    ap.add_argument("--trace-loss", action="store_true", help="debug: synchronise and print the loss after every step")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        rc, out0 = launch_children(args.gpus, sys.argv[1:])
        sys.stdout.write(out0); sys.stdout.flush()
        raise SystemExit(rc)
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # BENCH_SHARE_GPU=1: rehearsal of the N-rank path on a ONE-GPU box -- every rank uses device 0.  RCCL refuses two
    # ranks on one device, so the rehearsal's transport is gloo (device tensors staged through the host); everything
    # else (launcher, shards, ranged reductions, barrier, max-over-ranks timing) is the real path.
    share = bool(os.environ.get("BENCH_SHARE_GPU"))
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    reducer = None
    # BENCH_FORCE_DIST=1: take the RCCL path even at world size 1 (rehearses process-group set-up, the
    # ranged all-reduces and the barrier on a one-GPU box; the all-reduces then run over a single rank)
    force_dist = bool(os.environ.get("BENCH_FORCE_DIST")) and "RANK" in os.environ
    rccl_ranks, dist_backend, comm = 1, None, {"comm_cus": 0, "cu_budget": 256}
    if world > 1 or force_dist:
        import torch.distributed as dist
        from gipvit.dist import RcclReducer, comm_setup, quiet_init_process_group
        backend = dist_backend = os.environ.get("BENCH_DIST_BACKEND", "gloo" if share else "nccl")      # "nccl" IS RCCL on ROCm
        comm = comm_setup(world, backend)      # CUs left to RCCL's channels + the matching launch budget (before the library loads)
        if backend == "nccl":
            quiet_init_process_group("nccl", device_id=dev)
        else:
            quiet_init_process_group(backend)
        reducer = RcclReducer()
        reducer.always = force_dist      # rehearsal: issue the coalesced ranges at world size 1 too
        rccl_ranks = dist.get_world_size() if backend == "nccl" else 0       # ranks RCCL carries (0: another transport, see dist_backend)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    if args.precision == "f16":          # one process computes in one 16-bit format: chosen before the library is first loaded
        os.environ["GIPVIT_ACT_FORMAT"] = "f16"
    from gipvit.engine import DinoEngine
    from gipvit import roofline
    n_local = 8 if args.config == "c3" else 0
    # DINO recipe (paper defaults): AdamW, lr = 5e-4 * global_batch / 256, wd 0.04, clip 3.0
    lr = 5e-4 * args.batch * world / 256.0
    eng = DinoEngine(arch=args.arch, img_size=224, out_dim=65536, batch=args.batch, n_local=n_local, lr=lr, weight_decay=0.04,
                     clip_grad=3.0, device=dev, reducer=reducer, precision="fp32" if args.precision == "fp32" else "bf16")
    from gipvit.models import init_vit_state, init_dino_head_state
    eng.load_state(init_vit_state(args.arch, 224, 0, seed=0), init_dino_head_state(eng.D, 65536, seed=1))
    tiles = synth_tiles(args.batch, 256, 1234 + rank, dev)
    micro_tiles = [tiles] + [synth_tiles(args.batch, 256, 4321 + 97 * j + rank, dev) for j in range(1, args.micro)]

    if args.random_crops:
        from gipvit.multicrop import MultiCropSampler
        sampler = MultiCropSampler(args.batch, 256, 2, n_local, seed=1234 + rank)
        views = None
        if args.view_augment:
            from gipvit.multicrop import ViewAugmentSampler
            views = ViewAugmentSampler(args.batch, 2, n_local, seed=77 + rank)
        step0 = lambda: eng.step(tiles, boxes=sampler.sample(dev), views=views.sample(dev) if views is not None else None)
    elif args.micro > 1:
        step0 = lambda: eng.step_micro(micro_tiles)
    else:
        step0 = lambda: eng.step(tiles)

    # The step's critical path (everything but the teacher forward and the weight-gradient GEMMs, which the
    # engine puts on its side stream) runs on a HIGH-priority stream: its kernels get the CU slots first and
    # the side stream's tiles fill what is left (measured 17.2 -> 16.7 ms/step).  BENCH_DEFAULT_STREAM=1 opts out.
    main_stream = None if os.environ.get("BENCH_DEFAULT_STREAM") else torch.cuda.Stream(dev, priority=-1)

    def on_main(fn):
        if main_stream is None:
            return fn()
        main_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(main_stream):
            out = fn()
        return out

    def step():
        l = on_main(step0)
        if args.trace_loss:
            torch.cuda.synchronize()
            print(f"[trace] t={eng.t} loss={float(l):.5f}", file=sys.stderr, flush=True)
    for _ in range(args.warmup):
        step()

    def barrier():
        if world > 1 or force_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if not os.environ.get("BENCH_NO_MID_BARRIER"):
        barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1 or force_dist:
        t = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    loss = float(eng.loss)
    # roofline of the dominant kernel: a few more steps of the same workload with per-launch timing on -- run by EVERY rank (the
    # steps contain the data-parallel all-reduces), reported from rank 0
    roof = roofline.dominant_kernel_roofline(lambda: on_main(lambda: eng.step(tiles)), steps=3, vit=eng.vit)

    # FLOPs actually executed per tile: BASELINE.md's count prices every token of every block (what the reference's module + autograd
    # compute); with the CLS-only last block (engine.VitRunner.cls_last) the projection + MLP of that block skip the non-CLS rows --
    # per such row 18 D^2 forward, and for the student 18 D^2 (dX) + 18 D^2 (dW) more in backward.
    gflop_ref = GFLOP_PER_TILE.get((args.arch, args.config))
    cls_only = bool(eng.vit._cls_tail(eng.g_stu))
    gflop_exec = gflop_ref
    if gflop_ref is not None and cls_only:
        rows_g, rows_l = 2 * (224 // 16) ** 2, n_local * (96 // 16) ** 2
        gflop_exec = gflop_ref - (54 * (rows_g + rows_l) + 18 * rows_g) * eng.D ** 2 / 1e9
    if rank == 0:
        tiles_s = args.batch * args.micro * world * args.steps / dt
        out = {
            "metric": "tiles/sec/GPU ViT-S/16 DINO (2g+8l crops, 256px) at 1/2/4/8 MI355X",
            "value": round(tiles_s, 2), "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "f16": "f16", "bf16": "bf16"}[args.precision], "data": "synthetic", "rccl_ranks": rccl_ranks, "dist_backend": dist_backend,
            "config": {"workload": f"{args.arch}/16 DINO 2x224+{n_local}x96 crops of 256px NHWC u8 tiles, K=65536 ({args.config})",
                       "tiles_per_gpu": args.batch * args.micro, "micro_batches": args.micro, "global_tiles": args.batch * args.micro * world, "parallelism": f"dp{world}",
                       "comm_cus": comm["comm_cus"], "cu_budget": comm["cu_budget"], "side_stream": eng.vit.side is not None, "main_stream_high_priority": main_stream is not None, "random_crops": bool(args.random_crops), "view_augment": bool(args.view_augment and args.random_crops)},
            "tiles_per_s_per_gpu": round(tiles_s / world, 2),
            "cls_only_last_block": cls_only,
            "gflop_per_tile": None if gflop_ref is None else {"every_token_of_every_block": gflop_ref, "executed": round(gflop_exec, 2)},
            "mfma_frac_whole_step": round(tiles_s / world * gflop_exec / 1e3
                                          / (MFMA_F32_PEAK_TFLOPS if args.precision == "fp32" else MFMA_BF16_PEAK_TFLOPS), 4)
            if gflop_ref is not None else None,
            "final_loss": round(loss, 4),
        }
        out["roofline"] = roof
        out["cpu_baseline"] = None if (args.no_cpu_baseline or world > 1) else cpu_baseline(args.arch, n_local)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
