"""Lab: two half-batch DINO chains on DISJOINT halves of the chip (CU-masked streams), against one full-batch step.
The question: when the two halves are out of phase, do the HBM-bound epilogues of one half run under the
MFMA-bound k-loops of the other at the higher per-CU rate of profiles/r03_cu_stream_rate.txt?
    python tools/halfchip_chains.py full                      # one engine, B = 64, default streams
    GIPVIT_LIB=tools/lab_build/lib_cubudget.so GIPVIT_CU_BUDGET=128 python tools/halfchip_chains.py {plain|evenodd|lowhigh|xcd} [B per chain]
(lab library: tools/lab.sh cubudget panel -DGV_LAB_CU_BUDGET -- the full-row launches are then sized for 128 CUs)"""
import ctypes, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipvit.engine import DinoEngine
from gipvit.models import init_vit_state, init_dino_head_state
import bench

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32


def hip_lib():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    raise RuntimeError("libamdhip64 is not mapped")


def masked_stream(hip, words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def make(b):
    e = DinoEngine(arch="vit_small", img_size=224, out_dim=65536, batch=b, lr=1e-4, clip_grad=3.0, device=dev)
    e.load_state(init_vit_state("vit_small", 224, 0, seed=0), init_dino_head_state(384, 65536, seed=1))
    return e, bench.synth_tiles(b, 256, 1234, dev)


def run(engs, streams, steps=20, warm=5):
    def one():
        for (e, t), s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.step(t)
    for _ in range(warm): one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): one()
    t_host = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    print(f"  (host enqueue {t_host * 1e3:.2f} ms per iteration)", flush=True)
    return (time.perf_counter() - t0) / steps


torch.zeros(1, device=dev)
if mode == "full":
    eng = [make(2 * B)]
    dt = run(eng, [torch.cuda.Stream(dev, priority=-1)])
    print(f"full: one engine B={2 * B}: {dt * 1e3:.2f} ms/step  {2 * B / dt:.0f} tiles/s", flush=True)
else:
    hip = hip_lib()
    masks = {"plain": None,
             "evenodd": ([0x55555555] * 8, [0xAAAAAAAA] * 8),
             "lowhigh": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
             "xcd": ([0x0F0F0F0F] * 8, [0xF0F0F0F0] * 8)}[mode]
    engs = [make(B) for _ in range(2)]
    if masks is None:
        mains = [torch.cuda.Stream(dev, priority=-1) for _ in engs]
    else:
        mains = [masked_stream(hip, m) for m in masks]
        for (e, _), m in zip(engs, masks):
            e.vit.side = masked_stream(hip, m)
    dt = run(engs, mains)
    print(f"{mode}: two chains B={B} each (CU budget {os.environ.get('GIPVIT_CU_BUDGET', '256')}): {dt * 1e3:.2f} ms/step  {2 * B / dt:.0f} tiles/s", flush=True)
    dt1 = run(engs[:1], mains[:1])
    print(f"{mode}: one chain alone on its half: {dt1 * 1e3:.2f} ms/step  {B / dt1:.0f} tiles/s", flush=True)
