"""What does communication cost the step on ONE GPU?  (round-3 verdict, multi-GPU readiness.)

Every heavy launch of this build is one workgroup per CU holding the CU's whole register file and LDS, so RCCL's channel
workgroups cannot co-reside with them.  This tool runs the bench step with a STAND-IN for the gradient all-reduce: where the
engine releases a range (dist.reduction_plan: head, coalesced block ranges, the end message), C persistent copy workgroups on
a third stream stream that range twice (an all-reduce's reduce-scatter + all-gather passes over a rank's HBM) --
    GIPVIT_CU_BUDGET=<256 - C | 256> python tools/comm_standin.py --c C [--steps 20]
prints ms per step.  tools/comm_standin.sh runs the table: C = 0 / 8 / 16 / 32, launches sized for 256 CUs and for 256 - C."""
import argparse, ctypes, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--c", type=int, default=8, help="stand-in workgroups (CUs taken by communication); 0 = no communication")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--passes", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "lab_build", "libcomm_standin.so"))
lib.comm_standin_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


class StandInReducer:
    """engine.NoReducer's interface; world stays 1 (nothing is averaged), `always` makes the engine release its ranges."""
    world, always = 1, True

    def __init__(self, c):
        self.c = c
        self.stream = torch.cuda.Stream(dev)
        self.scratch = None
        self.bytes = 0

    def _copy(self, t):
        if self.c <= 0:
            return
        if self.scratch is None or self.scratch.numel() < t.numel():
            self.scratch = torch.empty(max(t.numel(), 48 << 20), dtype=t.dtype, device=dev)
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream()); self.stream.wait_event(ev)
        n = t.numel() * t.element_size() // 16 * 16
        rc = lib.comm_standin_launch(t.data_ptr(), self.scratch.data_ptr(), n, self.c, args.passes, self.stream.cuda_stream)
        assert rc == 0, rc
        self.bytes += n

    def reduce_range(self, buf, lo, hi):
        if hi > lo:
            self._copy(buf[lo:hi])

    def reduce_tensor(self, t):
        self._copy(t)

    def finish(self):
        torch.cuda.current_stream().wait_stream(self.stream)


from gipvit.engine import DinoEngine
from gipvit.models import init_vit_state, init_dino_head_state
sys.path.insert(0, ROOT)
from bench import synth_tiles
red = StandInReducer(args.c)
eng = DinoEngine(arch="vit_small", img_size=224, out_dim=65536, batch=64, n_local=8, lr=5e-4 * 64 / 256, weight_decay=0.04, clip_grad=3.0, device=dev, reducer=red)
eng.load_state(init_vit_state("vit_small", 224, 0, seed=0), init_dino_head_state(eng.D, 65536, seed=1))
tiles = synth_tiles(64, 256, 1234, dev)
main = torch.cuda.Stream(dev, priority=-1)


def step():
    main.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(main):
        eng.step(tiles)


for _ in range(args.warmup):
    step()
torch.cuda.synchronize()
red.bytes = 0
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"comm_workgroups": args.c, "cu_budget": int(os.environ.get("GIPVIT_CU_BUDGET", "256")), "ms_per_step": round(1e3 * dt / args.steps, 3),
                  "tiles_per_s": round(64 * args.steps / dt, 1), "standin_MB_per_step": round(red.bytes / args.steps / 1e6, 1),
                  "plan": [(str(t), (hi - lo) * 4 >> 20) for t, lo, hi in eng._reduction_plan()]}), flush=True)
