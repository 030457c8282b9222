"""Lab: timelines of the two-context wide kernel (csrc/panel2.hip) from s_memtime stamps.
    bash tools/lab/build_wide2.sh
    GIPVIT_WIDE2=1 GIPVIT_LIB=tools/lab_build/lib_w2.so [GIPVIT_WIDE2_LAB=bits] [GIPVIT_WIDE2_STAGGER=n] python tools/wide2_stamps.py
Per workgroup and panel: k-loop start, k-loop end, epilogue end (100-MHz? no: shader-clock s_memtime ticks)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gipvit import ops as o, _lib as L
dev = torch.device("cuda:0"); bf16 = torch.bfloat16
g = torch.Generator().manual_seed(1)
M, K = int(os.environ.get("W2_M", "44160")), int(os.environ.get("W2_K", "384"))


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
    return ts[len(ts) // 2], ts[0]


for N, epi, name in ((1536, L.EPI_BIAS | L.EPI_GELU | L.EPI_SAVE_PRE, "fc1"), (1152, L.EPI_BIAS, "qkv")):
    A = torch.randn(M, K, generator=g).to(dev).to(bf16)
    W = (0.05 * torch.randn(N, K, generator=g)).to(dev).to(bf16)
    bias = torch.randn(N, generator=g).to(dev)
    C = torch.empty(M, N, dtype=bf16, device=dev); aux = torch.empty(M, N, dtype=bf16, device=dev)
    fn = lambda: o.linear(A, W, C, M, N, K, epilogue=epi, bias=bias, aux_out=aux if epi & L.EPI_SAVE_PRE else None)
    t = timeit(fn)
    fn(); torch.cuda.synchronize()
    buf = np.zeros(1024 * 8 * 4, dtype=np.uint64)
    rc = L.lib.gv_wide2_dbg_read(buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    st = buf.reshape(1024, 8, 4).astype(np.int64)
    live = st[:, 0, 0] > 0
    npan = int((st[live][0, :, 0] > 0).sum())
    t0 = st[live][:, 0, 0].min()
    role = st[:, 0, 3] & 0xFF
    key = st[:, 0, 3] >> 8
    print(f"== {name} M {M} K {K}: {t[0]:.1f} us (min {t[1]:.1f}); {int(live.sum())} workgroups, {npan} panels each, "
          f"{len(set(key[live].tolist()))} distinct CU keys, roles: {np.bincount(role[live].astype(int)).tolist()}")
    kl = (st[live][:, :npan, 1] - st[live][:, :npan, 0])
    ep = (st[live][:, :npan, 2] - st[live][:, :npan, 1])
    for r in sorted(set(role[live].tolist())):
        m = role[live] == r
        print(f"   role {r}: first k-loop starts at {np.median(st[live][m][:, 0, 0] - t0):8.0f}; k-loop per panel {np.array2string(np.median(kl[m], axis=0), precision=0)}; "
              f"epilogue per panel {np.array2string(np.median(ep[m], axis=0), precision=0)}; ends at {np.median(st[live][m][:, npan - 1, 2] - t0):8.0f}")
    print(f"   launch span {st[live][:, :npan, 2].max() - t0} ticks")
    # one CU's two contexts side by side
    keys = key[live]; ids = np.nonzero(live)[0]
    for kk in list(dict.fromkeys(keys.tolist()))[:2]:
        for w in ids[keys == kk]:
            print(f"   cu {kk:5d} wg {w:4d} role {role[w]}: " + " | ".join(f"{st[w, i, 0] - t0:7d} {st[w, i, 1] - t0:7d} {st[w, i, 2] - t0:7d}" for i in range(npan)))
