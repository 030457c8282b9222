"""Lab: where a wide panel's time goes, from s_memtime stamps (library built with -DGV_PANEL_STAMPS, tools/lab.sh):
    GIPVIT_LIB=tools/lab_build/lib_stamps.so python tools/panel_stamps.py
stamps per (workgroup, wave, panel): 0 loop top, 1 tile 0 landed, 2 k-loop done, 3 bias loaded, 4 first barrier passed,
5 fragments done, 6 last barrier passed."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gipvit import ops as o, _lib as L
dev = torch.device("cuda:0"); bf16 = torch.bfloat16
g = torch.Generator().manual_seed(1)
M, K = 44160, 384
for N, epi, name in ((1536, L.EPI_BIAS | L.EPI_GELU | L.EPI_SAVE_PRE, "fc1"), (1152, L.EPI_BIAS, "qkv")):
    A = torch.randn(M, K, generator=g).to(dev).to(bf16)
    W = (0.05 * torch.randn(N, K, generator=g)).to(dev).to(bf16)
    bias = torch.randn(N, generator=g).to(dev)
    C = torch.empty(M, N, dtype=bf16, device=dev); aux = torch.empty(M, N, dtype=bf16, device=dev)
    for _ in range(3):
        o.linear(A, W, C, M, N, K, epilogue=epi, bias=bias, aux_out=aux if epi & L.EPI_SAVE_PRE else None)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 8 * 8, dtype=np.uint64)
    rc = L.lib.gv_panel_dbg_read(buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    st = buf.reshape(256, 8, 8, 8).astype(np.int64)
    npan = int((st[0, 0, :, 0] > 0).sum())
    t0 = st[:, :, 0, 0][st[:, :, 0, 0] > 0].min()
    print(f"== {name}: panels per workgroup {npan}; memtime ticks (100 MHz = 10 ns each?) relative to the first stamp")
    for wg in (0, 9, 100):
        for wave in (0, 4):
            rows = []
            for it in range(npan):
                s = st[wg, wave, it, :7]
                rows.append(" ".join(f"{int(x - t0):6d}" for x in s))
            pass
    d = np.diff(st[:, :, :npan, :7], axis=-1)
    valid = st[:, :, :npan, 0] > 0
    names = ["tile0 wait", "k-loop", "bias load", "barrier A", "fragments", "barrier B"]
    for k, nm in enumerate(names):
        print(f"  {nm:10s}: mean {d[..., k][valid].mean():8.1f} ticks")
    span = (st[:, :, :npan, 6].max() - t0)
    print(f"  whole launch: {span} ticks")

# ---- the LayerNorm-fused kernels (stamps: 0 loop top, 1 tile 0 landed, 2 k-loop done (+ ring drained), 3 epilogue start, 6 epilogue end)
f32 = torch.float32
for name, K in (("proj + LN fwd", 384), ("fc2 + LN fwd", 1536)):
    A = torch.randn(M, K, generator=g).to(dev).to(bf16)
    W = (0.05 * torch.randn(384, K, generator=g)).to(dev).to(bf16)
    bias = torch.randn(384, generator=g).to(dev); resid = torch.randn(M, 384, generator=g).to(dev)
    out = torch.empty(M, 384, dtype=f32, device=dev); y = torch.empty(M, 384, dtype=bf16, device=dev)
    gam, bet = torch.ones(384, device=dev), torch.zeros(384, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    for _ in range(3):
        o.linear_ln_fwd(A, W, out, M, K, bias=bias, resid=resid, gamma=gam, beta=bet, y=y, mean=mean, rstd=rstd)
    torch.cuda.synchronize()
    rc = L.lib.gv_panel_dbg_read(buf.ctypes.data_as(ctypes.c_void_p)); assert rc == 0
    st = buf.reshape(256, 8, 8, 8).astype(np.int64)[:, :, 0, :]
    v = st[:, :, 0] > 0
    print(f"== {name}: tile0 wait {(st[..., 1] - st[..., 0])[v].mean():.0f}  k-loop {(st[..., 2] - st[..., 1])[v].mean():.0f}  "
          f"epilogue {(st[..., 6] - st[..., 3])[v].mean():.0f}  (pre-epilogue {(st[..., 3] - st[..., 2])[v].mean():.0f}) ticks; "
          f"span {st[..., 6][v].max() - st[..., 0][v].min()}")
for name, K in (("qkv dX + LN bwd", 1152), ("fc1 dX + LN bwd", 1536)):
    dY = torch.randn(M, K, generator=g).to(dev).to(bf16)
    W = (0.05 * torch.randn(K, 384, generator=g)).to(dev).to(bf16)
    x = torch.randn(M, 384, generator=g).to(dev); gg = torch.randn(M, 384, generator=g).to(dev)
    gb = torch.empty(M, 384, dtype=bf16, device=dev)
    gam = torch.ones(384, device=dev); mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
    part = torch.empty(L.LN_PARTIAL_BLOCKS, 3, 384, device=dev)
    for _ in range(3):
        o.linear_ln_bwd(dY, W, x, mean, rstd, gam, gg, gb, part, M, K)
    torch.cuda.synchronize()
    rc = L.lib.gv_panel_dbg_read(buf.ctypes.data_as(ctypes.c_void_p)); assert rc == 0
    st = buf.reshape(256, 8, 8, 8).astype(np.int64)[:, :, 0, :]
    v = st[:, :, 0] > 0
    print(f"== {name}: tile0 wait {(st[..., 1] - st[..., 0])[v].mean():.0f}  k-loop {(st[..., 2] - st[..., 1])[v].mean():.0f}  "
          f"epilogue {(st[..., 6] - st[..., 3])[v].mean():.0f}  (pre-epilogue {(st[..., 3] - st[..., 2])[v].mean():.0f}) ticks; "
          f"span {st[..., 6][v].max() - st[..., 0][v].min()}")
