#!/bin/bash
# Lab: which operands should carry a nontemporal hint so that the NEXT kernel's A operand is still in L2 / Infinity Cache?
# The product stores nontemporally: the saved GELU pre-activation (fc1's epilogue), the wide products' outputs up to FM = 8,
# the residual-gradient rows of the backward full-row kernel.  This script times further candidates against it on one box,
# back to back: -DGV_NT_XOUT / -DGV_NT_Y (the f32 residual row / the LayerNorm output of the forward full-row kernel),
# -DGV_NT_X (LayerNorm backward's input row), -DGV_NT_AUX_LD (the pre-activation read of GELU'-dX), -DGV_NT_DWX (the saved
# activations the dW launch stages: slower, its workgroups share them through L2).  Prints the per-kernel rows of every build.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PKG="$ROOT/gipmed-project-self-supervised-vit_amd"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
show() { python3 - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], "tiles/s", d["ms_per_step"], "ms")
for k in d["roofline"]["gemm_kernels"][:10]:
    print("   ", k["kernel"], k["avg_us"], "us", k["ms_per_step"], "ms/step")
PY
}
python3 "$ROOT/bench.py" --no-cpu-baseline > "$ROOT/gpurun_out/nt_base.json" 2> /dev/null
show "$ROOT/gpurun_out/nt_base.json"
cp "$PKG/libgipvit_hip.so" /tmp/libgipvit_product.so
i=0
for DEFS in "-DGV_NT_XOUT" "-DGV_NT_Y" "-DGV_NT_X -DGV_NT_AUX_LD"; do
    i=$((i+1))
    "$HIPCC" --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $DEFS -c "$PKG/csrc/panel.hip" -o /tmp/panel_lab.o
    "$HIPCC" --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $DEFS -c "$PKG/csrc/gemm.hip" -o /tmp/gemm_lab.o
    OBJS=$(ls "$PKG"/csrc/_obj/*.o | grep -v '/panel.o' | grep -v '/gemm.o')
    "$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$PKG/libgipvit_hip.so" /tmp/panel_lab.o /tmp/gemm_lab.o $OBJS
    python3 "$ROOT/bench.py" --no-cpu-baseline > "$ROOT/gpurun_out/nt_lab$i.json" 2> /dev/null
    echo "== $DEFS"
    show "$ROOT/gpurun_out/nt_lab$i.json"
done
cp /tmp/libgipvit_product.so "$PKG/libgipvit_hip.so"
python3 "$ROOT/bench.py" --no-cpu-baseline > "$ROOT/gpurun_out/nt_base2.json" 2> /dev/null
show "$ROOT/gpurun_out/nt_base2.json"
