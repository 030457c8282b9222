#!/bin/bash
# Tuning-lab runs of the ping-pong dW kernel (gemm_dw8.h); restores the product build afterwards.
#   tools/dw8_lab.sh stamps        library with -DGV_DW8_STAMPS: tools/dw_bench.py prints the per-segment cycles of a phase
#   tools/dw8_lab.sh variants      library with -DGV_DW8_LAB: ablation variants (GIPVIT_DW8_VAR) timed one after the other
set -e
MODE="${1:-stamps}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PKG="$ROOT/gipmed-project-self-supervised-vit_amd"
cp "$PKG/libgipvit_hip.so" /tmp/libgipvit_product.so
DEF=-DGV_DW8_STAMPS; [ "$MODE" = variants ] && DEF=-DGV_DW8_LAB
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $DEF -c "$PKG/csrc/gemm.hip" -o /tmp/gemm_lab.o
OBJS=$(ls "$PKG"/csrc/_obj/*.o | grep -v '/gemm.o')
hipcc --offload-arch=gfx950 -shared -fPIC -o "$PKG/libgipvit_hip.so" /tmp/gemm_lab.o $OBJS
if [ "$MODE" = variants ]; then
    for v in 0 32 24 56 0; do echo "VAR=$v"; GIPVIT_DW8_VAR=$v python "$ROOT/tools/dw_bench.py" 44160 1536x384 2>&1 | grep dW || true; done
else
    python "$ROOT/tools/dw_bench.py" || true
fi
cp /tmp/libgipvit_product.so "$PKG/libgipvit_hip.so"
