#!/bin/bash
# Interleaved A/B of one lab build of csrc/panel.hip against the product library on one box:
#   tools/ab_lab.sh "-DGV_NT_XOUT -DGV_NT_Y" [pairs]
# prints tiles/s and ms/step of every run (product, lab, product, lab, ...).
set -e
DEFS="$1"; PAIRS="${2:-3}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PKG="$ROOT/gipmed-project-self-supervised-vit_amd"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
cp "$PKG/libgipvit_hip.so" /tmp/libgipvit_product.so
"$HIPCC" --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $DEFS -c "$PKG/csrc/panel.hip" -o /tmp/panel_lab.o
OBJS=$(ls "$PKG"/csrc/_obj/*.o | grep -v '/panel.o')
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o /tmp/libgipvit_lab.so /tmp/panel_lab.o $OBJS
run() { cp "$1" "$PKG/libgipvit_hip.so"; python3 "$ROOT/bench.py" --no-cpu-baseline --steps 40 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['value'], 'tiles/s', d['ms_per_step'], 'ms')"; }
for i in $(seq "$PAIRS"); do run /tmp/libgipvit_product.so product; run /tmp/libgipvit_lab.so "lab($DEFS)"; done
cp /tmp/libgipvit_product.so "$PKG/libgipvit_hip.so"
