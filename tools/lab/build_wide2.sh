#!/bin/bash
# Lab build of the two-context wide kernel (tools/lab/panel2.hip) into tools/lab_build/lib_w2.so: the lab copy of panel.hip with -DGV_LAB_WIDE2 (its
# gv_panel_wide then tries gv_panel_wide2 first) + panel2.hip with the stamps / ablation switches, linked with the product's other objects.
#   bash tools/lab/build_wide2.sh && GIPVIT_WIDE2=1 GIPVIT_LIB=tools/lab_build/lib_w2.so python tools/wide2_stamps.py
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
PKG="$ROOT/gipmed-project-self-supervised-vit_amd"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OUT="$ROOT/tools/lab_build"; mkdir -p "$OUT"
python3 "$PKG/build.py" > /dev/null
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -I$PKG/csrc"
"$HIPCC" $F -DGV_LAB_WIDE2 -c "$ROOT/tools/lab/csrc/panel.hip" -o "$OUT/w2_panel.o"
"$HIPCC" $F -DGV_WIDE2_LAB -c "$ROOT/tools/lab/panel2.hip" -o "$OUT/w2_panel2.o"
OBJS=$(ls "$PKG"/csrc/_obj/*.o | grep -v "/panel.o")
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/lib_w2.so" "$OUT/w2_panel.o" "$OUT/w2_panel2.o" $OBJS
rm -f "$OUT/w2_panel.o" "$OUT/w2_panel2.o"
echo "$OUT/lib_w2.so"
