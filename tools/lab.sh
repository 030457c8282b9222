#!/bin/bash
# Lab build of ONE translation unit with extra -D switches, linked against the product's other objects into
# tools/lab_build/lib_<tag>.so.  The unit is taken from tools/lab/csrc/ when a lab copy exists there (the copies keep the
# GV_LAB_* / GV_NT_* / *_STAMPS switches that the product sources no longer carry), else from the product's csrc/.  The product library is never touched: a lab run selects its library with
#   GIPVIT_LIB=tools/lab_build/lib_<tag>.so python tools/wide_bench.py        (gipvit/_lib.py reads GIPVIT_LIB)
# Built here (hipcc cross-compiles), the .so travels to the GPU box with the snapshot.
#   tools/lab.sh <tag> <unit: panel|gemm|attention|...> "<-D switches>"
set -e
TAG="$1"; UNIT="$2"; DEFS="$3"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PKG="$ROOT/gipmed-project-self-supervised-vit_amd"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OUT="$ROOT/tools/lab_build"; mkdir -p "$OUT"
python3 "$PKG/build.py" > /dev/null                      # product objects up to date
SRC="$ROOT/tools/lab/csrc/$UNIT.hip"; [ -f "$SRC" ] || SRC="$PKG/csrc/$UNIT.hip"
"$HIPCC" --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -I"$PKG/csrc" $DEFS -c "$SRC" -o "$OUT/$TAG.o"
OBJS=$(ls "$PKG"/csrc/_obj/*.o | grep -v "/$UNIT.o")
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/lib_$TAG.so" "$OUT/$TAG.o" $OBJS
rm -f "$OUT/$TAG.o"
echo "$OUT/lib_$TAG.so"
