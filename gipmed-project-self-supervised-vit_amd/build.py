"""Build libgipvit_hip.so and libgipvit_hip_f16.so (gfx950) in-tree with hipcc.  No torch involved.

The two libraries are the same sources: the second is compiled with -DGV_ACT_F16, which makes IEEE half the 16-bit operand /
activation format instead of bfloat16 (include/gipvit.h gv_act_format; reference --amp --amp-dtype float16, train.py:452-465)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libgipvit_hip.so")
# (library, object directory, extra flags)
VARIANTS = {"bf16": (LIB, OBJ, []), "f16": (os.path.join(HERE, "libgipvit_hip_f16.so"), os.path.join(CSRC, "_obj_f16"), ["-DGV_ACT_F16"])}
SOURCES = ["abi", "gemm", "panel", "layernorm", "rowops", "patch", "augment", "attention", "f32path", "dino_loss", "optim"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, variants=("bf16", "f16")) -> str:
    """Compile the requested variants; returns the path of the bf16 (product default) library."""
    for v in variants:
        _build_variant(v, force, verbose)
    return LIB


def _build_variant(variant: str, force: bool, verbose: bool) -> str:
    lib_path, obj_dir, extra = VARIANTS[variant]
    hipcc = _hipcc()
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "gipvit.h")]

    def compile_one(name):
        src, obj = os.path.join(CSRC, name + ".hip"), os.path.join(obj_dir, name + ".o")
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {name}.hip:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(lib_path, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
