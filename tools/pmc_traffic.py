#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes into HBM bytes per launch per kernel.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>

FETCH_SIZE / WRITE_SIZE are KiB.  On gfx950 FETCH_SIZE counts 128-B requests at 64 B
(MI355X_MICROARCH.md, HBM section), so the read side is doubled.  Kernel names are reduced to
`name<template args>` so that mangled and demangled spellings of one instantiation agree with
gipvit.ops.linear_timing_read()."""
import collections
import csv
import glob
import json
import re
import sys


def canon(name: str) -> str:
    m = re.search(r"gemm_kernelILb(\d)ELb(\d)E(DF16b|f)Lb(\d)ELi(n?\d+)E", name)
    if m:
        b = lambda x: "true" if x == "1" else "false"
        epi = m.group(5).replace("n", "-")
        return f"gemm_kernel<{b(m.group(1))}, {b(m.group(2))}, {'bf16' if m.group(3) == 'DF16b' else 'float'}, {b(m.group(4))}, {epi}>"
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z_0-9]+?)I(DF16b|f)EEv", name)      # rocprofv3 leaves __bf16 template arguments mangled
    if m:
        return f"{m.group(1)}<{'bf16' if m.group(2) == 'DF16b' else 'float'}>"
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("gvgemm::", "")
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("__hip_bfloat16", "bf16")


def fold(d, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = canon(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    return tot, cnt


def main():
    fd, wd, out = sys.argv[1:4]
    ft, fc = fold(fd, "FETCH_SIZE")
    wt, wc = fold(wd, "WRITE_SIZE")
    rows = []
    for k in sorted(ft, key=lambda k: -(2 * ft[k] + wt.get(k, 0))):
        n = fc[k]
        rows.append({"kernel": k, "launches": n, "read_MB_per_launch_x2corrected": round(2 * ft[k] * 1024 / n / 1e6, 2),
                     "write_MB_per_launch": round(wt.get(k, 0) * 1024 / max(wc.get(k, 1), 1) / 1e6, 2)})
    json.dump(rows, open(out, "w"), indent=1)
    for r in rows[:12]:
        print(r)


if __name__ == "__main__":
    main()
