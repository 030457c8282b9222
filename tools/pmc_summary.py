#!/usr/bin/env python3
"""Average per-dispatch counter values per kernel from rocprofv3 --pmc CSV output.  python tools/pmc_summary.py <dir> [kernel substring]"""
import collections, csv, glob, os, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in rows.items():
    if flt in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())}, "n=", len(next(iter(cs.values()))))
