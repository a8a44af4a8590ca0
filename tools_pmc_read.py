#!/usr/bin/env python3
"""Prints per-kernel counter values from a rocprofv3 --pmc CSV (last dispatch of each kernel)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"][:60]
    acc.setdefault(k, collections.OrderedDict())
    acc[k].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "render" not in k and "trace" not in k: continue
    print(k)
    for name, v in c.items():
        print(f"   {name:28s} n={len(v)} last={v[-1]:.4g}")
