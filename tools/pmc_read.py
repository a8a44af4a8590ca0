#!/usr/bin/env python3
"""Prints per-kernel counter values from rocprofv3 --pmc CSVs (last dispatch of each kernel).
usage: pmc_read.py <counter_collection.csv> [...]"""
import collections
import csv
import sys

acc = collections.OrderedDict()
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("k_wf", "k_render", "k_trace", "rocprim", "k_raster", "k_shadow", "k_lcell", "k_radix")):
            continue
        if "rocprim" in k:
            k = "rocprim radix sort (all passes)"
        else:
            k = k[k.find("k_"):].split("(")[0]
        acc.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for name, v in c.items():
        print(f"   {name:26s} dispatches={len(v):3d} last={v[-1]:.5g} mean={sum(v)/len(v):.5g}")
