#!/usr/bin/env python3
"""Prints the per-kernel summary of a rocprofv3 --kernel-trace --stats run (kernel_stats.csv).
usage: stats_read.py <output directory of the run>"""
import csv
import glob
import os
import sys

for path in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)):
    print("#", os.path.relpath(path, sys.argv[1]))
    for r in csv.DictReader(open(path)):
        name = r["Name"]
        name = name[name.find("k_"):].split("(")[0] if "k_" in name else name[:60]
        print(f"{name:40s} calls={int(r['Calls']):5d} avg_us={float(r['AverageNs'])/1e3:10.2f} "
              f"total_ms={float(r['TotalDurationNs'])/1e6:9.3f} pct={float(r['Percentage']):6.2f}")
