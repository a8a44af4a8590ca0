#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE lines of a pmc_summary.txt (tools/measure.sh) into the JSON that
bench.py's `traffic_profile` names: bytes per launch of each kernel, as the counters report them.

usage: tools/traffic_json.py gpurun_out/<tag>/pmc_summary.txt > profiles/<round>_pmc_traffic.json
"""
import json
import re
import sys


def main(path):
    out, kernel = {}, None
    for line in open(path):
        if not line.startswith(" ") and line.strip():
            kernel = line.strip()
            continue
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE|TCC_REQ_sum|TCC_HIT_sum|TCC_MISS_sum)\s+dispatches=\s*(\d+)\s+last=\S+\s+mean=(\S+)", line)
        if m and kernel:
            out.setdefault(kernel, {})[m.group(1)] = float(m.group(3))
    res = {}
    for k, v in out.items():
        # the instrumented (counting) variants of the frame kernels run once per bench and are not the product
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v or ("<true" in k and not k.startswith("k_raster_bin")):
            continue
        fetch_kb, write_kb = v["FETCH_SIZE"], v["WRITE_SIZE"]
        res[k] = {
            "fetch_kb": fetch_kb,
            "write_kb": write_kb,
            "bytes_per_launch": int((fetch_kb + write_kb) * 1000),
            "bytes_per_launch_fetch_doubled": int((2 * fetch_kb + write_kb) * 1000),
            "tcc_req": v.get("TCC_REQ_sum"),
            "tcc_hit": v.get("TCC_HIT_sum"),
            "tcc_miss": v.get("TCC_MISS_sum"),
        }
    doc = {
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_* in separate passes over bench.py --steps 2 (tools/measure.sh, tools/pmc.sh); mean over the dispatches of each kernel",
        "units": "FETCH_SIZE and WRITE_SIZE in KB of 1000 B as reported. On gfx950 FETCH_SIZE reports half the bytes of wide (16 B per lane) coalesced reads; bytes_per_launch_fetch_doubled applies that correction to all of the fetch (an upper bound for kernels that gather)",
        "kernels": res,
    }
    json.dump(doc, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1])
