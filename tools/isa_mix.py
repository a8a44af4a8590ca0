#!/usr/bin/env python3
"""Instruction mix of one kernel in the library's assembly listing (make -C raytracing_amd/csrc asm writes
raytracing_amd/lib/rt_kernels.s): vector, scalar and memory instructions per loop depth, lane-spill
instructions (v_readlane / v_writelane) among the vector ones, and the most frequent opcodes.
usage: tools/isa_mix.py <substring of the kernel's mangled name> [listing]"""
import collections
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
listing = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "raytracing_amd", "lib", "rt_kernels.s")
want = sys.argv[1]
lines = open(listing).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.endswith(":") is False and re.match(r"^_Z\w*" + re.escape(want) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
depth, per_depth, ops = 0, collections.defaultdict(lambda: [0, 0, 0, 0]), collections.Counter()
for l in lines[start:end]:
    if l.startswith(".LBB"):
        m = re.search(r"Depth=(\d+)", l)
        depth = int(m.group(1)) if m else 0
        continue
    if not l.startswith("\t") or l.startswith("\t;") or l.startswith("\t."):
        continue
    op = l.split()[0]
    ops[op] += 1
    c = per_depth[depth]
    if op.startswith("v_"):
        c[0] += 1
        if op.startswith("v_readlane") or op.startswith("v_writelane"):
            c[3] += 1
    elif op.startswith("s_"):
        c[1] += 1
    else:
        c[2] += 1
print(lines[start][:100])
print("loop depth: vector scalar memory (lane spills among the vector ones)")
for d in sorted(per_depth):
    v, s, m, sp = per_depth[d]
    print(f"  {d}: {v:6d} {s:6d} {m:5d} ({sp})")
print("most frequent:", ", ".join(f"{o} {n}" for o, n in ops.most_common(16)))
