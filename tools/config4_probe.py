#!/usr/bin/env python3
"""Config 4 probe: soup-N with diffuse bounces at a given frame size. usage: config4.py [tris] [w] [h]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import Context, Scene, render_params, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 960
H = int(sys.argv[3]) if len(sys.argv) > 3 else 540
t0 = time.perf_counter(); objs = scenes.soup(n, 0x5EED0002); t1 = time.perf_counter()
ctx = Context(0); sc = Scene(ctx, objs); t2 = time.perf_counter()
print(f"soup {n}: generate {t1-t0:.1f} s, build {t2-t1:.1f} s", flush=True)
frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
for dd, label in ((1, "primary+shadow"), (2, "diffuse x10")):
    p = render_params(W, H, 2, reflect_depth=4, diffuse_depth=dd, sun=scenes.SUN_T0, seed=int(os.environ.get("PROBE_SEED", "0")))
    st = sc.render_device_stats(p, frame.data_ptr())
    sc.render_device(p, frame.data_ptr()); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): sc.render_device(p, frame.data_ptr())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print(f"{label:16s} {W}x{H}: {dt*1e3:8.2f} ms, {st['rays']/1e6:8.1f} Mrays, {st['rays']/dt/1e6:8.1f} Mrays/s, tri/ray {st['tri_tests']/st['rays']:.1f}", flush=True)
