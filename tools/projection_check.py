#!/usr/bin/env python3
"""Camera rays by projection against the tree walk on whole frames: hit ids must be identical.
usage: projection_check.py  (prints one line per case, then the total of differing samples)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import Context, Scene, render_params, scenes

cases = [(1_000_000, s, 3840, 2160, 2) for s in (1, 2, 3)] + \
        [(300_000, s, 1920, 1080, 2) for s in (4, 5)] + \
        [(100_000, 6, 1280, 720, 4), (2_000_000, 7, 3840, 2160, 2), (50_000, 8, 1000, 1000, 8), (500_000, 9, 2560, 1080, 1)]
ctx = Context(0)
sh = torch.cuda.current_stream().cuda_stream
total = 0
os.environ["RAYS_HIP_RASTER_RATIO"] = "1e9"
for n, seed, W, H, grid in cases:
    scene = Scene(ctx, scenes.soup(n, 0x5EED0100 + seed))
    p = render_params(W, H, grid, reflect_depth=1, diffuse_depth=1, sun=scenes.SUN_T0)
    out = []
    for mode in ("1", "0"):
        os.environ["RAYS_HIP_RASTER"] = mode
        frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        ids = torch.zeros(H * W * grid * grid, dtype=torch.int32, device="cuda")
        scene.render_device(p, frame.data_ptr(), ids.data_ptr(), sh)
        torch.cuda.synchronize()
        out.append((frame.clone(), ids.clone()))
    d = int((out[0][1] != out[1][1]).sum().item())
    same = bool(torch.equal(out[0][0], out[1][0]))
    hit = float((out[0][1] != -1).float().mean().item())
    total += d
    print(f"soup-{n} seed {seed} {W}x{H} grid {grid}: {d} of {ids.numel()} hit ids differ, frames equal {same}, hit rate {hit:.3f}", flush=True)
    scene.close()
print("total differing samples:", total)
ctx.close()
