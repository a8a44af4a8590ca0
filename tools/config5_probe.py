#!/usr/bin/env python3
"""Config 5 on one GPU: soup-10M at 7680x4320, 16 samples per pixel (4x4 grid), camera + shadow rays.
Checks that the frame is deterministic and consistent with a sharded render of the same frame."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import Context, Scene, render_params, scenes
from raytracing_amd import distributed as rd
W, H = 7680, 4320
objs = scenes.soup(10_000_000, 0x5EED0002)
ctx = Context(0)
t0 = time.perf_counter(); sc = Scene(ctx, objs); print(f"build {time.perf_counter()-t0:.1f} s", flush=True)
p = render_params(W, H, 4, reflect_depth=1, diffuse_depth=1, sun=scenes.SUN_T0)
frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
st = sc.render_device_stats(p, frame.data_ptr())
sc.render_device(p, frame.data_ptr()); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3): sc.render_device(p, frame.data_ptr())
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
print(f"whole frame: {dt*1e3:.1f} ms, {st['rays']/1e6:.1f} Mrays, {st['rays']/dt/1e6:.0f} Mrays/s", flush=True)
ref = frame.clone()
# the same frame as 8 shards, assembled with the de-interleave kernel
n = 8
slab = rd.slab_tiles(W, H, n)
gathered = torch.zeros((n, slab * rd.TILE_BYTES), dtype=torch.uint8, device="cuda")
worst = 0.0
for r in range(n):
    pr = render_params(W, H, 4, reflect_depth=1, diffuse_depth=1, sun=scenes.SUN_T0, shard_index=r, shard_count=n)
    sc.render_device(pr, gathered[r].data_ptr()); torch.cuda.synchronize()
    t = time.perf_counter(); sc.render_device(pr, gathered[r].data_ptr()); torch.cuda.synchronize()
    worst = max(worst, time.perf_counter() - t)
out = torch.zeros_like(ref)
ctx.deinterleave_device(gathered.data_ptr(), slab, W, H, n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print(f"8 shards: slowest shard {worst*1e3:.1f} ms ({dt/worst:.2f}x of the whole frame), assembled frame equals whole frame: {bool(torch.equal(out, ref))}", flush=True)
