#!/usr/bin/env python3
"""Shard 0 of 8 of the benchmark frame, twenty frames (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RAYS_HIP_FRAMES_IN_FLIGHT"] = "1"
import torch
from raytracing_amd import Context, Scene, render_params, scenes
W, H = 3840, 2160
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = Context(0)
scene = Scene(ctx, scenes.soup(1_000_000, 0x5EED0001))
frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
sh = torch.cuda.current_stream().cuda_stream
p = render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=scenes.SUN_T0, shard_index=0, shard_count=n)
for _ in range(23):
    scene.render_device(p, frame.data_ptr(), None, sh)
torch.cuda.synchronize()
scene.close(); ctx.close()
