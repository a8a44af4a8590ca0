#!/usr/bin/env python3
"""The reference's default scene with its own constants at 800x800 (four samples, mirror depth 4, ten
scatter rays), twenty frames: run under rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RAYS_HIP_FRAMES_IN_FLIGHT"] = "1"
import torch
from raytracing_amd import Context, Scene, render_params, scenes
ctx = Context(0)
scene = Scene(ctx, scenes.default_scene())
p = render_params(800, 800, 2, reflect_depth=4, diffuse_depth=2, diffuse_scatt=10, sun=scenes.SUN_T0, seed=1)
frame = torch.zeros((800, 800, 4), dtype=torch.uint8, device="cuda")
sh = torch.cuda.current_stream().cuda_stream
for _ in range(22):
    scene.render_device(p, frame.data_ptr(), None, sh)
torch.cuda.synchronize()
scene.close(); ctx.close()
