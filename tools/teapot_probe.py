#!/usr/bin/env python3
"""Frame times of the reference's default scene (config 2: 1920x1080, four samples, camera + shadow rays;
and the reference's full constants at 800x800) under the library's choices and with them overridden.
usage: python tools/teapot_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from raytracing_amd import Context, Scene, render_params, scenes  # noqa: E402

os.environ["RAYS_HIP_FRAMES_IN_FLIGHT"] = "1"
ctx = Context(0)
scene = Scene(ctx, scenes.default_scene())
sh = torch.cuda.current_stream().cuda_stream


def timed(p, steps=30):
    frame = torch.zeros((p.height, p.width, 4), dtype=torch.uint8, device="cuda")
    st = scene.render_device_stats(p, frame.data_ptr())
    for _ in range(3):
        scene.render_device(p, frame.data_ptr(), None, sh)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        scene.render_device(p, frame.data_ptr(), None, sh)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return st, dt


c2 = render_params(1920, 1080, 2, reflect_depth=1, diffuse_depth=1, sun=scenes.SUN_T0)
full = render_params(800, 800, 2, reflect_depth=4, diffuse_depth=2, diffuse_scatt=10, sun=scenes.SUN_T0, seed=1)
for label, env in [("default", {}), ("tree walk", {"RAYS_HIP_RASTER": "0", "RAYS_HIP_SHADOW": "walk"}),
                   ("projection + walk", {"RAYS_HIP_RASTER_RATIO": "1e9", "RAYS_HIP_SHADOW": "walk"}),
                   ("projection + cells", {"RAYS_HIP_RASTER_RATIO": "1e9", "RAYS_HIP_SHADOW": "cells"}),
                   ("walk + cells", {"RAYS_HIP_RASTER": "0", "RAYS_HIP_SHADOW": "cells"})]:
    os.environ.update(env)
    st, dt = timed(c2)
    print(f"config 2, {label:20s}: {dt * 1e3:6.3f} ms  {st['rays'] / dt / 1e6:8.0f} Mrays/s  projection {st['camera_by_projection']} cells {st['shadow_by_cells']}",
          flush=True)
    for k in env:
        del os.environ[k]
st, dt = timed(full, 10)
print(f"reference constants 800x800: {dt * 1e3:6.3f} ms  {st['rays'] / dt / 1e6:8.0f} Mrays/s  {1 / dt:5.1f} frames/s")
scene.close()
ctx.close()
