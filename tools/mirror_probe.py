#!/usr/bin/env python3
"""Time of a mirror frame (reflect_depth 4, no diffuse rays: the megakernel k_render) of the reference's
own scene and of a soup. usage: [RAYS_HIP_LIB=..] python tools/mirror_probe.py [label]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import Context, Scene, _ffi, render_params, scenes

label = sys.argv[1] if len(sys.argv) > 1 else ""
ctx = Context(0)
for name, objects in (("teapot", scenes.default_scene()), ("soup-1M", scenes.soup(1000000, 0x5EED0001))):
    if name != "teapot":
        for ob in objects:
            ob["material"] = _ffi.material(1.0, 0.5, 0.5)
    scene = Scene(ctx, objects)
    p = render_params(1920, 1080, 2, reflect_depth=4, diffuse_depth=1, sun=scenes.SUN_T0)
    frame = torch.zeros((1080, 1920, 4), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        scene.render_device(p, frame.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        scene.render_device(p, frame.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print(f"{label:10s} {name:8s} mirror frame 1920x1080x4: {(time.time() - t0) * 100:.2f} ms", flush=True)
    scene.close()
ctx.close()
