#!/usr/bin/env python3
"""Strong scaling of the benchmark frame measured on one GPU: shard 0 of 1, 2, 4 and 8 rendered by
itself, one frame at a time (compute only, no gather), phase times from the library's events.
usage: python tools/shard_probe.py > profiles/<round>_shard_probe.txt"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RAYS_HIP_FRAMES_IN_FLIGHT"] = "1"
import torch  # noqa: E402

from raytracing_amd import Context, Scene, render_params, scenes  # noqa: E402

W, H = 3840, 2160
ctx = Context(0)
scene = Scene(ctx, scenes.soup(1_000_000, 0x5EED0001))
frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")   # a shard's slab is no larger
hip = ctypes.CDLL("libamdhip64.so")


def ev():
    e = ctypes.c_void_p()
    assert hip.hipEventCreate(ctypes.byref(e)) == 0
    return e


def ms(a, b):
    m = ctypes.c_float()
    assert hip.hipEventElapsedTime(ctypes.byref(m), a, b) == 0
    return m.value


sh = torch.cuda.current_stream().cuda_stream
whole = None
for n in (1, 2, 4, 8):
    p = render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=scenes.SUN_T0, shard_index=0, shard_count=n)
    for _ in range(3):
        scene.render_device(p, frame.data_ptr(), None, sh)
    evs = [[ev() for _ in range(8)] for _ in range(10)]
    for e in evs:
        scene.render_device_events(p, frame.data_ptr(), [x.value for x in e], None, sh)
        torch.cuda.synchronize()
    ph = [float(np.mean([ms(e[k], e[k + 1]) for e in evs])) for k in range(4)]
    total = float(np.mean([ms(e[0], e[4]) for e in evs]))
    whole = whole or total
    print(f"shard 0 of {n}: {total:5.2f} ms ({whole / total:4.2f} x)  camera rays {ph[0]:5.2f}  sort {ph[1]:5.2f}  "
          f"shadow rays {ph[2]:5.2f}  resolve {ph[3]:5.2f}", flush=True)
scene.close()
ctx.close()
