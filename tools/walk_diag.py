#!/usr/bin/env python3
"""Which of the tree walks returns what for given samples of a scene of tools/cells_fuzz.py.
usage: [FUZZ_LARGE=1] [FUZZ_W= FUZZ_H=] python tools/walk_diag.py <seed> <sample>..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402,F401

import cells_fuzz as fuzz  # noqa: E402
from oracle import oracle_ffi  # noqa: E402
from raytracing_amd import Context, Scene, render_params, scenes  # noqa: E402

seed = int(sys.argv[1])
samples = [int(a) for a in sys.argv[2:]]
rng = np.random.default_rng(seed)
sun = fuzz.sun_of(rng)
objs = fuzz.scene_of(rng, sun)
W, H = fuzz.W, fuzz.H
p = render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=sun)
o = oracle_ffi.OracleScene(objs)
rays = np.concatenate([scenes.camera_rays(W, H, 2, pixels=[s // 4])[s % 4:s % 4 + 1] for s in samples])
print("oracle            ", o.trace_closest(rays)["tri"], o.trace_closest(rays)["t"])


def frame_ids(label, **env):
    os.environ.update({k: str(v) for k, v in env.items()})
    ctx = Context(0)
    g = Scene(ctx, objs)
    _, ids = g.render(p, want_hit_ids=True)
    print(f"{label:18s}", ids[samples])
    r = g.trace_closest(rays)
    print(f"{'  rt_trace_closest':18s}", r["tri"], r["t"])
    g.close(); ctx.close()
    for k in env:
        del os.environ[k]


frame_ids("projection", RAYS_HIP_RASTER_RATIO="1e9", RAYS_HIP_SHADOW="walk")
frame_ids("beam walk", RAYS_HIP_RASTER="0", RAYS_HIP_SHADOW="walk")
frame_ids("no beams", RAYS_HIP_RASTER="0", RAYS_HIP_SHADOW="walk", RAYS_HIP_BEAM_DIV="1e30")
frame_ids("megakernel", RAYS_HIP_RASTER="0", RAYS_HIP_SHADOW="walk", RAYS_HIP_MEGAKERNEL="1")
frame_ids("device-built tree", RAYS_HIP_RASTER="0", RAYS_HIP_SHADOW="walk", RAYS_HIP_BUILD="device")
o.close()
