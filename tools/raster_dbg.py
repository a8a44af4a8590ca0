#!/usr/bin/env python3
"""Debug-build helper: renders one scene of tools/cells_fuzz.py by projection and by tree walk, and for
the first sample whose hit ids differ has the debug library (built with -DRT_DEBUG_RASTER) report what the
binning and the tile kernel did with the triangle the tree walk found.
usage: FUZZ_LARGE=1 RAYS_HIP_LIB=.../librays_hip_dbg.so python tools/raster_dbg.py <seed> [width height]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402,F401

import cells_fuzz as fuzz  # noqa: E402
from raytracing_amd import RT_MISS, Context, Scene, render_params  # noqa: E402

seed = int(sys.argv[1])
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 360)
rng = np.random.default_rng(seed)
sun = fuzz.sun_of(rng)
objs = fuzz.scene_of(rng, sun)
ctx = Context(0)
g = Scene(ctx, objs)
p = render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=sun)
os.environ["RAYS_HIP_RASTER_RATIO"] = "1e9"
os.environ["RAYS_HIP_RASTER_BLOCKS"] = "1e30"
os.environ["RAYS_HIP_SHADOW"] = "walk"
frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
st = g.render_device_stats(p, frame.data_ptr())
print("camera rays by projection:", st["camera_by_projection"], flush=True)
_, ids = g.render(p, want_hit_ids=True)
os.environ["RAYS_HIP_RASTER"] = "0"
_, ids_walk = g.render(p, want_hit_ids=True)
del os.environ["RAYS_HIP_RASTER"]
bad = np.nonzero(ids != ids_walk)[0]
print(f"{len(bad)} samples differ", flush=True)
verts = objs[0]["verts"].reshape(-1, 3, 3)
for s in bad[:4]:
    pix, sub = divmod(int(s), 4)
    x, y = pix % W, pix // W
    subx, suby = sub & 1, 1 if sub in (1, 2) else 0
    su, sw = 2 * x + subx, 2 * y + suby
    tri = int(ids_walk[s]) if ids_walk[s] != RT_MISS else int(ids[s])
    print(f"sample {s}: pixel ({x},{y}) sub {sub} -> column {su} row {sw}; projection {ids[s]} tree walk {ids_walk[s]}; reporting on {tri}", flush=True)
    print("vertices:\n", verts[tri], flush=True)
    os.environ["RAYS_HIP_DBG_TRI"], os.environ["RAYS_HIP_DBG_SU"], os.environ["RAYS_HIP_DBG_SW"] = str(tri), str(su), str(sw)
    g.render(p)
    torch.cuda.synchronize()
    sys.stdout.flush()
g.close(); ctx.close()
