#!/usr/bin/env python3
"""Looks into one scene of tools/cells_fuzz.py whose paths disagree: which samples differ, and what the
CPU oracle says about them.
usage: [FUZZ_LARGE=1] python tools/fuzz_diag.py <seed>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402,F401

import cells_fuzz as fuzz  # noqa: E402
from oracle import oracle_ffi  # noqa: E402
from raytracing_amd import RT_MISS, Context, Scene, render_params, scenes  # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
sun = fuzz.sun_of(rng)
objs = fuzz.scene_of(rng, sun)
W, H = fuzz.W, fuzz.H
ctx = Context(0)
g = Scene(ctx, objs)
o = oracle_ffi.OracleScene(objs)
p = render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=sun)
os.environ["RAYS_HIP_RASTER_RATIO"] = "1e9"
os.environ["RAYS_HIP_SHADOW"] = "cells"
a, ids = g.render(p, want_hit_ids=True)
os.environ["RAYS_HIP_SHADOW"] = "walk"
b, _ = g.render(p, want_hit_ids=True)
os.environ["RAYS_HIP_RASTER"] = "0"
c, ids_walk = g.render(p, want_hit_ids=True)
del os.environ["RAYS_HIP_SHADOW"], os.environ["RAYS_HIP_RASTER"]
verts = objs[0]["verts"].reshape(-1, 3, 3)
for s in np.nonzero(ids != ids_walk)[0]:
    pix, sub = divmod(int(s), 4)
    ray = scenes.camera_rays(W, H, 2, pixels=[pix])[sub:sub + 1]
    ref = o.trace_closest(ray)
    lin = g.trace_closest_linear(ray) if hasattr(g, "trace_closest_linear") else None
    print(f"sample {s} (pixel {pix % W},{pix // W} sub {sub}): projection {ids[s]}  tree walk {ids_walk[s]}  oracle {ref['tri'][0]} t {ref['t'][0]!r}"
          + (f"  linear scan on the GPU {lin['tri'][0]}" if lin is not None else ""))
    for t in {int(ids[s]), int(ids_walk[s]), int(ref["tri"][0])} - {int(RT_MISS)}:
        v = verts[t]
        print(f"   triangle {t}: box sides {np.ptp(v, axis=0)}  leaf {t // 8}")
diff = np.nonzero((a != b).any(axis=2).reshape(-1))[0]
for pix in diff:
    rays = scenes.camera_rays(W, H, 2, pixels=[int(pix)])
    ref = o.trace_closest(rays)
    print(f"pixel {pix % W},{pix // W}: cells {a.reshape(-1, 4)[pix]}  walk {b.reshape(-1, 4)[pix]}")
    for k in range(4):
        if ref["tri"][k] == RT_MISS:
            continue
        # the shadow ray of the sample, as the reference builds it (hit point moved out along the normal)
        t = int(ref["tri"][k])
        n = objs[0]["tri_data"][t, 0:3]
        d = rays[k, 3:6]
        pos = (d * np.float32(ref["t"][k])).astype(np.float32) + (n * np.float32(0.00001)).astype(np.float32)
        sd = (np.asarray(sun, dtype=np.float32) - pos).astype(np.float32)
        sd = (sd / np.float32(np.sqrt(np.float32((sd * sd).sum())))).astype(np.float32)
        sray = np.concatenate([pos, sd]).astype(np.float32)[None, :]
        occ_o = o.trace_any(sray)[0]
        occ_g = g.trace_any(sray)[0]
        print(f"   sample {k}: hit {t} lit {float((n * sd).sum()) > 0}  shadow ray occluded: oracle {occ_o}  gpu any-hit walk {occ_g}")
ref_rgba, ref_ids, _ = o.render(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=sun, want_hit_ids=True, pixel_stride=1) if len(diff) and os.environ.get("FULL") else (None, None, None)
if ref_rgba is not None:
    for pix in diff:
        print(f"pixel {pix}: oracle frame {ref_rgba.reshape(-1, 4)[pix]}")
g.close(); o.close(); ctx.close()
