#!/usr/bin/env python3
"""Incoherent nearest-hit rays in soup-N: per-lane walk vs wave walk, unsorted vs sorted by origin cell (+ octant)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import Context, Scene, scenes
n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
rng = np.random.default_rng(1)
o = np.stack([rng.uniform(-10, 10, N), rng.uniform(-10, 10, N), rng.uniform(5, 8, N)], 1).astype(np.float32)
d = rng.normal(size=(N, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
def spread3(v):
    v = v.astype(np.uint64) & 0x3FF
    v = (v | (v << 16)) & 0x30000FF; v = (v | (v << 8)) & 0x300F00F; v = (v | (v << 4)) & 0x30C30C3; v = (v | (v << 2)) & 0x9249249
    return v
q = np.clip(((o - np.array([-10, -10, 5])) / 20.0 * 1024).astype(np.int64), 0, 1023)
morton = spread3(q[:, 0]) | (spread3(q[:, 1]) << 1) | (spread3(q[:, 2]) << 2)
octant = ((d[:, 0] > 0).astype(np.uint64) | ((d[:, 1] > 0).astype(np.uint64) << 1) | ((d[:, 2] > 0).astype(np.uint64) << 2))
orders = {
    "unsorted": np.arange(N),
    "origin30": np.argsort(morton, kind="stable"),
    "origin15+octant": np.argsort(((morton >> 15) << 3) | octant, kind="stable"),
    "octant+origin30": np.argsort((octant << 30) | morton, kind="stable"),
    "origin21+octant": np.argsort(((morton >> 9) << 3) | octant, kind="stable"),
}
ctx = Context(0); sc = Scene(ctx, scenes.soup(n_tris, 0x5EED0002))
out = torch.zeros(N * 4, dtype=torch.int32, device="cuda")
sh = torch.cuda.current_stream().cuda_stream
for name, idx in orders.items():
    rays = torch.from_numpy(np.concatenate([o[idx], d[idx]], 1).copy()).cuda()
    sc.trace_closest_device(rays.data_ptr(), N, out.data_ptr(), sh); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): sc.trace_closest_device(rays.data_ptr(), N, out.data_ptr(), sh)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    hits = (out.view(N, 4)[:, 0] != -1).float().mean().item()
    print(f"{os.environ.get('RAYS_HIP_TRACE_WAVE','lane'):5s} {name:18s} {dt*1e3:8.2f} ms {N/dt/1e6:8.1f} Mrays/s hit {hits:.3f}", flush=True)
