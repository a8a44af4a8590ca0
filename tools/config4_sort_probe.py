#!/usr/bin/env python3
"""Scatter rays of the 10 M-triangle soup (ten cosine-weighted rays off every camera-ray hit of a
960 x 540 x 4 frame) through the batch nearest-hit call, in the frame's own order and sorted by
origin cell and by direction octant. usage: config4_sort_probe.py [tris]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import Context, Scene, scenes, RT_MISS

n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
W, H, G = 960, 540, 2
objs = scenes.soup(n_tris, 0x5EED0002)
ctx = Context(0); sc = Scene(ctx, objs)
sh = torch.cuda.current_stream().cuda_stream
dev = "cuda"
# camera rays in 8 x 8 pixel blocks (the frame's own order is tiled, too), four samples each
ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
blk = ((ys // 8) * ((W + 7) // 8) + xs // 8) * 64 + (ys % 8) * 8 + xs % 8
order = torch.argsort(blk.reshape(-1))
px = xs.reshape(-1)[order].repeat_interleave(4).float(); py = ys.reshape(-1)[order].repeat_interleave(4).float()
s = torch.arange(4, device=dev).repeat(W * H)
ox = torch.where(s % 2 == 1, 0.75, 0.25); oy = torch.where((s == 1) | (s == 2), 0.75, 0.25)
d = torch.stack([(2 * (px + ox) / W - 1) * float(W // H), 1 - 2 * (py + oy) / H, torch.ones_like(px)], 1)
rays = torch.cat([torch.zeros_like(d), d], 1).contiguous()
N0 = rays.shape[0]
out = torch.zeros(N0 * 4, dtype=torch.int32, device=dev)
sc.trace_closest_device(rays.data_ptr(), N0, out.data_ptr(), sh); torch.cuda.synchronize()
o4 = out.view(N0, 4)
hit = o4[:, 0] != -1
t = o4[:, 1].view(torch.float32)
normals = torch.from_numpy(objs[0]["tri_data"][:, :3].copy()).to(dev)
# hit ids index the scene's own triangle order, which for a soup is the input order
n = normals[o4[hit, 0].long()]
pos = d[hit] * t[hit, None] + n * 1e-5
g = torch.Generator(device=dev); g.manual_seed(1)
K = 10
nn = n.repeat_interleave(K, 0); pp = pos.repeat_interleave(K, 0)
u1 = torch.rand(nn.shape[0], device=dev, generator=g); u2 = torch.rand(nn.shape[0], device=dev, generator=g)
st_, ct = torch.sqrt(u1), torch.sqrt(1 - u1); psi = 2 * np.pi * u2
arb = torch.where(nn[:, 2:3].abs() < 1e-6, torch.tensor([0.0, 0.0, 1.0], device=dev), torch.tensor([1.0, 0.0, 0.0], device=dev))
t1 = torch.nn.functional.normalize(torch.cross(nn, arb, dim=1), dim=1); t2 = torch.nn.functional.normalize(torch.cross(nn, t1, dim=1), dim=1)
dd = torch.nn.functional.normalize((st_ * torch.cos(psi))[:, None] * t1 + (st_ * torch.sin(psi))[:, None] * t2 + ct[:, None] * nn, dim=1)
N = dd.shape[0]
def spread3(v):
    v = v & 0x3FF
    v = (v | (v << 16)) & 0x30000FF; v = (v | (v << 8)) & 0x300F00F; v = (v | (v << 4)) & 0x30C30C3; v = (v | (v << 2)) & 0x9249249
    return v
q = torch.clamp(((pp - torch.tensor([-10.0, -10.0, 5.0], device=dev)) / 20.0 * 1024).long(), 0, 1023)
morton = spread3(q[:, 0]) | (spread3(q[:, 1]) << 1) | (spread3(q[:, 2]) << 2)
octant = (dd[:, 0] > 0).long() | ((dd[:, 1] > 0).long() << 1) | ((dd[:, 2] > 0).long() << 2)
orders = {
    "frame order": torch.arange(N, device=dev),
    "origin cell (30 bits)": torch.argsort(morton, stable=True),
    "origin cell (15 bits) + octant": torch.argsort(((morton >> 15) << 3) | octant, stable=True),
    "octant + origin cell": torch.argsort((octant << 30) | morton, stable=True),
    "shuffled": torch.randperm(N, device=dev, generator=g),
}
res = torch.zeros(N * 4, dtype=torch.int32, device=dev)
print(f"soup-{n_tris}: {N} scatter rays off {int(hit.sum())} hits", flush=True)
for name, idx in orders.items():
    r = torch.cat([pp[idx], dd[idx]], 1).contiguous()
    sc.trace_closest_device(r.data_ptr(), N, res.data_ptr(), sh); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): sc.trace_closest_device(r.data_ptr(), N, res.data_ptr(), sh)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"  {name:32s} {dt*1e3:8.2f} ms {N/dt/1e6:8.1f} Mrays/s", flush=True)
sc.close(); ctx.close()
