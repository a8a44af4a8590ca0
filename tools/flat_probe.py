#!/usr/bin/env python3
"""Camera-phase time of frames full of axis-aligned flat geometry (a tessellated wall facing the camera,
a room of such walls), where a hit lies in the very plane of its leaf's box and the box test with the
ray cut at the hit can round either way.
usage: [RAYS_HIP_LIB=..] [FLAT_WALK=1] python tools/flat_probe.py [label]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import cells_fuzz as fuzz
from raytracing_amd import Context, Scene, render_params, scenes

W, H = int(os.environ.get("PROBE_W", "3840")), int(os.environ.get("PROBE_H", "2160"))


def grid(origin, du, dv, n):
    """n x n quads (two triangles each) spanning origin + [0,1] du + [0,1] dv, in Morton order of the quads."""
    origin, du, dv = (np.asarray(a, dtype=np.float64) for a in (origin, du, dv))
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    i, j = i.reshape(-1), j.reshape(-1)
    key = np.zeros(len(i), dtype=np.uint64)
    for b in range(12):
        key |= ((i >> b) & 1).astype(np.uint64) << np.uint64(2 * b) | ((j >> b) & 1).astype(np.uint64) << np.uint64(2 * b + 1)
    order = np.argsort(key, kind="stable")
    i, j = i[order], j[order]
    P = lambda a, b: origin[None, :] + (a / n)[:, None] * du[None, :] + (b / n)[:, None] * dv[None, :]
    p00, p10, p01, p11 = P(i, j), P(i + 1, j), P(i, j + 1), P(i + 1, j + 1)
    t = np.stack([np.stack([p00, p10, p11], axis=1), np.stack([p00, p11, p01], axis=1)], axis=1).reshape(-1, 3, 3)
    return t.astype(np.float32)


def scene_of(which, n=None):
    if which == "wall":      # one wall at z = 10 filling the frame
        return [fuzz.as_object(grid((-20, -12, 10), (40, 0, 0), (0, 24, 0), n or 200))]
    if which == "room":      # back wall, floor, ceiling, two side walls
        m = n or 96
        parts = [grid((-8, -4, 16), (16, 0, 0), (0, 8, 0), m), grid((-8, -4, 2), (16, 0, 0), (0, 0, 14), m),
                 grid((-8, 4, 2), (16, 0, 0), (0, 0, 14), m), grid((-8, -4, 2), (0, 8, 0), (0, 0, 14), m),
                 grid((8, -4, 2), (0, 8, 0), (0, 0, 14), m)]
        return [fuzz.as_object(np.concatenate(parts))]
    if which == "tilted":    # the same wall turned a little: nothing axis-aligned
        return [fuzz.as_object(grid((-20, -12, 10), (40, 0, 3), (0, 24, 2), n or 200))]
    raise SystemExit(which)


def main():
    hip = ctypes.CDLL("libamdhip64.so")
    def ev():
        e = ctypes.c_void_p(); assert hip.hipEventCreate(ctypes.byref(e)) == 0; return e
    def ms(a, b):
        m = ctypes.c_float(); assert hip.hipEventElapsedTime(ctypes.byref(m), a, b) == 0; return m.value

    label = sys.argv[1] if len(sys.argv) > 1 else ""
    os.environ["RAYS_HIP_RASTER_RATIO"] = "1e9"
    if os.environ.get("FLAT_WALK"):       # the same frames with every ray by tree walk
        os.environ["RAYS_HIP_RASTER"] = "0"
        os.environ["RAYS_HIP_SHADOW"] = "walk"
    ctx = Context(0)
    for which in ("wall", "tilted", "room"):
        scene = Scene(ctx, scene_of(which))
        p = render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=(3.0, 2.0, 5.0))
        frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        st = scene.render_device_stats(p, frame.data_ptr())
        sh = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            scene.render_device(p, frame.data_ptr(), None, sh)
        evs = [[ev() for _ in range(8)] for _ in range(6)]
        for e in evs:
            scene.render_device_events(p, frame.data_ptr(), [x.value for x in e], None, sh)
        torch.cuda.synchronize()
        ph = [np.mean([ms(e[k], e[k + 1]) for e in evs]) for k in range(4)]
        print(f"{label:12s} {which:7s} by projection {st['camera_by_projection']}  by cells {st['shadow_by_cells']}  total {sum(ph):6.2f} ms  cam {ph[0]:6.2f} sort {ph[1]:5.2f} shd {ph[2]:5.2f} res {ph[3]:5.2f}", flush=True)
        scene.close()
    ctx.close()


if __name__ == "__main__":
    main()
