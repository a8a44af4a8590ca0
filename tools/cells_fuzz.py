#!/usr/bin/env python3
"""Randomised cross-check of the light-space shadow search and of the camera rays by projection: soups
of assorted triangle sizes and shapes (needles, large triangles, triangles turned edge-on to the sun,
flat layers at equal depth) under suns near and far on every side. Each scene is rendered three ways
(projection + cells, projection + tree walk, tree walk for everything); where two of them differ, in a
pixel or a hit id, the CPU oracle renders the frame and decides. A scene counts as bad when projection
or the cells differ from the oracle; a tree walk that differs from the oracle (it reaches a triangle
through the triangle's own box, the reference through the box of its leaf of eight, and a triangle test
that rounds badly can tell the two apart) is reported by itself.
FUZZ_ORACLE=1 has the oracle render every scene, not only those whose paths differ (the paths share
the rule by which they confirm a nearest hit, so their agreeing proves less than the oracle's does);
FUZZ_SPLIT=1 cuts the soup into three objects, two of them with trailing n % 8 triangles (=2: the last
one only, which projection accepts).
usage: [FUZZ_LARGE=1|medium] [FUZZ_ORACLE=1] [FUZZ_SPLIT=1] [FUZZ_W=640 FUZZ_H=360] python tools/cells_fuzz.py [rounds] [first seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from raytracing_amd import Context, Scene, render_params, scenes  # noqa: E402

W, H = int(os.environ.get("FUZZ_W", "640")), int(os.environ.get("FUZZ_H", "360"))


def as_object(v):
    n = len(v)
    nrm = np.nan_to_num(scenes.calculate_normals(v[:, 0], v[:, 1], v[:, 2]), nan=0.0, posinf=0.0, neginf=0.0)
    col = np.tile(np.array([[0.8, 0.6, 0.4]], dtype=np.float32), (n, 1))
    return {"verts": v.reshape(-1, 3), "tris": np.arange(3 * n, dtype=np.uint32).reshape(n, 3),
            "tri_data": np.concatenate([nrm, col], axis=1).astype(np.float32), "material": (0.3, 0.0, 1.0)}


def scene_of(rng, sun):
    n = int(rng.choice([40_000, 80_000, 160_000]))
    n -= n % 8
    c = np.stack([rng.uniform(-10, 10, n), rng.uniform(-6, 6, n), rng.uniform(5, 25, n)], axis=1)
    # in Morton order of the centres, as the benchmark soups are: a leaf's eight triangles lie together
    lo, hi = np.array([-10.0, -6.0, 5.0]), np.array([10.0, 6.0, 25.0])
    c = c[np.argsort(scenes._morton30(c, lo, hi), kind="stable")]
    size = rng.choice([0.05, 0.1, 0.3, 1.0], size=(n, 1), p=[0.35, 0.45, 0.17, 0.03])
    if os.environ.get("FUZZ_LARGE"):        # a few hundred triangles of 3 to 30 units among them
        large = rng.random(n) < 200.0 / n
        sizes = [1.5, 3.0, 6.0] if os.environ["FUZZ_LARGE"] == "medium" else [3.0, 10.0, 30.0]
        size[large] = rng.choice(sizes, size=(int(large.sum()), 1))
    e1 = rng.uniform(-1, 1, (n, 3)) * size
    e2 = rng.uniform(-1, 1, (n, 3)) * size
    kind = rng.integers(0, 100, n)
    kind = np.where(kind < 5, 0, np.where(kind < 6, 1, np.where(kind < 16, 2, 3)))   # 5 % needles, 1 % edge-on, 10 % flat
    needle = kind == 0                      # second edge nearly along the first
    e2[needle] = e1[needle] * rng.uniform(0.2, 1.0, (needle.sum(), 1)) + rng.normal(size=(needle.sum(), 3)) * 1e-3
    edge_on = kind == 1                     # first edge along the direction to the sun, or nearly
    to_sun = np.asarray(sun, dtype=np.float64) - c
    to_sun /= np.linalg.norm(to_sun, axis=1, keepdims=True)
    e1[edge_on] = to_sun[edge_on] * size[edge_on] * rng.uniform(0.5, 2.0, (edge_on.sum(), 1)) + \
        rng.normal(size=(edge_on.sum(), 3)) * rng.choice([0.0, 1e-6, 1e-4, 1e-2], size=(edge_on.sum(), 1))
    flat = kind == 2                        # layers across the light's dominant axis at a handful of depths
    ax = int(np.argmax(np.abs(np.asarray(sun) - np.array([0.0, 0.0, 15.0]))))
    e1[flat, ax] = 0.0
    e2[flat, ax] = 0.0
    levels = np.array([8.0, 8.0 + 1e-5, 12.0, 12.001, 20.0]) if ax == 2 else np.array([-2.0, -2.0 + 1e-5, 0.0, 1e-3, 3.0])
    c[flat, ax] = rng.choice(levels, size=flat.sum())
    a = c - (e1 + e2) / 3.0
    v = np.stack([a, a + e1, a + e2], axis=1).astype(np.float32)
    v[..., 2] = np.maximum(v[..., 2], 0.5)
    if os.environ.get("FUZZ_SPLIT") == "2":    # only the last object has trailing triangles: projection takes such scenes
        a = n // 3 - (n // 3) % 8
        return [as_object(v[:a]), as_object(v[a:2 * a]), as_object(v[2 * a:n - 3])]
    if os.environ.get("FUZZ_SPLIT"):
        a = n // 3 + 3 - (n // 3) % 8          # 3 trailing triangles
        b = a + n // 3 + 5 - (n // 3) % 8      # 5 trailing triangles
        return [as_object(v[:a]), as_object(v[a:b]), as_object(v[b:])]
    return [as_object(v)]


def sun_of(rng):
    far = rng.choice([30.0, 100.0, 1.0e3, 1.0e5])
    d = rng.normal(size=3)
    d /= np.linalg.norm(d)
    return tuple(float(x) for x in (np.array([0.0, 0.0, 15.0]) + d * far))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = Context(0)
    took = bad = walk_off = projected = 0
    for k in range(rounds):
        rng = np.random.default_rng(seed0 + k)
        sun = sun_of(rng)
        objs = scene_of(rng, sun)
        g = Scene(ctx, objs)
        p = render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=sun)
        frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        os.environ["RAYS_HIP_RASTER_RATIO"] = "1e9"   # projection even where the tree walk would be cheaper
        os.environ["RAYS_HIP_SHADOW"] = "cells"
        st = g.render_device_stats(p, frame.data_ptr())
        a, ids = g.render(p, want_hit_ids=True)
        os.environ["RAYS_HIP_SHADOW"] = "walk"
        b = g.render(p)
        os.environ["RAYS_HIP_RASTER"] = "0"
        c, ids_walk = g.render(p, want_hit_ids=True)
        del os.environ["RAYS_HIP_SHADOW"], os.environ["RAYS_HIP_RASTER"]
        diff = int((a != b).any(axis=2).sum())
        id_diff = int((ids != ids_walk).sum())
        took += int(st["shadow_by_cells"])
        projected += int(st["camera_by_projection"])
        verdict = ""
        if diff != 0 or id_diff != 0 or not np.array_equal(b, c) or os.environ.get("FUZZ_ORACLE"):
            from oracle import oracle_ffi
            o = oracle_ffi.OracleScene(objs)
            ref, ref_ids, _ = o.render(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=sun, want_hit_ids=True)
            o.close()
            # a, ids: projection + cells; b: projection + walk; c, ids_walk: walk + walk
            lsb = lambda x: int((np.abs(x.astype(np.int16) - ref.astype(np.int16)).max(axis=2) > 1).sum())  # noqa: E731
            fast_off = int((ids != ref_ids).sum()) + lsb(a)
            walks_off = int((ids_walk != ref_ids).sum()) + lsb(c)
            bad += fast_off != 0
            walk_off += walks_off != 0
            verdict = (f"  ORACLE: projection + cells off in {int((ids != ref_ids).sum())} ids / {lsb(a)} pixels, projection + walk in "
                       f"{lsb(b)} pixels, tree walk in {int((ids_walk != ref_ids).sum())} ids / {lsb(c)} pixels")
        print(f"seed {seed0 + k}: sun ({sun[0]:.1f}, {sun[1]:.1f}, {sun[2]:.1f})  tris {g.num_tris}  by cells {st['shadow_by_cells']}  "
              f"longest list {st['shadow_list_longest']}  shadow rays {st['rays'] - st['primary_rays']}  pixels that differ {diff}  "
              f"camera rays by projection {st['camera_by_projection']}  hit ids that differ {id_diff}{verdict}", flush=True)
        g.close()
    ctx.close()
    print(f"{rounds} scenes at {W}x{H}, {took} by cells, {projected} by projection; {bad} where projection or cells differ from the oracle, "
          f"{walk_off} where the tree walk does")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
