#!/usr/bin/env python3
"""Hit-id agreement of the HIP path with the CPU oracle on the benchmark configurations, with the
mismatches classified. Run on a GPU box from the repo root; writes one JSON object.

  config 2: default scene, 1920x1080, every camera ray (8,294,400) - batch search and frame kernels
  config 3: 1M-triangle soup, 3840x2160: every 16th pixel's camera rays (2,073,600) against the oracle
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_ffi as O  # noqa: E402
from raytracing_amd import Context, Scene, render_params, scenes, RT_MISS  # noqa: E402


def classify(gpu_tri, gpu_t, ref):
    bad = np.flatnonzero(gpu_tri != ref["tri"])
    out = {"rays": int(len(ref)), "mismatches": int(len(bad)), "agreement": 1.0 - len(bad) / len(ref)}
    if len(bad):
        gm, rm = gpu_tri[bad] == RT_MISS, ref["tri"][bad] == RT_MISS
        both = ~gm & ~rm
        rel = np.abs(gpu_t[bad][both] - ref["t"][bad][both]) / np.maximum(ref["t"][bad][both], 1e-30)
        out["classes"] = {
            "gpu_miss_ref_hit": int((gm & ~rm).sum()),
            "gpu_hit_ref_miss": int((~gm & rm).sum()),
            "both_hit_t_within_1e-5": int((rel <= 1e-5).sum()),
            "both_hit_gpu_nearer": int(((rel > 1e-5) & (gpu_t[bad][both] < ref["t"][bad][both])).sum()),
            "both_hit_gpu_farther": int(((rel > 1e-5) & (gpu_t[bad][both] > ref["t"][bad][both])).sum()),
        }
    return out


def main():
    ctx = Context(0)
    report = {}
    # ---- config 2
    objs = scenes.default_scene()
    g, o = Scene(ctx, objs), O.OracleScene(objs)
    rays = scenes.camera_rays(1920, 1080, 2)
    t0 = time.perf_counter()
    ref = o.trace_closest(rays)
    cpu_s = time.perf_counter() - t0
    h = g.trace_closest(rays)
    report["config2_batch_search"] = classify(h["tri"], h["t"], ref)
    report["config2_batch_search"]["oracle_seconds"] = cpu_s
    _, ids = g.render(render_params(1920, 1080, 2, reflect_depth=1, diffuse_depth=1), want_hit_ids=True)
    bad = ids != ref["tri"]
    report["config2_frame_kernels"] = {"rays": int(len(ids)), "mismatches": int(bad.sum()),
                                       "agreement": 1.0 - float(bad.mean())}
    _, ids = g.render(render_params(1920, 1080, 2, seed=5), want_hit_ids=True)
    bad = ids != ref["tri"]
    report["config2_frame_full_recursion_primary_ids"] = {"rays": int(len(ids)), "mismatches": int(bad.sum())}
    g.close()
    # ---- config 3
    objs = scenes.soup(1_000_000, 0x5EED0001)
    g, o = Scene(ctx, objs), O.OracleScene(objs)
    W, H = 3840, 2160
    pix = np.arange(0, W * H, 16)
    rays = scenes.camera_rays(W, H, 2, pixels=pix)
    t0 = time.perf_counter()
    ref = o.trace_closest(rays)
    cpu_s = time.perf_counter() - t0
    h = g.trace_closest(rays)
    report["config3_batch_search_subsample"] = classify(h["tri"], h["t"], ref)
    report["config3_batch_search_subsample"]["oracle_seconds"] = cpu_s
    _, ids = g.render(render_params(W, H, 2, reflect_depth=1, diffuse_depth=1), want_hit_ids=True)
    sub = ids.reshape(-1, 4)[pix].reshape(-1)
    bad = sub != ref["tri"]
    report["config3_frame_kernels_subsample"] = {"rays": int(len(sub)), "mismatches": int(bad.sum()),
                                                 "agreement": 1.0 - float(bad.mean())}
    # shadow search on the same subsample: rays from the oracle's hit points towards the sun
    hit = ref["tri"] != RT_MISS
    pos = rays[hit, 0:3] + rays[hit, 3:6] * ref["t"][hit, None]
    sun = np.array(scenes.SUN_T0, dtype=np.float32)
    d = sun - pos
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    srays = np.concatenate([(pos + d * np.float32(1e-4)).astype(np.float32), d], axis=1)[:500000]
    a, b = g.trace_any(srays), o.trace_any(srays)
    report["config3_shadow_search_subsample"] = {"rays": int(len(srays)), "mismatches": int((a != b).sum()),
                                                 "occluded_fraction": float(b.mean())}
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
