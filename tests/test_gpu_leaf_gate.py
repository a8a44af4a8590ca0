"""The reference tests a triangle only for rays that pass the box of its leaf of eight (src/main.rs:103-111,
192-200). Projection and the light-space cells reach triangles by other routes and put the hits they are
about to accept through that test (ref_leaf_gate); these scenes hold triangles for which it decides."""
import importlib.util
import os

import numpy as np
import pytest

from raytracing_amd import Scene, render_params

pytestmark = pytest.mark.gpu

W, H = 640, 360


def _fuzz(monkeypatch):
    monkeypatch.setenv("FUZZ_LARGE", "1")   # a few hundred triangles of 3 to 30 units among the small ones
    spec = importlib.util.spec_from_file_location(
        "cells_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "cells_fuzz.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    return fuzz


def _scene(fuzz, seed):
    rng = np.random.default_rng(seed)
    sun = fuzz.sun_of(rng)
    return fuzz.scene_of(rng, sun), sun


def _stats(g, p):
    import torch
    frame = torch.zeros((p.height, p.width, 4), dtype=torch.uint8, device="cuda")
    return g.render_device_stats(p, frame.data_ptr())


def test_needle_whose_test_rounds_badly_is_held_back_by_its_leaf_box(ctx, oracle, monkeypatch):
    """Seed 20094 of tools/cells_fuzz.py with FUZZ_LARGE: triangle 10754 is a needle 13 units long and
    0.1 wide; its triangle test, rounded as the reference rounds it, accepts the camera ray of sample
    column 27, row 584 at t = 4.4218 (u + v = 0.9999), in front of the triangle's own nearest vertex at
    z = 4.4226, and the ray passes the box of the needle's leaf by 0.0014. The reference never tests the
    needle for that ray and returns triangle 10830 behind it. Projection finds every triangle a sample's
    ray can be accepted by, so it has to apply the leaf's box test itself; the whole frame's hit ids must be
    the oracle's, with a 20-unit triangle in the scene whose box is nine frames wide."""
    fuzz = _fuzz(monkeypatch)
    objs, sun = _scene(fuzz, 20094)
    p = render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=sun)
    monkeypatch.setenv("RAYS_HIP_RASTER_RATIO", "1e9")   # projection whatever the work estimate says
    monkeypatch.setenv("RAYS_HIP_SHADOW", "walk")
    g = Scene(ctx, objs)
    o = oracle.OracleScene(objs)
    try:
        assert _stats(g, p)["camera_by_projection"] == 1
        rgba, ids = g.render(p, want_hit_ids=True)
        ref, ref_ids, _ = o.render(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=sun, want_hit_ids=True)
        s = (292 * W + 13) * 4 + 3
        assert ref_ids[s] == 10830 and ids[s] == 10830
        assert np.array_equal(ids, ref_ids), f"{int((ids != ref_ids).sum())} hit ids differ"
        assert np.abs(rgba.astype(np.int16) - ref.astype(np.int16)).max() <= 1
        # the older form of projection (nearest hits by atomic minima in device memory) applies the same test
        monkeypatch.setenv("RAYS_HIP_RASTER", "atomic")
        small = render_params(160, 90, 2, reflect_depth=1, diffuse_depth=1, sun=sun)
        _, ids_atomic = g.render(small, want_hit_ids=True)
        monkeypatch.setenv("RAYS_HIP_RASTER", "0")
        _, ids_walk = g.render(small, want_hit_ids=True)
        assert np.array_equal(ids_atomic, ids_walk)
    finally:
        g.close()
        o.close()


def test_ray_in_the_plane_of_a_large_triangle(ctx, oracle, monkeypatch):
    """Seed 20333: triangle 43741 (18 units) has an edge along the direction to the sun, and the shadow ray
    of pixel (207, 210) lies in its plane, two units outside the triangle's own box but inside the box of its
    leaf. The determinant of the reference's test is rounding noise there, above the test's epsilon, and the
    test accepts (t = 2.24, u = 0.38, v = 0.33): the reference calls the pixel shadowed. The light-space cells
    give such a triangle the patch of its leaf's box and agree; a tree over the triangles' own boxes cannot
    (its deviation is one pixel here and stays written down, not asserted away)."""
    fuzz = _fuzz(monkeypatch)
    objs, sun = _scene(fuzz, 20333)
    p = render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=sun)
    g = Scene(ctx, objs)
    o = oracle.OracleScene(objs)
    try:
        ref, ref_ids, _ = o.render(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=sun, want_hit_ids=True)
        monkeypatch.setenv("RAYS_HIP_SHADOW", "cells")
        assert _stats(g, p)["shadow_by_cells"] == 1
        rgba, ids = g.render(p, want_hit_ids=True)
        assert np.array_equal(ids, ref_ids)
        assert np.abs(rgba.astype(np.int16) - ref.astype(np.int16)).max() <= 1
        monkeypatch.setenv("RAYS_HIP_SHADOW", "walk")
        walk = g.render(p)
        off = np.argwhere(np.abs(walk.astype(np.int16) - ref.astype(np.int16)).max(axis=2) > 1)
        assert len(off) <= 1 and all((y, x) == (210, 207) for y, x in off)
    finally:
        g.close()
        o.close()


def test_nearest_hit_whose_leaf_box_ends_an_ulp_behind_it(ctx, oracle, monkeypatch):
    """Seed 60253 at 1920 x 1080: triangle 63289 (30 x 20 units, 0.39 deep) is the nearest triangle the
    exact test accepts for sample 3 of pixel (1626, 82), an ulp nearer than triangle 60275 at t = 3.6285462.
    The reference reaches 60275 first and then tests the box of 63289's leaf with the ray cut at that
    distance: rounded as the box test rounds, the box begins an ulp behind it, and the leaf is never
    looked into. The outcome depends on the order of the search, which projection has to play back for
    such a sample (a nearest hit that fails its leaf's box with the ray cut at the hit itself)."""
    from raytracing_amd import scenes
    fuzz = _fuzz(monkeypatch)
    objs, sun = _scene(fuzz, 60253)
    w, h = 1920, 1080
    p = render_params(w, h, 2, reflect_depth=1, diffuse_depth=1, sun=sun)
    monkeypatch.setenv("RAYS_HIP_RASTER_RATIO", "1e9")
    monkeypatch.setenv("RAYS_HIP_SHADOW", "walk")
    g = Scene(ctx, objs)
    o = oracle.OracleScene(objs)
    try:
        assert _stats(g, p)["camera_by_projection"] == 1
        _, ids = g.render(p, want_hit_ids=True)
        monkeypatch.setenv("RAYS_HIP_RASTER", "0")
        _, ids_walk = g.render(p, want_hit_ids=True)
        pix, sub = 82 * w + 1626, 3
        ray = scenes.camera_rays(w, h, 2, pixels=[pix])[sub:sub + 1]
        ref = o.trace_closest(ray)
        assert ref["tri"][0] == 60275
        assert ids_walk[4 * pix + sub] == 60275 and ids[4 * pix + sub] == 60275
        assert np.array_equal(ids, ids_walk), f"{int((ids != ids_walk).sum())} hit ids differ"
        # four more samples of this frame hang on the order (two coplanar triangles an ulp apart at x = -2)
        for s in (3381520, 3381522, 3558194, 3573562):
            ray = scenes.camera_rays(w, h, 2, pixels=[s // 4])[s % 4:s % 4 + 1]
            assert o.trace_closest(ray)["tri"][0] == ids[s] == 20698
    finally:
        g.close()
        o.close()


def test_coplanar_layers_follow_the_order_of_the_reference_search(ctx, oracle, monkeypatch):
    """Seed 60000: layers of overlapping coplanar triangles. Where two of them cover a sample their distances
    differ by an ulp or not at all, and which one the reference reports depends on the order of its search
    (an earlier-numbered triangle makes the reference cull the packet of a later one that computes an ulp
    nearer): the least (t, id) is the wrong answer at some ten thousand samples of this frame. Every search
    has to give the reference's: projection, the wave-wide walks of the camera rays, and the three batch
    walks; so does the older form of projection."""
    from raytracing_amd import scenes
    fuzz = _fuzz(monkeypatch)
    objs, sun = _scene(fuzz, 60000)
    w, h = 1920, 1080
    p = render_params(w, h, 2, reflect_depth=1, diffuse_depth=1, sun=sun)
    monkeypatch.setenv("RAYS_HIP_SHADOW", "walk")
    g = Scene(ctx, objs)
    o = oracle.OracleScene(objs)
    try:
        _, ref_ids, _ = o.render(w, h, 2, reflect_depth=1, diffuse_depth=1, sun=sun, want_hit_ids=True)
        monkeypatch.setenv("RAYS_HIP_RASTER_RATIO", "1e9")
        assert _stats(g, p)["camera_by_projection"] == 1
        _, ids = g.render(p, want_hit_ids=True)
        assert np.array_equal(ids, ref_ids), f"projection: {int((ids != ref_ids).sum())} hit ids differ"
        monkeypatch.setenv("RAYS_HIP_RASTER", "atomic")
        _, ids = g.render(p, want_hit_ids=True)
        assert np.array_equal(ids, ref_ids), f"projection with atomics: {int((ids != ref_ids).sum())} hit ids differ"
        monkeypatch.setenv("RAYS_HIP_RASTER", "0")
        _, ids = g.render(p, want_hit_ids=True)
        assert np.array_equal(ids, ref_ids), f"tree walk: {int((ids != ref_ids).sum())} hit ids differ"
        # the batch walks on the rays of the pixels where the least (t, id) is not the reference's answer
        lin = None
        rows = np.unique(np.arange(0, w * h, 97))
        rays = scenes.camera_rays(w, h, 2, pixels=rows.tolist())
        want = ref_ids.reshape(-1, 4)[rows].reshape(-1)
        for how in ("persistent", "wave", "lane"):
            monkeypatch.setenv("RAYS_HIP_TRACE", how)
            got = g.trace_closest(rays)["tri"]
            assert np.array_equal(got, want), f"{how}: {int((got != want).sum())} of {len(want)} hit ids differ"
    finally:
        g.close()
        o.close()


@pytest.mark.parametrize("depths", [(3, 1), (2, 2)])
def test_order_rule_reaches_mirror_and_scatter_rays(ctx, oracle, monkeypatch, depths):
    """Seed 60000 again (coplanar layers whose triangles face either way), now shiny: mirror frames go
    through the frame kernel's own searches, frames with scatter rays through the persistent walk and the
    kernel behind it, and a secondary ray that picks the wrong one of two coplanar triangles leaves with
    the wrong normal. Frames against the oracle's, every path."""
    from raytracing_amd import _ffi
    fuzz = _fuzz(monkeypatch)
    objs, sun = _scene(fuzz, 60000)
    for ob in objs:
        ob["material"] = _ffi.material(1.0, 0.5, 1.0)
    w, h = 320, 180
    reflect, diffuse = depths
    p = render_params(w, h, 2, reflect_depth=reflect, diffuse_depth=diffuse, diffuse_scatt=4, sun=sun, seed=7)
    g = Scene(ctx, objs)
    o = oracle.OracleScene(objs)
    try:
        ref, ref_ids, _ = o.render(w, h, 2, reflect_depth=reflect, diffuse_depth=diffuse, diffuse_scatt=4, sun=sun, seed=7,
                                   want_hit_ids=True)
        for raster in ("1", "0"):
            monkeypatch.setenv("RAYS_HIP_RASTER", raster)
            monkeypatch.setenv("RAYS_HIP_RASTER_RATIO", "1e9")
            rgba, ids = g.render(p, want_hit_ids=True)
            assert np.array_equal(ids, ref_ids), f"RAYS_HIP_RASTER={raster}: {int((ids != ref_ids).sum())} hit ids differ"
            off = np.abs(rgba.astype(np.int16) - ref.astype(np.int16)).max(axis=2) > 1
            assert not off.any(), f"RAYS_HIP_RASTER={raster}: {int(off.sum())} pixels differ by more than 1 LSB"
    finally:
        g.close()
        o.close()


def test_axis_aligned_room_every_search_equals_the_oracle(ctx, oracle, monkeypatch):
    """Five tessellated axis-aligned walls (tools/flat_probe.py): every hit lies in the very plane of its
    leaf's box, a fifth of them fail the box test with the ray cut at the hit by an ulp or two, and rays
    along the walls' edges hit triangles whose leaves are closed to them. All of that has to come out as
    the reference has it, by every search."""
    from raytracing_amd import scenes
    spec = importlib.util.spec_from_file_location(
        "flat_probe", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "flat_probe.py"))
    flat = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(flat)
    objs = flat.scene_of("room", 40)
    sun = (3.0, 2.0, 5.0)
    p = render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=sun)
    monkeypatch.setenv("RAYS_HIP_SHADOW", "walk")
    g = Scene(ctx, objs)
    o = oracle.OracleScene(objs)
    try:
        ref, ref_ids, _ = o.render(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=sun, want_hit_ids=True)
        assert (ref_ids != 0xFFFFFFFF).mean() > 0.9
        monkeypatch.setenv("RAYS_HIP_RASTER_RATIO", "1e9")
        assert _stats(g, p)["camera_by_projection"] == 1
        for how in ("1", "atomic", "0"):
            monkeypatch.setenv("RAYS_HIP_RASTER", how)
            rgba, ids = g.render(p, want_hit_ids=True)
            assert np.array_equal(ids, ref_ids), f"RAYS_HIP_RASTER={how}: {int((ids != ref_ids).sum())} hit ids differ"
            assert np.abs(rgba.astype(np.int16) - ref.astype(np.int16)).max() <= 1
        rays = scenes.camera_rays(W, H, 2)
        for how in ("persistent", "wave", "lane"):
            monkeypatch.setenv("RAYS_HIP_TRACE", how)
            got = g.trace_closest(rays)["tri"]
            assert np.array_equal(got, ref_ids), f"{how}: {int((got != ref_ids).sum())} of {len(ref_ids)} hit ids differ"
    finally:
        g.close()
        o.close()


@pytest.mark.parametrize("seed", [20094, 5003])
def test_rays_sent_back_to_the_reference_search(ctx, monkeypatch, seed):
    """A ray whose occluder fails the leaf's box test goes to a second kernel that runs the reference's
    own shadow search for it (every leaf's box, then its eight triangles). RAYS_HIP_CELLS_REDO=all sends
    every ray the lists found occluded that way: the frame must not change."""
    fuzz = _fuzz(monkeypatch)
    objs, sun = _scene(fuzz, seed)
    p = render_params(320, 180, 2, reflect_depth=1, diffuse_depth=1, sun=sun)
    g = Scene(ctx, objs)
    try:
        monkeypatch.setenv("RAYS_HIP_SHADOW", "cells")
        st = _stats(g, p)
        assert st["shadow_by_cells"] == 1 and st["shadow_redone"] == 0
        rgba = g.render(p)
        monkeypatch.setenv("RAYS_HIP_CELLS_REDO", "all")
        assert _stats(g, p)["shadow_redone"] > 1000
        again = g.render(p)
        assert np.array_equal(rgba, again), f"{int((rgba != again).any(axis=2).sum())} pixels differ"
        assert (rgba != 0).any()
    finally:
        g.close()
