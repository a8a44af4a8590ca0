"""Quirks of the reference's search that depend on object order and on rounding at the edge of
acceptance, through every walk the library has, against the CPU oracle."""
import numpy as np
import pytest

from raytracing_amd import RT_MISS, Scene, render_params, scenes

pytestmark = pytest.mark.gpu

TRI = np.array([[-1, -1, 5], [1, -1, 5], [0, 1, 5]], dtype=np.float32)


def _obj(zs, shift=(0.0, 0.0), scale=1.0, material=(0.3, 0.0, 1.0)):
    """One triangle per entry of zs, fronto-parallel, at that depth."""
    n = len(zs)
    verts = np.concatenate([
        (TRI * np.array([scale, scale, 1], dtype=np.float32) +
         np.array([shift[0], shift[1], z - 5.0], dtype=np.float32)).astype(np.float32) for z in zs])
    return {"verts": verts, "tris": np.arange(3 * n, dtype=np.uint32).reshape(n, 3),
            "tri_data": np.tile(np.array([[0, 0, -1, 1, 1, 1]], dtype=np.float32), (n, 1)),
            "material": material}


def _bit_equal(gpu, ref):
    assert np.array_equal(gpu["tri"], ref["tri"]), f"{(gpu['tri'] != ref['tri']).sum()} hit ids differ"
    hit = ref["tri"] != RT_MISS
    for f in ("t", "u", "v"):
        assert np.array_equal(gpu[f][hit].view(np.uint32), ref[f][hit].view(np.uint32)), f


def _all_walks(ctx, oracle, objs, monkeypatch, W=96, H=64):
    """Batch nearest-hit search by the persistent walk, the plain per-lane walk, the wave walks and
    the reference-order scan, the any-hit search, and a rendered frame's hit ids (camera-ray kernel
    of the pipeline, and the one-kernel frame), all against the oracle."""
    g, o = Scene(ctx, objs), oracle.OracleScene(objs)
    try:
        rays = scenes.camera_rays(W, H, 2)
        ref = o.trace_closest(rays)
        assert (ref["tri"] != RT_MISS).any()
        _bit_equal(g.trace_closest(rays), ref)
        for mode in ("lane", "wave"):
            monkeypatch.setenv("RAYS_HIP_TRACE", mode)
            _bit_equal(g.trace_closest(rays), ref)
            if mode == "lane":
                assert np.array_equal(g.trace_any(rays), o.trace_any(rays))
            monkeypatch.delenv("RAYS_HIP_TRACE")
        _bit_equal(g.trace_closest_linear(rays), ref)
        assert np.array_equal(g.trace_any(rays), o.trace_any(rays))
        kw = dict(reflect_depth=1, diffuse_depth=1, sun=(3.0, 40.0, -30.0))
        ref_rgba, ref_ids, _ = o.render(W, H, 2, want_hit_ids=True, **kw)
        assert np.array_equal(ref_ids, ref["tri"])
        for mega in (False, True):
            if mega:
                monkeypatch.setenv("RAYS_HIP_MEGAKERNEL", "1")
            rgba, ids = g.render(render_params(W, H, 2, **kw), want_hit_ids=True)
            if mega:
                monkeypatch.delenv("RAYS_HIP_MEGAKERNEL")
            assert np.array_equal(ids, ref_ids), f"{(ids != ref_ids).sum()} frame ids differ (one kernel: {mega})"
            assert np.abs(rgba.astype(np.int16) - ref_rgba.astype(np.int16)).max() <= 1
        return ref
    finally:
        g.close()
        o.close()


def test_trailing_triangles_leave_closest_stale_for_later_objects(ctx, oracle, monkeypatch):
    """src/main.rs:157-169: a hit on one of an object's trailing n % 8 triangles does not lower
    `closest`, so an object later in the World replaces it even from farther away; in the reverse
    order the nearer trailing triangle wins because it compares against the lowered distance."""
    near3 = _obj([5.0, 5.0, 5.0])
    far8 = _obj([9.0] * 8)
    centre = (64 // 2) * 96 * 4 + (96 // 2) * 4     # first sample of the centre pixel
    ref = _all_walks(ctx, oracle, [near3, far8], monkeypatch)
    assert ref["tri"][centre] >= 3 and ref["t"][centre] == 9.0      # the far object wins
    ref = _all_walks(ctx, oracle, [far8, near3], monkeypatch)
    assert ref["tri"][centre] == 10 and ref["t"][centre] == 5.0     # last trailing triangle of the near one


def test_mixed_objects_with_trailing_triangles_in_every_order(ctx, oracle, monkeypatch):
    """Objects of 3, 8, 11 and 19 triangles (0, 1, 1, 2 eight-wide leaves; 3, 0, 3, 3 trailing) that
    overlap on the screen at interleaved depths, in several World orders."""
    a = _obj([7.0, 6.0, 8.0], shift=(-0.4, 0.1))
    b = _obj([6.5 + 0.25 * i for i in range(8)], shift=(0.3, -0.2), scale=1.5)
    c = _obj([9.0 - 0.3 * i for i in range(11)], shift=(0.0, 0.3), scale=2.0)
    d = _obj([5.5 + 0.2 * ((7 * i) % 19) for i in range(19)], shift=(0.2, 0.0), scale=1.2)
    for order in ([a, b, c, d], [d, c, b, a], [c, a, d, b], [b, d, a, c]):
        _all_walks(ctx, oracle, order, monkeypatch)


def _slivers(rng, n, length, width, z0, z1):
    """n long thin triangles: a base edge of `length` and an apex `width` off the middle of it."""
    c = np.stack([rng.uniform(-6, 6, n), rng.uniform(-4, 4, n), rng.uniform(z0, z1, n)], axis=1)
    e = rng.normal(size=(n, 3))
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    w = rng.normal(size=(n, 3))
    w -= (w * e).sum(axis=1, keepdims=True) * e
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    a = c - 0.5 * length * e
    b = c + 0.5 * length * e
    cc = c + width * w
    v = np.stack([a, b, cc], axis=1).astype(np.float32)
    # keep everything in front of the camera plane
    v[..., 2] = np.maximum(v[..., 2], 0.5)
    return v


def _edge_on(rng, n, size):
    """n large triangles whose planes pass within a hair of the camera (the origin): seen edge-on,
    the determinant of the ray-triangle test is a rounding error along their whole length."""
    d1 = np.stack([rng.uniform(-0.8, 0.8, n), rng.uniform(-0.5, 0.5, n), np.ones(n)], axis=1)
    d2 = np.stack([rng.uniform(-0.8, 0.8, n), rng.uniform(-0.5, 0.5, n), np.ones(n)], axis=1)
    a = d1 * rng.uniform(2.0, 6.0, (n, 1))
    b = d1 * rng.uniform(size * 0.5, size, (n, 1))
    c = d2 * rng.uniform(size * 0.5, size, (n, 1))
    off = rng.normal(size=(n, 3)) * rng.choice([0.0, 1e-5, 1e-3, 3e-2], size=(n, 1))
    return np.stack([a + off, b + off, c + off], axis=1).astype(np.float32)


def _as_object(v):
    n = len(v)
    nrm = scenes.calculate_normals(v[:, 0], v[:, 1], v[:, 2])
    nrm = np.nan_to_num(nrm, nan=0.0, posinf=0.0, neginf=0.0)
    col = np.tile(np.array([[0.8, 0.6, 0.4]], dtype=np.float32), (n, 1))
    return {"verts": v.reshape(-1, 3), "tris": np.arange(3 * n, dtype=np.uint32).reshape(n, 3),
            "tri_data": np.concatenate([nrm, col], axis=1).astype(np.float32), "material": (0.3, 0.0, 1.0)}


@pytest.mark.parametrize("kind", ["slivers", "edge_on", "mixed"])
def test_camera_rays_of_large_slivers_and_edge_on_triangles(ctx, oracle, monkeypatch, kind):
    """Triangles with edges of 10 to 80 units that are needle-thin or seen edge-on: the determinant of
    the ray-triangle test comes within a few EPSILON of its threshold and u, v, t carry large rounding
    errors, so the set of samples the reference accepts is ragged. Camera rays by projection (when the
    library takes it) and by the tree walk must both give the oracle's ids."""
    rng = np.random.default_rng({"slivers": 11, "edge_on": 12, "mixed": 13}[kind])
    if kind == "slivers":
        v = np.concatenate([_slivers(rng, 32, 40.0, 0.02, 8, 30), _slivers(rng, 32, 80.0, 0.3, 10, 40)])
    elif kind == "edge_on":
        v = np.concatenate([_edge_on(rng, 40, 30.0), _edge_on(rng, 24, 80.0)])
    else:
        v = np.concatenate([_slivers(rng, 24, 10.0, 0.05, 6, 20), _edge_on(rng, 24, 20.0),
                            _slivers(rng, 16, 60.0, 1.0, 10, 40)])
    objs = [_as_object(v)]
    g, o = Scene(ctx, objs), oracle.OracleScene(objs)
    try:
        monkeypatch.setenv("RAYS_HIP_RASTER_RATIO", "1e9")
        for (W, H, grid) in ((160, 90, 2), (256, 144, 1), (96, 54, 4)):
            kw = dict(reflect_depth=1, diffuse_depth=1)
            ref_rgba, ref_ids, _ = o.render(W, H, grid, want_hit_ids=True, **kw)
            assert (ref_ids != RT_MISS).sum() > 20
            p = render_params(W, H, grid, **kw)
            rgba, ids = g.render(p, want_hit_ids=True)
            assert np.array_equal(ids, ref_ids), f"{kind} {W}x{H}x{grid}: {(ids != ref_ids).sum()} ids differ"
            monkeypatch.setenv("RAYS_HIP_RASTER", "0")
            rgba_w, ids_w = g.render(p, want_hit_ids=True)
            monkeypatch.delenv("RAYS_HIP_RASTER")
            assert np.array_equal(ids_w, ref_ids), f"{kind} {W}x{H}x{grid} tree walk: {(ids_w != ref_ids).sum()} ids differ"
            assert np.array_equal(rgba_w, rgba)
            rays = scenes.camera_rays(W, H, grid)
            _bit_equal(g.trace_closest(rays), o.trace_closest(rays))
    finally:
        g.close()
        o.close()


@pytest.mark.parametrize("device_build", [False, True])
def test_a_vertex_that_is_not_finite_hides_nothing_else(ctx, oracle, monkeypatch, device_build):
    """Triangles with an infinite coordinate can be hit by no ray (their test yields infinities and NaNs
    made by the arithmetic, which the reference does not accept either), but they sit in the tree among
    the others: every walk must still find every other triangle, as the reference-order scan and the
    oracle do. (An infinite box used to make every box above it infinite, and the compact form of the
    nodes dropped such children.)"""
    objs = scenes.soup(32768, 0x5EED0021)
    v = objs[0]["verts"].copy().reshape(-1, 3, 3)
    v[17, 0, 0] = np.inf
    v[18000, 2, 1] = -np.inf
    v[32000, 1, 2] = np.inf
    objs[0]["verts"] = v.reshape(-1, 3)
    if device_build:
        monkeypatch.setenv("RAYS_HIP_BUILD", "device")
    g, o = Scene(ctx, objs), oracle.OracleScene(objs)
    try:
        rays = scenes.camera_rays(160, 90, 2)
        ref = o.trace_closest(rays)
        assert (ref["tri"] != RT_MISS).mean() > 0.05 and not np.isin(ref["tri"], [17, 18000, 32000]).any()
        _bit_equal(g.trace_closest(rays), ref)
        _bit_equal(g.trace_closest_linear(rays), ref)
        for mode in ("lane", "wave"):
            monkeypatch.setenv("RAYS_HIP_TRACE", mode)
            _bit_equal(g.trace_closest(rays), ref)
            monkeypatch.delenv("RAYS_HIP_TRACE")
        assert np.array_equal(g.trace_any(rays), o.trace_any(rays))
        kw = dict(reflect_depth=1, diffuse_depth=1, sun=(3.0, 40.0, -30.0))
        ref_rgba, ref_ids, _ = o.render(160, 90, 2, want_hit_ids=True, **kw)
        rgba, ids = g.render(render_params(160, 90, 2, **kw), want_hit_ids=True)
        assert np.array_equal(ids, ref_ids)
        assert np.abs(rgba.astype(np.int16) - ref_rgba.astype(np.int16)).max() <= 1
    finally:
        g.close()
        o.close()


def _projected_frame_check(ctx, oracle, objs, monkeypatch, sizes, expect_projection=True):
    import torch

    g, o = Scene(ctx, objs), oracle.OracleScene(objs)
    try:
        monkeypatch.setenv("RAYS_HIP_RASTER_RATIO", "1e9")
        for (W, H, grid) in sizes:
            kw = dict(reflect_depth=1, diffuse_depth=1, sun=(3.0, 40.0, -30.0))
            p = render_params(W, H, grid, **kw)
            frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            assert g.render_device_stats(p, frame.data_ptr())["camera_by_projection"] == int(expect_projection)
            rgba, ids = g.render(p, want_hit_ids=True)
            ref_rgba, ref_ids, _ = o.render(W, H, grid, want_hit_ids=True, **kw)
            assert np.array_equal(ids, ref_ids), f"{(ids != ref_ids).sum()} hit ids differ at {W}x{H}x{grid}"
            assert np.abs(rgba.astype(np.int16) - ref_rgba.astype(np.int16)).max() <= 1
            monkeypatch.setenv("RAYS_HIP_RASTER", "0")
            rgba_walk, ids_walk = g.render(p, want_hit_ids=True)
            monkeypatch.delenv("RAYS_HIP_RASTER")
            assert np.array_equal(ids_walk, ids) and np.array_equal(rgba_walk, rgba)
    finally:
        g.close()
        o.close()


def test_projection_with_trailing_triangles_after_the_eight_wide_ones(ctx, oracle, monkeypatch):
    """Camera rays by projection in scenes whose trailing n % 8 triangles all come after the last
    eight-wide triangle in object order (the reference's own scene is one: two teapots of whole leaves,
    then a floor of two triangles): the trailing triangles are tested sample by sample against the nearest
    eight-wide hit of the whole scene, the last one that beats it wins. In any other order that distance
    depends on the objects in between, and the frame is left to the tree walk."""
    a = _obj([7.0, 6.0, 8.0], shift=(-0.4, 0.1))                                      # 3 trailing
    b = _obj([6.5 + 0.25 * i for i in range(8)], shift=(0.3, -0.2), scale=1.5)         # one leaf
    c = _obj([9.0 - 0.3 * i for i in range(11)], shift=(0.0, 0.3), scale=2.0)          # one leaf + 3 trailing
    d = _obj([5.5 + 0.2 * ((7 * i) % 19) for i in range(19)], shift=(0.2, 0.0), scale=1.2)   # two leaves + 3 trailing
    e = _obj([10.0, 4.0], shift=(0.1, -0.1), scale=3.0)                                # 2 trailing, one in front of everything
    sizes = [(96, 64, 2), (50, 30, 4)]
    for order in ([b, c, a], [b, d, a, e], [b, b, d], [d], [b, e]):
        _projected_frame_check(ctx, oracle, order, monkeypatch, sizes)
    for order in ([a, b], [c, b], [b, c, d]):
        _projected_frame_check(ctx, oracle, order, monkeypatch, sizes[:1], expect_projection=False)


def test_default_scene_by_projection(ctx, oracle, default_objects, monkeypatch):
    """The reference's own scene (src/main.rs:406-448): the floor's two triangles are trailing triangles
    10,000 units wide that pass behind the camera; the teapots are projected, the floor is tested per
    sample."""
    _projected_frame_check(ctx, oracle, default_objects, monkeypatch, [(320, 180, 2), (200, 200, 1), (160, 90, 4)])


@pytest.mark.parametrize("shards_below", ["0", "4000000000"])
def test_batch_sizes_around_the_hand_out_boundaries(ctx, soup64k, monkeypatch, shards_below):
    """The persistent walk gives the first half of a batch to its waves by number and hands the rest out in
    chunks of 64 through one counter or through eight (small scenes; RAYS_HIP_WALK_SHARDS_BELOW picks at scene
    build). Batches of one ray, of a chunk more or less, of a whole grid's worth (8,192 waves x 64) more or
    less and of no multiple of anything must come out ray for ray as from the plain per-lane walk."""
    monkeypatch.setenv("RAYS_HIP_WALK_SHARDS_BELOW", shards_below)
    g = Scene(ctx, soup64k)
    try:
        rng = np.random.default_rng(11)
        n_max = 2 * 8192 * 64 + 777
        org = rng.uniform(-8, 8, (n_max, 3)).astype(np.float32) + np.array([0, 0, 12], dtype=np.float32)
        d = rng.normal(size=(n_max, 3)).astype(np.float32)
        rays = np.concatenate([org, d], axis=1)
        monkeypatch.setenv("RAYS_HIP_TRACE", "lane")
        want = g.trace_closest(rays)
        want_any = g.trace_any(rays)
        monkeypatch.delenv("RAYS_HIP_TRACE")
        for n in (1, 63, 64, 65, 8 * 64 + 1, 8192 * 64 - 1, 8192 * 64, 8192 * 64 + 65, n_max):
            got = g.trace_closest(rays[:n])
            assert np.array_equal(got["tri"], want["tri"][:n]), n
            hit = got["tri"] != 0xFFFFFFFF
            assert np.array_equal(got["t"][hit].view(np.uint32), want["t"][:n][hit].view(np.uint32)), n
            assert np.array_equal(g.trace_any(rays[:n]), want_any[:n]), n
        assert (want["tri"] != 0xFFFFFFFF).sum() > 1000
    finally:
        g.close()
