"""Shadow rays by light-space cells (per-cell triangle lists, exact test on queued pairs) against the
tree walk and against the CPU oracle's frames: soups under suns on every side, scenes of several
objects with trailing triangles, triangles seen edge-on from the light, and the stand-by when the
lists are too long. Frames are compared whole: ids equal, colours within 1 LSB of the oracle's and
identical between the two shadow paths."""
import numpy as np
import pytest

from raytracing_amd import RT_MISS, Scene, render_params, scenes
from test_gpu_quirks import _as_object, _obj, _slivers

pytestmark = pytest.mark.gpu


def _took_cells(stats):
    """The library's own word, and what it implies: the cells walk no tree, so every node visit of
    the frame is then a camera ray's."""
    if stats["shadow_by_cells"]:
        assert stats["node_visits"] == stats["primary_node_visits"]
        assert stats["shadow_list_entries"] >= stats["shadow_list_longest"] > 0 and stats["shadow_list_cells"] > 0
    return bool(stats["shadow_by_cells"])


def _render_stats(g, p):
    import torch
    fr = torch.zeros((p.height, p.width, 4), dtype=torch.uint8, device="cuda")
    st = g.render_device_stats(p, fr.data_ptr())
    torch.cuda.synchronize()
    return fr.cpu().numpy(), st


@pytest.fixture(scope="module")
def soup200k(ctx):
    objs = scenes.soup(200_000, 0x5EED0007)
    g = Scene(ctx, objs)
    yield objs, g
    g.close()


SUNS = {
    "reference_t0": scenes.SUN_T0,
    "reference_quarter_turn": scenes.sun_at(15.708),      # dominant axis x
    "overhead": (0.0, 1.0e5, 15.0),                       # dominant axis y
    "near_and_aside": (30.0, 40.0, -20.0),                # strong perspective
    "grazing": (1.0e5, 100.0, 15.0),
    "behind_the_camera": (-2.0, 3.0, -60.0),
}


@pytest.mark.parametrize("name", sorted(SUNS))
def test_cells_equal_the_tree_walk_under_suns_on_every_side(soup200k, monkeypatch, name):
    objs, g = soup200k
    p = render_params(1280, 720, 2, reflect_depth=1, diffuse_depth=1, sun=SUNS[name])
    rgba, st = _render_stats(g, p)
    assert _took_cells(st), "the light-space cells were not taken"
    assert st["rays"] > st["primary_rays"]
    monkeypatch.setenv("RAYS_HIP_SHADOW", "walk")
    rgba_walk, st_walk = _render_stats(g, p)
    assert not _took_cells(st_walk)
    assert st_walk["rays"] == st["rays"]
    assert np.array_equal(rgba, rgba_walk), f"{(rgba != rgba_walk).any(axis=2).sum()} pixels differ"


@pytest.mark.parametrize("name", ["reference_t0", "near_and_aside"])
def test_cells_frame_against_the_oracle(soup200k, oracle, name):
    objs, g = soup200k
    o = oracle.OracleScene(objs)
    try:
        W, H, stride = 1280, 720, 23
        kw = dict(reflect_depth=1, diffuse_depth=1, sun=SUNS[name])
        rgba, ids = g.render(render_params(W, H, 2, **kw), want_hit_ids=True)
        ref_rgba, ref_ids, _ = o.render(W, H, 2, pixel_stride=stride, want_hit_ids=True, **kw)
        pix = np.arange(0, W * H, stride)
        same = ids.reshape(-1, 4)[pix] == ref_ids.reshape(-1, 4)[pix]
        assert 1.0 - same.mean() <= 1e-5
        ok = same.all(axis=1)
        d = np.abs(rgba.reshape(-1, 4)[pix][ok].astype(np.int16) - ref_rgba.reshape(-1, 4)[pix][ok].astype(np.int16))
        assert d.max() <= 1, f"{(d > 1).sum()} channels differ by more than 1 LSB"
    finally:
        o.close()


def test_sun_inside_the_scene_and_long_lists_leave_it_to_the_tree_walk(soup200k, monkeypatch):
    objs, g = soup200k
    kw = dict(reflect_depth=1, diffuse_depth=1)
    p_in = render_params(640, 360, 2, sun=(0.5, 1.0, 15.0), **kw)
    rgba, st = _render_stats(g, p_in)
    assert not _took_cells(st)
    p = render_params(640, 360, 2, **kw)
    rgba, st = _render_stats(g, p)
    assert _took_cells(st)
    monkeypatch.setenv("RAYS_HIP_SHADOW_CELL_MEAN", "1")     # no scene has lists this short
    rgba_walk, st_walk = _render_stats(g, p)
    assert not _took_cells(st_walk), "the stand-by did not take the launch"
    assert np.array_equal(rgba, rgba_walk)


def _frames(ctx, oracle, objs, monkeypatch, sizes, sun):
    g, o = Scene(ctx, objs), oracle.OracleScene(objs)
    try:
        for (W, H, grid) in sizes:
            kw = dict(reflect_depth=1, diffuse_depth=1, sun=sun)
            ref_rgba, ref_ids, _ = o.render(W, H, grid, want_hit_ids=True, **kw)
            assert (ref_ids != RT_MISS).sum() > 20
            p = render_params(W, H, grid, **kw)
            monkeypatch.setenv("RAYS_HIP_SHADOW", "cells")
            rgba, st = _render_stats(g, p)
            assert _took_cells(st), "the light-space cells were not taken"
            assert st["rays"] > st["primary_rays"], "no shadow rays in this frame"
            _, ids = g.render(p, want_hit_ids=True)
            monkeypatch.setenv("RAYS_HIP_SHADOW", "walk")
            rgba_walk, _ = _render_stats(g, p)
            monkeypatch.delenv("RAYS_HIP_SHADOW")
            assert np.array_equal(ids, ref_ids)
            d = np.abs(rgba.astype(np.int16) - ref_rgba.astype(np.int16))
            assert d.max() <= 1, f"{W}x{H}x{grid}: {(d > 1).sum()} channels off the oracle's by more than 1 LSB"
            assert np.array_equal(rgba, rgba_walk)
    finally:
        g.close()
        o.close()


def test_cells_with_several_objects_and_trailing_triangles(ctx, oracle, monkeypatch):
    """Objects of 3, 8, 11 and 19 triangles (trailing triangles take the scalar test, each object's
    box gates its triangles) that shadow one another, in several World orders."""
    a = _obj([7.0, 6.0, 8.0], shift=(-0.4, 0.1))
    b = _obj([6.5 + 0.25 * i for i in range(8)], shift=(0.3, -0.2), scale=1.5)
    c = _obj([9.0 - 0.3 * i for i in range(11)], shift=(0.0, 0.3), scale=2.0)
    d = _obj([5.5 + 0.2 * ((7 * i) % 19) for i in range(19)], shift=(0.2, 0.0), scale=1.2)
    floor = _obj([12.0] * 8, scale=6.0)
    for tri in (a, b, c, d, floor):     # face the light below
        tri["tri_data"][:, 0:3] = np.array([0.0, 0.0, -1.0], dtype=np.float32)
    for order in ([a, b, c, d, floor], [floor, d, c, b, a], [c, a, floor, d, b]):
        _frames(ctx, oracle, order, monkeypatch, [(96, 64, 2), (120, 80, 1)], sun=(3.0, 4.0, -30.0))


def _edge_on_to(rng, sun, n, size):
    """n triangles whose planes pass within a hair of the sun: every shadow ray meets them edge-on."""
    sun = np.asarray(sun, dtype=np.float64)
    def towards():
        t = np.stack([rng.uniform(-5, 5, n), rng.uniform(-3, 3, n), rng.uniform(6, 16, n)], axis=1)
        return t - sun
    d1, d2 = towards(), towards()
    a = sun + d1 * rng.uniform(0.7, 0.9, (n, 1))
    b = sun + d1 * (1.0 + rng.uniform(0.0, size, (n, 1)) / np.linalg.norm(d1, axis=1, keepdims=True))
    c = sun + d2 * rng.uniform(0.8, 1.1, (n, 1))
    off = rng.normal(size=(n, 3)) * rng.choice([0.0, 1e-5, 1e-3, 3e-2], size=(n, 1))
    v = np.stack([a + off, b + off, c + off], axis=1).astype(np.float32)
    v[..., 2] = np.maximum(v[..., 2], 0.5)
    return v


@pytest.mark.parametrize("kind", ["edge_on", "slivers", "mixed"])
def test_cells_with_triangles_edge_on_to_the_light(ctx, oracle, monkeypatch, kind):
    """Occluders that the light sees edge-on or as needles, above receivers that face it: the exact
    test's determinant is a rounding error along such a triangle, and the rays it accepts are ragged.
    Both shadow paths must give the oracle's frame."""
    rng = np.random.default_rng({"edge_on": 21, "slivers": 22, "mixed": 23}[kind])
    sun = (3.0, 40.0, -30.0)
    recv = scenes.soup(4096, 0x5EED0011)[0]
    rv = recv["verts"].reshape(-1, 3, 3).copy()
    rv[..., 2] = 14.0 + (rv[..., 2] - 5.0) * 0.1     # a slab of small triangles at z = 14 .. 16
    big = np.array([[[-8, -5, 17], [8, -5, 17], [0, 6, 17.5]]] * 8, dtype=np.float32)
    if kind == "edge_on":
        occ = _edge_on_to(rng, sun, 64, 12.0)
    elif kind == "slivers":
        occ = np.concatenate([_slivers(rng, 32, 30.0, 0.02, 6, 12), _slivers(rng, 32, 12.0, 0.2, 5, 13)])
    else:
        occ = np.concatenate([_edge_on_to(rng, sun, 32, 20.0), _slivers(rng, 24, 20.0, 0.05, 6, 12),
                              _edge_on_to(rng, sun, 8, 2.0)])
    objs = [_as_object(np.concatenate([occ, rv, big]).astype(np.float32))]
    _frames(ctx, oracle, objs, monkeypatch, [(160, 90, 2), (128, 72, 4)], sun=sun)


def _layers(rng, axis, n, gaps):
    """n stacks of triangles that lie across `axis` (0, 1 or 2), each stack a receiver with copies of
    itself, shrunk and shifted a little, at the given gaps towards smaller coordinates of that axis:
    seen from a sun down that axis the copies start where the receiver's shadow rays start."""
    out = []
    for _ in range(n):
        c = np.array([rng.uniform(-4, 4), rng.uniform(-2.5, 2.5), rng.uniform(7, 13)])
        size = rng.uniform(0.4, 2.5)
        tri = rng.uniform(-1, 1, (3, 3)) * size
        tri[:, axis] = rng.choice([0.0, 0.0, 1e-3, 0.05]) * rng.uniform(-1, 1, 3)   # flat, or nearly
        depth = 0.0
        for gap in gaps:
            depth -= gap
            t = tri * rng.uniform(0.5, 1.0) + rng.uniform(-0.1, 0.1, 3) * size
            t[:, axis] = tri[:, axis] + depth
            out.append(c + t)
        out.append(c + tri)
    v = np.array(out, dtype=np.float32)
    v[..., 2] = np.maximum(v[..., 2], 0.5)
    return v


@pytest.mark.parametrize("axis,sun", [(2, (1.0, 2.0, -1.0e5)), (2, (3.0, 4.0, -30.0)), (1, (0.5, 1.0e5, 9.0)),
                                      (0, (-60.0, 2.0, 8.0))])
def test_cells_with_occluders_at_the_depth_of_the_rays(ctx, oracle, monkeypatch, axis, sun):
    """A list entry is skipped by rays that start nearer the sun than the triangle lies. Here the
    occluders lie at the very depth the shadow rays start from: coplanar copies, copies a few units
    of rounding, 1e-5, 1e-3 and 0.1 in front, a ray's own triangle (distance zero, either sign after
    rounding), and tilted triangles that cut through the stacks."""
    rng = np.random.default_rng(31 + axis + int(abs(sun[0])))
    stacks = _layers(rng, axis, 40, [0.0, 2e-6, 1e-5, 1e-3, 0.1])
    cutters = _slivers(rng, 16, 6.0, 1.5, 7, 13)
    pad = (-len(stacks) - len(cutters)) % 8     # whole leaves: every triangle takes the eight-wide test
    fill = scenes.soup(pad + 64, 0x5EED0012)[0]["verts"].reshape(-1, 3, 3)
    objs = [_as_object(np.concatenate([stacks, cutters, fill]).astype(np.float32))]
    _frames(ctx, oracle, objs, monkeypatch, [(160, 90, 2), (96, 54, 4)], sun=sun)


def test_cells_equal_the_tree_walk_on_random_scenes(ctx, monkeypatch):
    """The scenes of tools/cells_fuzz.py (needles, triangles edge-on to the sun, flat layers at equal
    depth, sizes from 0.05 to 1 unit, suns 30 to 100,000 units away on every side): the frame by cells must
    be the frame by tree walk. The tool itself runs as many scenes as it is asked for and also checks the
    camera rays by projection against the tree walk."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location(
        "cells_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "cells_fuzz.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    taken = 0
    for seed in range(5000, 5008):
        rng = np.random.default_rng(seed)
        sun = fuzz.sun_of(rng)
        g = Scene(ctx, fuzz.scene_of(rng, sun))
        try:
            p = render_params(640, 360, 2, reflect_depth=1, diffuse_depth=1, sun=sun)
            monkeypatch.setenv("RAYS_HIP_SHADOW", "cells")
            rgba, st = _render_stats(g, p)
            taken += _took_cells(st)
            monkeypatch.setenv("RAYS_HIP_SHADOW", "walk")
            rgba_walk, _ = _render_stats(g, p)
            assert np.array_equal(rgba, rgba_walk), f"seed {seed}: {(rgba != rgba_walk).any(axis=2).sum()} pixels differ"
        finally:
            g.close()
    assert taken >= 6, "the scenes were meant for the light-space cells"


@pytest.mark.parametrize("shards", [2, 5])
def test_cells_in_sharded_frames(soup200k, monkeypatch, shards):
    """A shard's rays are a fraction of the frame's, spread over all the light-space cells; the lists
    are the whole scene's on every shard. The shards' tiles must add up to the frame rendered whole,
    and to the frame the tree walk gives."""
    objs, g = soup200k
    W, H = 1000, 600     # not a multiple of the tile size
    kw = dict(reflect_depth=1, diffuse_depth=1, sun=SUNS["reference_quarter_turn"])
    full, st = _render_stats(g, render_params(W, H, 2, **kw))
    assert _took_cells(st)
    acc = np.zeros_like(full)
    for r in range(shards):
        part = g.render(render_params(W, H, 2, shard_index=r, shard_count=shards, **kw))
        assert not (acc.astype(bool) & part.astype(bool))[..., 3].any(), "two shards wrote one pixel"
        acc |= part
    assert np.array_equal(acc, full)
    monkeypatch.setenv("RAYS_HIP_SHADOW", "walk")
    assert np.array_equal(g.render(render_params(W, H, 2, **kw)), full)


def test_cells_follow_a_moving_sun_frame_by_frame(soup200k, monkeypatch):
    """The lists are rebuilt for every frame's sun: frames enqueued back to back (two in flight inside
    the library) under a sun that moves must equal the tree walk's frames of the same suns."""
    import torch
    objs, g = soup200k
    W, H = 640, 360
    params = [render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=scenes.sun_at(3.3 * k)) for k in range(5)]
    monkeypatch.setenv("RAYS_HIP_SHADOW", "walk")
    ref = [g.render(p) for p in params]
    monkeypatch.delenv("RAYS_HIP_SHADOW")
    monkeypatch.setenv("RAYS_HIP_FRAMES_IN_FLIGHT", "2")
    bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in params]
    torch.cuda.synchronize()
    for p, buf in zip(params, bufs):
        g.render_device(p, buf.data_ptr())
    torch.cuda.synchronize()
    for k, buf in enumerate(bufs):
        assert np.array_equal(buf.cpu().numpy(), ref[k]), f"frame {k}"
    assert any(not np.array_equal(ref[0], r) for r in ref[1:])
